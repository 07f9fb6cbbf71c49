"""Same-box A/B of frlw_tuning_t::walk_window_table (kf_split_whole<true> leaves the walk its window starts): the headline TAF
encode (10 M events, 1280x720) with the knob at 0 and 1, alternating.   WTAB=0|1 python tools/time_wtab.py  runs one variant only
(for a per-kernel trace under rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from frlw_evd_amd import _lib, synth, event_representation as er
H, W, K, nw, win, n = 720, 1280, 8, 8, 10_000, 10_000_000
hot = os.environ.get("HOT") == "1"
ev = synth.synth_events(1003, n, W, H, nw * win, hotspot=hot)
dat = torch.from_numpy(synth.to_dat8(ev).view(np.uint8).reshape(-1, 8)).cuda()
state = torch.full((H, W, 2, K), -6000.0, device="cuda")
def run(knob, steps):
    er.TUNING = _lib.FrlwTuning(walk_window_table=knob)
    for _ in range(3):
        er.encode_taf_dat(dat, (H, W), state, 0, win, nw, K, check=False, fast=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        er.encode_taf_dat(dat, (H, W), state, 0, win, nw, K, check=False, fast=True)
    e1.record(); torch.cuda.synchronize()
    er.raise_deferred()
    return e0.elapsed_time(e1) / steps * 1e3
only = os.environ.get("WTAB")
if only is not None:
    print(f"walk_window_table={only}: {run(int(only), 30):.1f} us"); sys.exit(0)
for rep in range(4):
    print("  ".join(f"wtab={k}: {run(k, 40):.1f} us" for k in (0, 1)), flush=True)
