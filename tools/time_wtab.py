"""Same-box A/B of frlw_tuning_t::walk_window_table (kf_split_whole<true> leaves the walk its window starts): the headline TAF
encode (10 M events, 1280x720) with the knob at 0 and 1, alternating.   WTAB=0|1 python tools/time_wtab.py  runs one variant only
(for a per-kernel trace under rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from frlw_evd_amd import _lib, synth, event_representation as er
hot = os.environ.get("HOT") == "1"
if os.environ.get("CFG") == "gen1x64":  # 64 GEN1-shaped streams of 1 M events in one call (tile bins, 8192-event chunks)
    H, W, K, nw, win, n, B = 240, 304, 8, 8, 10_000, 1_000_000, 64
else:
    H, W, K, nw, win, n, B = 720, 1280, 8, 8, 10_000, 10_000_000, 1
recs = [synth.to_dat8(synth.synth_events(1003 + j, n, W, H, nw * win, hotspot=hot)) for j in range(B)]
offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
dat = torch.from_numpy(np.concatenate(recs).view(np.uint8).reshape(-1, 8)).cuda()
del recs
state = torch.full((B, H, W, 2, K), -6000.0, device="cuda")
def enc():
    er.encode_taf_batch(dat, offs, (H, W), state, 0, win, nw, K, check=False)
def run(knob, steps):
    er.TUNING = _lib.FrlwTuning(walk_window_table=knob)
    for _ in range(3):
        enc()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        enc()
    e1.record(); torch.cuda.synchronize()
    er.raise_deferred()
    return e0.elapsed_time(e1) / steps * 1e3
only = os.environ.get("WTAB")
if only is not None:
    print(f"walk_window_table={only}: {run(int(only), 30):.1f} us"); sys.exit(0)
for rep in range(4):
    print("  ".join(f"wtab={k}: {run(k, 40):.1f} us" for k in (0, 1)), flush=True)
