"""64 GEN1-shaped streams in one call: tile bins + split pass (the default above 512 pairs) against sub-tile bins (direct mode: the
consumers gather their own lists, no split pass), TAF and Event Volume.   python tools/time_direct64.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from frlw_evd_amd import _lib, synth, event_representation as er
H, W, K, nw, win, n, B = 240, 304, 8, 8, 10_000, 1_000_000, 64
recs = [synth.to_dat8(synth.synth_events(1003 + j, n, W, H, nw * win)) for j in range(B)]
offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
dat = torch.from_numpy(np.concatenate(recs).view(np.uint8).reshape(-1, 8)).cuda()
state = torch.full((B, H, W, 2, K), -6000.0, device="cuda")
recs = [synth.to_dat8(synth.synth_events(2003 + j, n, W, H, 250_000, t_offset=1)) for j in range(B)]
dat_ev = torch.from_numpy(np.concatenate(recs).view(np.uint8).reshape(-1, 8)).cuda()
del recs
def taf(): er.encode_taf_batch(dat, offs, (H, W), state, 0, win, nw, K, check=False)
def ev(): er.encode_ev_batch(dat_ev, offs, (H, W), 250_000, 250_000, 5, check=False)
def run(fn, knob, steps=20):
    er.TUNING = _lib.FrlwTuning(direct_bins=knob)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    er.raise_deferred()
    return e0.elapsed_time(e1) / steps * 1e3
for rep in range(3):
    print("TAF x64: " + "  ".join(f"direct_bins={k}: {run(taf, k):.1f} us" for k in (0, 1)) + "   EV x64: " + "  ".join(f"direct_bins={k}: {run(ev, k):.1f} us" for k in (0, 1)), flush=True)
