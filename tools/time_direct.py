#!/usr/bin/env python3
"""Direct partition mode (frlw_tuning_t::direct_bins) against tile bins + split pass, GEN1-shaped TAF / Event Volume batches of
B streams: device time per call for B = 1 .. 64 (where is the cross-over?).   python tools/time_direct.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frlw_evd_amd import _lib, event_representation as er, synth  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    H, W, K, win, n_win = 240, 304, 8, 10_000, 8
    n_ev = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    base = synth.to_dat8(synth.synth_events(7, n_ev, W, H, n_win * win))
    for B in (1, 2, 4, 7, 8, 12, 16, 24, 32, 64):
        rec = np.concatenate([base] * B)
        offs = np.arange(B + 1) * len(base)
        dat = torch.from_numpy(np.ascontiguousarray(rec).view(np.uint8).reshape(-1, 8).copy()).cuda()
        st = torch.full((B, H, W, 2, K), -6000.0, device="cuda")
        row = [f"B={B:3d}"]
        for direct in (1, 0):
            er.TUNING = _lib.FrlwTuning(direct_bins=direct)
            t_taf = timed(lambda: er.encode_taf_batch(dat, offs, (H, W), st, 0, win, n_win, K, check=False))
            t_ev = timed(lambda: er.encode_ev_batch(dat, offs, (H, W), n_win * win, n_win * win, 5, check=False))
            row.append(f"direct={direct}: TAF {t_taf:8.1f} us  EV {t_ev:8.1f} us")
        er.TUNING = None
        er.raise_deferred()
        print("   ".join(row))


if __name__ == "__main__":
    main()
