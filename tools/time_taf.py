#!/usr/bin/env python3
"""Device time of one TAF encode, fast (csrc/taf_fast.hip) vs general (csrc/encoders.hip) path, several workloads.
    python tools/time_taf.py [--only mpx,mpx_hot,gen1,gen1_b64] [--steps 20] [--no-general]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frlw_evd_amd import event_representation as er, synth  # noqa: E402


def dev(rec):
    return torch.from_numpy(np.ascontiguousarray(rec).view(np.uint8).reshape(-1, 8)).cuda()


def timeit(fn, steps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="mpx,mpx_hot,gen1,gen1_b64")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--no-general", action="store_true")
    a = ap.parse_args()
    K, win, nw = 8, 10_000, 8
    for name in a.only.split(","):
        if name in ("mpx", "mpx_hot", "gen1", "gen1_hot"):
            H, W, n = (720, 1280, 10_000_000) if name.startswith("mpx") else (240, 304, 1_000_000)
            rec = synth.to_dat8(synth.synth_events(1003, n, W, H, nw * win, hotspot=name.endswith("hot")))
            d = dev(rec)
            st = torch.full((H, W, 2, K), -6000.0, device="cuda")
            for fast in ([True] if a.no_general else [True, False]):
                ms = timeit(lambda: er.encode_taf_dat(d, (H, W), st, 0, win, nw, K, check=False, fast=fast), a.steps)
                alg = 8 * n + 2 * 4 * 2 * K * H * W + 2 * K * H * W
                print(f"{name:10s} {'fast' if fast else 'general':8s} {ms:8.4f} ms  {n / ms / 1e6:8.2f} Gev/s  {alg / ms / 1e6:8.1f} GB/s")
        elif name.startswith("gen1_b"):
            B = int(name[6:])
            H, W, n = 240, 304, 1_000_000
            recs = [synth.to_dat8(synth.synth_events(1005 + j, n, W, H, nw * win)) for j in range(B)]
            offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
            d = dev(np.concatenate(recs))
            st = torch.full((B, H, W, 2, K), -6000.0, device="cuda")
            ms = timeit(lambda: er.encode_taf_batch(d, offs, (H, W), st, 0, win, nw, K, check=False), a.steps)
            alg = B * (8 * n + 2 * 4 * 2 * K * H * W + 2 * K * H * W)
            print(f"{name:10s} {'fast':8s} {ms:8.4f} ms  {B * n / ms / 1e6:8.2f} Gev/s  {alg / ms / 1e6:8.1f} GB/s")


if __name__ == "__main__":
    main()
