#!/usr/bin/env python3
"""Randomised differential test of the fast TAF path (frlw_taf_encode_batch, csrc/taf_fast.hip) against the general path
(frlw_taf_encode, one call per sequence), which the test-suite pins to the oracle: state, f32 view and uint8 must agree
bit for bit.  Random batches (1-64 sequences, empty ones included), frame shapes, K, window counts / lengths, start times,
prior states, skew (hot spots, single hot pixels), time-sorted and shuffled streams, coordinate maps.

    python tools/fuzz_taf_fast.py [cases] [seed]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frlw_evd_amd import _lib, event_representation as er, synth  # noqa: E402


def dev(rec):
    if len(rec) == 0:
        return torch.empty((0, 8), dtype=torch.uint8, device="cuda")
    return torch.from_numpy(np.ascontiguousarray(rec).view(np.uint8).reshape(-1, 8).copy()).cuda()


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    bad = 0
    skipped = 0
    for case in range(cases):
        B = int(rng.choice([1, 1, 2, 3, 5, 8, 16, 64], p=[.2, .15, .2, .15, .1, .1, .07, .03]))
        H = int(rng.integers(8, 260))
        W = int(rng.integers(8, 700)) if rng.random() < 0.7 else int(rng.integers(600, 1300))
        if B >= 16:
            H, W = min(H, 120), min(W, 160)
        # every third case through the tile walk (kf_taf_tile; takes effect from 256 (sequence, tile) pairs on: make some)
        tile_walk = case % 3 == 2
        if tile_walk and rng.random() < 0.7:
            B, H, W = int(rng.choice([8, 16, 33])), int(rng.integers(150, 260)), int(rng.integers(250, 420))
        # the others: the partition mode at random (library's choice / sub-tile bins forced where the frame allows / tile bins)
        cmaj = int(rng.integers(-1, 2))  # the partition: library's choice / histogram + scans / chunk-major
        er.TUNING = _lib.FrlwTuning(taf_tile_walk=1) if tile_walk else [_lib.FrlwTuning(chunk_major=cmaj), _lib.FrlwTuning(direct_bins=1, chunk_major=cmaj), _lib.FrlwTuning(direct_bins=0, chunk_major=cmaj)][int(rng.integers(0, 3))]
        if not tile_walk and rng.random() < 0.15:  # the one-workgroup-per-CU scatter (big chunks), which only 6 M-event calls take by themselves
            er.TUNING = _lib.FrlwTuning(chunk_major=1, direct_bins=0, batches_per_wave=int(rng.choice([9, 12, 20])))
        K = int(rng.choice([8, 8, 8, 5, 4, 1, 7]))
        n_win = int(rng.choice([1, 2, 3, 8, 8, 13, 64]))
        win = int(rng.choice([1_000, 10_000, 10_000, 7_777, 50_000]))
        if n_win * win > 2_000_000:
            n_win = max(1, 2_000_000 // win)
        budget = 3_000_000 // B
        recs, starts = [], []
        for s in range(B):
            kind = rng.random()
            n = 0 if kind < 0.1 else int(rng.integers(1, max(2, min(budget, 400_000))))
            t0 = int(rng.integers(0, 3_000_000))
            ev = synth.synth_events(int(rng.integers(1 << 30)), n, W, H, n_win * win, hotspot=bool(rng.random() < 0.3), t_offset=t0)
            if n and rng.random() < 0.1:  # one hot pixel
                m = rng.random(n) < 0.5
                ev["x"][m], ev["y"][m] = W // 3, H // 2
            if n and rng.random() < 0.15:  # an event exactly on the end of the span and on window boundaries
                ev["t"][-1] = t0 + n_win * win
                ev["t"][0] = t0
            if n and rng.random() < 0.12:  # not time-sorted
                perm = rng.permutation(n)
                ev = {k: v[perm] for k, v in ev.items()}
            if n and rng.random() < 0.2:  # leave some windows empty
                w_idx = np.minimum((ev["t"] - t0) // win, n_win - 1)
                drop = int(rng.integers(0, n_win))
                keep = w_idx != drop
                ev = {k: v[keep] for k, v in ev.items()}
            recs.append(synth.to_dat8(ev))
            starts.append(t0)
        offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
        state0 = rng.uniform(-6000 if rng.random() < 0.3 else -40, 0, (B, H, W, 2, K)).astype(np.float32)
        flip = bool(rng.random() < 0.5)
        st = torch.from_numpy(state0).cuda()
        try:
            u8, view = er.encode_taf_batch(dev(np.concatenate(recs)), offs, (H, W), st, starts, win, n_win, K, want_view=True,
                                           flip_k=flip)
        except NotImplementedError:
            skipped += 1
            continue
        er.TUNING = None
        for s in range(B):
            sj = torch.from_numpy(state0[s]).cuda()
            uj, vj = er.encode_taf_dat(dev(recs[s]), (H, W), sj, starts[s], win, n_win, K, want_view=True, flip_k=flip, fast=False)
            if not (torch.equal(sj, st[s]) and torch.equal(vj, view[s]) and torch.equal(uj, u8[s])):
                bad += 1
                print(f"MISMATCH case {case} seq {s}: tile_walk={tile_walk} B={B} H={H} W={W} K={K} n_win={n_win} win={win} n={len(recs[s])} "
                      f"state={bool(torch.equal(sj, st[s]))} view={bool(torch.equal(vj, view[s]))} u8={bool(torch.equal(uj, u8[s]))}")
                break
    print(f"{cases} cases ({skipped} outside the fast path's shapes), {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
