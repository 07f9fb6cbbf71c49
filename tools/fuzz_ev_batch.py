#!/usr/bin/env python3
"""Randomised differential test of the batched Event Volume path (frlw_ev_encode_batch, csrc/taf_fast.hip) against the
general path (frlw_ev_encode, one call per sequence), which the test-suite pins to the oracle: the f32 volume and its uint8
form must agree bit for bit.  Random batches (1-64 label windows, empty ones included; few and many (sequence, tile) pairs:
segment split + kf_ev_sub vs the tile walk), frame shapes, bins, window lengths, own t_end per sequence, skew (hot spots,
single hot pixels), shuffled streams, events in front of the window (dropped) and exactly on its end.

    python tools/fuzz_ev_batch.py [cases] [seed]
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
    bad = skipped = 0
    for case in range(cases):
        B = int(rng.choice([1, 1, 2, 3, 5, 8, 16, 33, 64], p=[.15, .1, .15, .15, .1, .1, .1, .1, .05]))
        H = int(rng.integers(8, 260))
        W = int(rng.integers(8, 700)) if rng.random() < 0.7 else int(rng.integers(600, 1300))
        if B >= 16 and rng.random() < 0.5:
            H, W = min(H, 120), min(W, 160)
        bins = int(rng.choice([5, 5, 5, 1, 2, 3, 8]))
        win = int(rng.choice([1_000, 50_000, 250_000, 250_000, 777_777]))
        budget = 3_000_000 // B
        recs, ends = [], []
        for s in range(B):
            n = 0 if rng.random() < 0.1 else int(rng.integers(1, max(2, min(budget, 400_000))))
            t_end = int(rng.integers(win, 3_000_000))
            ev = synth.synth_events(int(rng.integers(1 << 30)), n, W, H, win, hotspot=bool(rng.random() < 0.3), t_offset=t_end - win + 1)
            if n and rng.random() < 0.1:  # one hot pixel
                m = rng.random(n) < 0.5
                ev["x"][m], ev["y"][m] = W // 3, H // 2
            if n and rng.random() < 0.2:  # exactly on the end of the window; some in front of it (dropped by the time filter)
                ev["t"][-1] = t_end
                k = int(rng.integers(0, max(1, n // 10)))
                ev["t"][:k] = np.maximum(0, ev["t"][:k] - win)
            if n and rng.random() < 0.12:  # not time-sorted
                perm = rng.permutation(n)
                ev = {k: v[perm] for k, v in ev.items()}
            recs.append(synth.to_dat8(ev))
            ends.append(t_end)
        offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
        # every third case: the opt-in tile walk (kf_ev_tile); the others: the partition mode at random
        er.TUNING = (_lib.FrlwTuning(taf_tile_walk=1) if case % 3 == 2 else
                     [_lib.FrlwTuning(chunk_major=int(rng.integers(-1, 2)), ev_lds_float_atomics=int(rng.integers(-1, 2))),
                      _lib.FrlwTuning(direct_bins=1, chunk_major=int(rng.integers(-1, 2)), ev_lds_float_atomics=int(rng.integers(-1, 2))),
                      _lib.FrlwTuning(direct_bins=0, chunk_major=int(rng.integers(-1, 2)))][int(rng.integers(0, 3))])
        try:
            out, u8 = er.encode_ev_batch(dev(np.concatenate(recs)), offs, (H, W), ends, win, bins, want_u8=True)
        except NotImplementedError:
            skipped += 1
            continue
        finally:
            er.TUNING = None
        for s in range(B):
            if len(recs[s]) == 0:
                ok = float(out[s].abs().sum()) == 0.0
            else:
                oj, uj = er.encode_ev_dat(dev(recs[s]), (H, W), ends[s], win, bins, want_u8=True)
                ok = bool(torch.equal(oj, out[s]) and torch.equal(uj, u8[s]))
            if not ok:
                bad += 1
                print(f"MISMATCH case {case} seq {s}: B={B} H={H} W={W} bins={bins} win={win} n={len(recs[s])}")
                break
    print(f"{cases} cases ({skipped} outside the path's shapes), {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
