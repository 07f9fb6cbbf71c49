"""Per-phase cycle breakdown of k_taf_tile (needs a -DFRLW_TILE_PROF build)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from frlw_evd_amd import synth, event_representation as er, _lib
lib = _lib.load()
lib.frlw_debug_prof.argtypes = [C.c_void_p, C.c_int]
names = ["prologue+closes", "count", "scan", "place", "order", "walk", "tail sync", "final closes", "write-out"]
for (H, W, n) in ((240, 304, 1_000_000), (720, 1280, 10_000_000)):
    ev = synth.synth_events(1005, n, W, H, 80_000)
    dat = torch.from_numpy(synth.to_dat8(ev).view(np.uint8).reshape(-1, 8).copy()).cuda()
    st = torch.full((H, W, 2, 8), -6000.0, device="cuda")
    for _ in range(3): er.encode_taf_dat(dat, (H, W), st, 0, 10_000, 8, 8, check=False)
    torch.cuda.synchronize()
    lib.frlw_debug_prof(None, 1)
    reps = 10
    for _ in range(reps): er.encode_taf_dat(dat, (H, W), st, 0, 10_000, 8, 8, check=False)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 16)()
    lib.frlw_debug_prof(out, 0)
    twl = 8 if W > 512 else 6
    tiles = ((W + (1 << twl) // 1 - 1) >> twl)  # only for the label; averages use the per-tile sum / reps
    tot = sum(out[i] for i in range(9))
    print(f"{W}x{H} n={n}: total cycles (sum over tiles, per encode) {tot / reps:.0f}")
    for i, nm in enumerate(names):
        print(f"  {nm:18s} {out[i] / reps:14.0f}  {100.0 * out[i] / tot:5.1f}%")
