"""Stress of the in-kernel split-K reduction: many forwards at several batch sizes (different split counts and tile counts), every
head tensor compared bit for bit with the first one of its batch size; a second engine (own scratch, own counters) interleaved
on a second stream.  Any stale read of another workgroup's partial tile would show up as a mismatch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from frlw_evd_amd.yolox import build_yolox
from frlw_evd_amd.yolox.model import recipe_state_dict
from frlw_evd_amd.detector import DetectorEngine
m = build_yolox(10, 2); m.load_state_dict(recipe_state_dict(m, seed=1004)); m.eval()
e1, e2 = DetectorEngine(m), DetectorEngine(m)
s2 = torch.cuda.Stream()
bad = 0
for B in (32, 8, 3, 17, 1):
    x = torch.from_numpy(np.random.default_rng(B).integers(0, 256, size=(B, 10, 256, 320)).astype(np.float32) / np.float32(255)).cuda()
    ref = e1.raw_outputs(x).clone()
    torch.cuda.synchronize()
    for it in range(120):
        a = e1.raw_outputs(x)
        with torch.cuda.stream(s2):
            b = e2.raw_outputs(x)
        torch.cuda.synchronize()
        if not torch.equal(a, ref) or not torch.equal(b, ref):
            bad += 1
    print(f"B={B}: 240 forwards on two engines / two streams, mismatches so far {bad}", flush=True)
print("sk stress:", "OK" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
