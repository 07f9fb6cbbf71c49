"""Does replaying the detector plan from a HIP graph (torch.cuda.CUDAGraph capture of frlw_det_run) help?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from frlw_evd_amd.yolox import build_yolox
from frlw_evd_amd.yolox.model import recipe_state_dict
B = 32
m = build_yolox(10, 2); m.load_state_dict(recipe_state_dict(m)); m = m.cuda().eval()
x = torch.rand(B, 10, 256, 320, device="cuda")
eng = m.engine()
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ref = eng.raw_outputs(x).clone()
print(f"eager plan   {timeit(lambda: eng.raw_outputs(x)):.3f} ms")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    eng.raw_outputs(x)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    out = eng.raw_outputs(x)
print(f"graph replay {timeit(lambda: g.replay()):.3f} ms")
g.replay(); torch.cuda.synchronize()
print("same result:", bool(torch.equal(out, ref)))
