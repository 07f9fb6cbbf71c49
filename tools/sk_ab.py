"""A/B of the in-kernel split-K reduction (developer library: FRLW_CONV_SK_INKERNEL=0|1 python tools/sk_ab.py out.pt): forward time,
and the head tensor saved for a bitwise comparison between the two forms."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from frlw_evd_amd.yolox import build_yolox
from frlw_evd_amd.yolox.model import recipe_state_dict
B = 32
m = build_yolox(10, 2); m.load_state_dict(recipe_state_dict(m, seed=1004)); m.eval()
x = torch.from_numpy(np.random.default_rng(5).integers(0, 256, size=(B, 10, 256, 320)).astype(np.float32) / np.float32(255)).cuda()
eng = m.engine()
a = eng.raw_outputs(x).clone()
b = eng.raw_outputs(x).clone()
assert torch.equal(a, b), "run-to-run"
for _ in range(10): eng.raw_outputs(x)
torch.cuda.synchronize()
ts = []
for r in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): eng.raw_outputs(x)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 30)
print(f"SK_INKERNEL={os.environ.get('FRLW_CONV_SK_INKERNEL', 'default')}: forward {sorted(ts)[1]:.4f} ms {[round(t, 4) for t in ts]}")
if len(sys.argv) > 1: torch.save(a.cpu(), sys.argv[1])
