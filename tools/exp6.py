import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frlw_evd_amd import e2e
src = e2e.SyntheticTafSource(32)
net = e2e.build_model(16, 2).eval()
idx = list(range(32))
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("encode_batch (batched) ms", t(lambda: src.encode_batch(idx)))
src.batched = False
print("encode_batch (per sample) ms", t(lambda: src.encode_batch(idx)))
src.batched = True
x = src.encode_batch(idx)
with torch.no_grad():
    print("detect ms", t(lambda: net(x)))
    print("raw fwd ms", t(lambda: net.engine().raw_outputs(x[..., 0, 0])))
