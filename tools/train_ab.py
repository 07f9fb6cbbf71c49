"""Same-box A/B of the train step (batch 64, 16-channel input, one HIP graph per step): run once per setting of an environment
knob -- e.g.  FRLW_TRAIN_FUSE=0 python tools/train_ab.py ; FRLW_TRAIN_FUSE=1 python tools/train_ab.py  -- and compare the
medians; prints the loss of the last step too (the fused blocks must not move it by a bit)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from frlw_evd_amd import e2e
from frlw_evd_amd.trainer import Trainer
B = int(os.environ.get("B", "64"))
m = e2e.build_model(in_channels=16, num_classes=2)
tr = Trainer(m, global_batch=B, nodes=1, iters_per_epoch=100, graph=True)
rng = np.random.default_rng(1005)
x = torch.from_numpy(rng.integers(0, 256, size=(B, 16, 256, 320, 1, 1), dtype=np.uint8)).float().div(255).cuda()
lab = torch.zeros(B, 80, 5, dtype=torch.float64)
lab[:, 0] = torch.tensor([0, 100, 90, 60, 40.0]); lab[:, 1] = torch.tensor([1, 220, 150, 50, 80.0])
lab = lab.cuda()
tr.train_step(x, lab, 0)
bx, bl = tr.input_buffers(); bx.copy_(x); bl.copy_(lab)
for i in range(3): tr.train_step(bx, bl, 1 + i)
torch.cuda.synchronize()
ts = []
for r in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(10): loss, _ = tr.train_step(bx, bl, 4 + 10 * r + i, sync=False)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10)
knobs = {k: v for k, v in os.environ.items() if k.startswith("FRLW_")}
print(f"{knobs}: train step {sorted(ts)[2]:.3f} ms {[round(t, 3) for t in ts]} loss {float(loss):.6f}")
