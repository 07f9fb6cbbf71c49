import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from frlw_evd_amd import e2e
B = 64
m = e2e.build_model(in_channels=16, num_classes=2); m.train()
opt = torch.optim.Adam(m.parameters(), lr=1e-4)
rng = np.random.default_rng(5)
x = torch.from_numpy(rng.integers(0, 256, size=(B, 16, 256, 320, 1, 1), dtype=np.uint8)).float().div(255).cuda()
lab = torch.zeros(B, 80, 5, dtype=torch.float64); lab[:, 0] = torch.tensor([0, 100, 90, 60, 40.]); lab[:, 1] = torch.tensor([1, 220, 150, 50, 80.]); lab = lab.cuda()
def step():
    opt.zero_grad(set_to_none=True)
    loss = m.head(m.neck(m.backbone(x[..., 0])), lab, x[..., 0])[0]
    (loss * 65536.0).backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.device_time_total > 0 and e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:22]:
    print(f"{e.self_device_time_total/1e3:8.3f} ms  x{e.count:4d}  {e.key:32s} {str(e.input_shapes)[:110]}")
