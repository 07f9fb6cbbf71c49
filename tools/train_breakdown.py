"""Where the train step's time goes (torch autograd path): backbone+neck+head-towers fwd, loss, bwd, Adam."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from frlw_evd_amd import e2e
B = int(os.environ.get("B", "32"))
m = e2e.build_model(in_channels=16, num_classes=2)
m.train()
opt = torch.optim.Adam(m.parameters(), lr=1e-4)
rng = np.random.default_rng(5)
x = torch.from_numpy(rng.integers(0, 256, size=(B, 16, 256, 320, 1, 1), dtype=np.uint8)).float().div(255).cuda()
lab = torch.zeros(B, 80, 5, dtype=torch.float64)
lab[:, 0] = torch.tensor([0, 100, 90, 60, 40.]); lab[:, 1] = torch.tensor([1, 220, 150, 50, 80.])
lab = lab.cuda()
def ev(): e = torch.cuda.Event(enable_timing=True); e.record(); return e
tot = np.zeros(5); n = 6
for it in range(n + 2):
    opt.zero_grad(set_to_none=True)
    t0 = ev()
    feats = m.neck(m.backbone(x[..., 0]))
    t1 = ev()
    loss = m.head(feats, lab, x[..., 0])[0]
    t2 = ev()
    (loss * 65536.0).backward()
    t3 = ev()
    opt.step()
    t4 = ev()
    torch.cuda.synchronize()
    w0 = time.time()
    if it >= 2:
        tot += [t0.elapsed_time(t1), t1.elapsed_time(t2), t2.elapsed_time(t3), t3.elapsed_time(t4), t0.elapsed_time(t4)]
tot /= n
print(f"B={B}: backbone+neck fwd {tot[0]:.2f} ms | head fwd + SimOTA loss {tot[1]:.2f} ms | backward {tot[2]:.2f} ms | Adam {tot[3]:.2f} ms | total {tot[4]:.2f} ms -> {B / tot[4] * 1e3:.0f} frames/s")
# head towers alone (no loss)
with torch.no_grad():
    m.eval()
m.train()
