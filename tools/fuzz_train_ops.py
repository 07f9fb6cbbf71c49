"""Randomised check of the training-mode BaseConv kernels against torch autograd in FLOAT64 (run on the GPU box).

float64 on purpose: MIOpen's float32 BatchNorm backward is itself off by 7-12 % in dgamma / dbeta when H * W is odd
(measured here against float64: e.g. (1, 292, 21, 53) -> 128 channels), so float32 torch cannot be the judge."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from frlw_evd_amd.yolox.network_blocks import BaseConv
from frlw_evd_amd.yolox import train_ops

def rel(a, b): return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst, bad = 0.0, 0
for case in range(n_cases):
    B = int(rng.integers(1, 9)); Cin = 4 * int(rng.integers(1, 80)); Cout = 4 * int(rng.integers(1, 80))
    k = int(rng.choice([1, 3])); stride = int(rng.choice([1, 2])) if k == 3 else 1
    H = int(rng.integers(2, 70)); W = int(rng.integers(2, 70))
    torch.manual_seed(case)
    mine = BaseConv(Cin, Cout, k, stride, act="silu").cuda().train()
    with torch.no_grad():
        mine.bn.weight.uniform_(0.5, 1.5); mine.bn.bias.normal_(0, 0.2)
    ref = BaseConv(Cin, Cout, k, stride, act="silu").cuda().train(); ref.load_state_dict(mine.state_dict()); ref = ref.double()
    x = torch.randn(B, Cin, H, W, device="cuda")
    xr = x.double().requires_grad_(True); xm = x.clone().requires_grad_(True)
    yr = ref.act(ref.bn(ref.conv(xr)))
    gy = torch.randn(yr.shape, device="cuda")
    yr.backward(gy.double())
    ym = train_ops.base_conv_train(xm, mine.conv, mine.bn); ym.backward(gy)
    errs = [rel(ym.detach().double(), yr.detach()), rel(xm.grad.double(), xr.grad),
            rel(mine.conv.weight.grad.double(), ref.conv.weight.grad), rel(mine.bn.weight.grad.double(), ref.bn.weight.grad),
            rel(mine.bn.bias.grad.double(), ref.bn.bias.grad), rel(mine.bn.running_var.double(), ref.bn.running_var)]
    worst = max(worst, max(errs))
    if max(errs) > 1e-3:
        bad += 1
        print(f"MISMATCH B={B} Cin={Cin} Cout={Cout} k={k} s={stride} {H}x{W}: {errs}")
print(f"{n_cases} cases, {bad} over 1e-3, worst relative error {worst:.2e}")
sys.exit(1 if bad else 0)
