"""Training-mode BaseConv kernels (csrc/train_ops.hip) against torch autograd of the same module in FLOAT64:
forward, input gradient, weight / gamma / beta gradients, running statistics.  Tolerance L3 (1e-3, SURVEY.md
section 8c); the kernels are exact-f32 MFMA with float64 statistics, observed ~1e-6.

The judge is float64 because MIOpen's float32 BatchNorm backward is itself off by 7-12 % in dgamma / dbeta when
H * W is odd (measured against float64, tools/fuzz_train_ops.py) -- not a regime of the detector (H, W are multiples of
32), but the randomised shapes below include it."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu
TOL = 1e-3


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,stride", [
    (2, 64, 32, 40, 32, 3, 1),     # stem-like
    (3, 32, 32, 40, 64, 3, 2),     # stride-2 downsample
    (2, 128, 16, 20, 128, 1, 1),   # 1x1
    (2, 256, 8, 10, 256, 3, 1),    # small map (split-K forward, many wgrad splits)
    (1, 40, 18, 22, 48, 3, 2),     # odd sizes, channels not a multiple of 32
    (4, 16, 9, 7, 20, 3, 1),
    (2, 24, 15, 13, 28, 3, 2),     # stride 2 on odd sizes: transposed-gather data gradient instead of parity classes
    (1, 292, 21, 53, 128, 1, 1),   # H * W odd: the shape where float32 MIOpen BatchNorm backward is wrong
])
def test_base_conv_train_vs_torch(B, Cin, H, W, Cout, k, stride):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from frlw_evd_amd.yolox.network_blocks import BaseConv
    from frlw_evd_amd.yolox import train_ops
    torch.manual_seed(B * 1000 + Cin)
    mine = BaseConv(Cin, Cout, k, stride, act="silu").cuda().train()
    with torch.no_grad():
        mine.bn.weight.uniform_(0.5, 1.5)
        mine.bn.bias.normal_(0, 0.2)
    ref = BaseConv(Cin, Cout, k, stride, act="silu").cuda().train()
    ref.load_state_dict(mine.state_dict())
    ref = ref.double()
    x = torch.randn(B, Cin, H, W, device="cuda")
    xr = x.double().requires_grad_(True)
    xm = x.clone().requires_grad_(True)
    gy = torch.randn(B, Cout, (H + 2 * ((k - 1) // 2) - k) // stride + 1, (W + 2 * ((k - 1) // 2) - k) // stride + 1, device="cuda")
    yr = ref.act(ref.bn(ref.conv(xr)))          # torch autograd, float64 (ATen kernels)
    yr.backward(gy.double())
    assert train_ops.eligible(xm, mine.conv, mine.bn, mine.act)
    ym = train_ops.base_conv_train(xm, mine.conv, mine.bn)
    ym.backward(gy)
    d = lambda t: t.detach().double()
    errs = [rel(d(ym), d(yr)), rel(d(xm.grad), xr.grad), rel(d(mine.conv.weight.grad), ref.conv.weight.grad),
            rel(d(mine.bn.weight.grad), ref.bn.weight.grad), rel(d(mine.bn.bias.grad), ref.bn.bias.grad),
            rel(d(mine.bn.running_mean), ref.bn.running_mean), rel(d(mine.bn.running_var), ref.bn.running_var)]
    assert max(errs) <= TOL, errs
    assert int(mine.bn.num_batches_tracked) == 1
    assert max(errs) <= 2e-5, errs  # observed accuracy (exact-f32 contraction, float64 statistics)
