"""Training-mode BaseConv kernels (csrc/train_ops.hip) against torch autograd (fp32) of the same module:
forward, input gradient, weight / gamma / beta gradients, running statistics.  Tolerance L3 (1e-3, SURVEY.md
section 8c); the kernels are exact-f32 MFMA with float64 statistics, observed ~1e-5."""
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
])
def test_base_conv_train_vs_torch(B, Cin, H, W, Cout, k, stride):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from frlw_evd_amd.yolox.network_blocks import BaseConv
    from frlw_evd_amd.yolox import train_ops
    torch.manual_seed(B * 1000 + Cin)
    ref = BaseConv(Cin, Cout, k, stride, act="silu").cuda().train()
    with torch.no_grad():
        ref.bn.weight.uniform_(0.5, 1.5)
        ref.bn.bias.normal_(0, 0.2)
    mine = BaseConv(Cin, Cout, k, stride, act="silu").cuda().train()
    mine.load_state_dict(ref.state_dict())
    x = torch.randn(B, Cin, H, W, device="cuda")
    xr = x.clone().requires_grad_(True)
    xm = x.clone().requires_grad_(True)
    gy = torch.randn(B, Cout, (H + 2 * ((k - 1) // 2) - k) // stride + 1, (W + 2 * ((k - 1) // 2) - k) // stride + 1, device="cuda")
    yr = ref.act(ref.bn(ref.conv(xr)))          # torch autograd (MIOpen / ATen)
    yr.backward(gy)
    assert train_ops.eligible(xm, mine.conv, mine.bn, mine.act)
    ym = train_ops.base_conv_train(xm, mine.conv, mine.bn)
    ym.backward(gy)
    assert rel(ym, yr) <= TOL
    assert rel(xm.grad, xr.grad) <= TOL
    assert rel(mine.conv.weight.grad, ref.conv.weight.grad) <= TOL
    assert rel(mine.bn.weight.grad, ref.bn.weight.grad) <= TOL
    assert rel(mine.bn.bias.grad, ref.bn.bias.grad) <= TOL
    assert rel(mine.bn.running_mean, ref.bn.running_mean) <= TOL
    assert rel(mine.bn.running_var, ref.bn.running_var) <= TOL
    assert int(mine.bn.num_batches_tracked) == 1
    # observed accuracy (exact-f32 contraction, float64 statistics)
    assert rel(ym, yr) <= 5e-5 and rel(mine.conv.weight.grad, ref.conv.weight.grad) <= 2e-4
