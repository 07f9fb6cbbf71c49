"""Training-mode BaseConv kernels (csrc/train_ops.hip) against torch autograd of the same module in FLOAT64:
forward, input gradient, weight / gamma / beta gradients, running statistics.  Tolerance L3 (1e-3, SURVEY.md
section 8c); both arithmetics of the contractions -- float32 MFMA (exact products, the default) and float32 products from
three bf16 MFMAs (FRLW_CONV_PRECISION=bf16x3) -- with float64 statistics, observed ~1e-6 / ~1e-5.

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
    (2, 32, 16, 20, 24, 3, 2),     # parity-grouped data gradient with Cout % 16 != 0: this layer keeps the float32 MFMA
])
@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
def test_base_conv_train_vs_torch(B, Cin, H, W, Cout, k, stride, precision, monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from frlw_evd_amd.yolox.network_blocks import BaseConv
    from frlw_evd_amd.yolox import train_ops
    monkeypatch.setenv("FRLW_CONV_PRECISION", precision)
    assert train_ops.conv_precision() == train_ops.PRECISIONS[precision]
    if (B, Cout, stride) == (2, 24, 2):
        assert train_ops.layer_precision(Cout, k, stride) == 0
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
    assert max(errs) <= (2e-5 if precision == "f32" else 5e-5), errs  # observed accuracy (float64 statistics)


@pytest.mark.parametrize("B,C,H,W,nc", [(2, 256, 8, 10, 2), (3, 128, 16, 20, 7), (1, 64, 5, 7, 1), (2, 512, 4, 5, 11), (64, 256, 8, 10, 2)])
def test_pred_level_vs_torch(B, C, H, W, nc):
    """csrc/pred_ops.hip (the three biased 1x1 prediction convolutions of a head level + their gradients) against the
    float64 torch modules; run twice: bit-identical (fixed summation order)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from frlw_evd_amd.yolox import train_ops
    torch.manual_seed(C + nc)
    convs = [torch.nn.Conv2d(C, n, 1).cuda() for n in (4, 1, nc)]
    reg = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    cls = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert train_ops.pred_eligible(reg, cls, *convs)
    gout = torch.randn(B, 5 + nc, H, W, device="cuda")

    def native():
        for t in (reg, cls, *[p for c in convs for p in c.parameters()]):
            t.grad = None
        out = train_ops.pred_level(reg, cls, *convs)
        out.backward(gout)
        return [out.detach().clone(), reg.grad.clone(), cls.grad.clone()] + [p.grad.clone() for c in convs for p in c.parameters()]
    got = native()
    again = native()
    for a, b in zip(got, again):
        assert torch.equal(a, b)
    r64, c64 = reg.detach().double().requires_grad_(True), cls.detach().double().requires_grad_(True)
    convs64 = [torch.nn.Conv2d(C, n, 1).cuda().double() for n in (4, 1, nc)]
    for c64m, c32m in zip(convs64, convs):
        c64m.load_state_dict({k: v.double() for k, v in c32m.state_dict().items()})
    ref = torch.cat([convs64[0](r64), convs64[1](r64), convs64[2](c64)], 1)
    ref.backward(gout.double())
    want = [ref.detach(), r64.grad, c64.grad] + [p.grad for c in convs64 for p in c.parameters()]
    for a, b in zip(got, want):
        assert a.shape == b.shape
        assert rel(a.double(), b) < TOL, (a.shape, rel(a.double(), b))


def test_focus_nhwc_matches_space_to_depth():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from frlw_evd_amd.yolox import train_ops
    from frlw_evd_amd.yolox.network_blocks import Focus
    x = torch.randn(3, 10, 32, 48, device="cuda")
    got = train_ops.focus_nhwc(x)
    assert got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, Focus.space_to_depth(x))


@pytest.mark.parametrize("B,C,H,W", [(2, 256, 8, 10), (3, 96, 16, 20), (1, 40, 5, 7), (64, 256, 8, 10)])
def test_spp_pools_vs_torch(B, C, H, W):
    """csrc/pred_ops.hip SPP pools (forward + gather backward) against torch's MaxPool2d + cat, incl. exact ties."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from frlw_evd_amd.yolox import train_ops
    torch.manual_seed(C)
    x0 = torch.randn(B, C, H, W, device="cuda")
    x0[:, : C // 4] = torch.round(x0[:, : C // 4] * 2) / 2   # many exact ties: the first maximum must win
    pools = [torch.nn.MaxPool2d(k, 1, k // 2) for k in (5, 9, 13)]
    xa = x0.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert train_ops.spp_pools_eligible(xa, pools)
    ya = train_ops.spp_pools(xa)
    xb = x0.clone().requires_grad_(True)
    yb = torch.cat([xb] + [m(xb) for m in pools], 1)
    assert torch.equal(ya, yb)
    g = torch.randn_like(yb)
    ya.backward(g)
    yb.backward(g)
    assert rel(xa.grad.double(), xb.grad.double()) < 1e-5
