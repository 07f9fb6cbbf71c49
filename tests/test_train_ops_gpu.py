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


def _grads_of(module, x, dy_wide, lo):
    """Forward + backward of ``module`` on a fresh leaf copy of x; the upstream gradient arrives as a channel slice of a wider
    channels_last tensor (what the backward of a concatenation hands out)."""
    xl = x.clone().requires_grad_(True)
    y = module(xl)
    y.backward(dy_wide[:, lo:lo + y.shape[1]])
    torch.cuda.synchronize()
    grads = {n: p.grad.clone() for n, p in module.named_parameters()}
    bufs = {n: b.clone() for n, b in module.named_buffers()}
    return y.detach().clone(), xl.grad.clone(), grads, bufs


@pytest.mark.parametrize("block", ["bottleneck", "csp", "csp_noshortcut"])
def test_fused_blocks_equal_the_unfused_sequence(block, monkeypatch):
    """Bottleneck with shortcut as ONE autograd node (shortcut added by the pass that writes conv2's activation, its gradient by
    the epilogue of conv1's data gradient) and conv1 | conv2 of a CSPLayer as one node (dx of the branches summed in an epilogue)
    against every BaseConv as its own node with torch's add / gradient accumulation in between (FRLW_TRAIN_FUSE=0): the same IEEE
    additions, so outputs, input gradient, every parameter gradient and the running statistics must be EQUAL."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import copy
    from frlw_evd_amd.yolox.network_blocks import Bottleneck, CSPLayer
    torch.manual_seed(5)
    C = 64
    if block == "bottleneck":
        m = Bottleneck(C, C, shortcut=True, expansion=1.0)
    else:
        m = CSPLayer(C, C, n=2, shortcut=block == "csp")
    for p in m.parameters():
        if p.dim() == 1:
            torch.nn.init.uniform_(p, 0.5, 1.5)
    m = m.cuda().train()
    ref = copy.deepcopy(m)
    start = copy.deepcopy(m.state_dict())
    x = torch.randn(3, C, 16, 20, device="cuda").contiguous(memory_format=torch.channels_last)
    dy_wide = torch.randn(3, C + 32, 16, 20, device="cuda").contiguous(memory_format=torch.channels_last)
    monkeypatch.setenv("FRLW_TRAIN_FUSE", "1")
    monkeypatch.setenv("FRLW_TRAIN_STACK", "0")  # (the stacked pair is compared to rounding below: its sums run in another order)
    probe = m(x.clone().requires_grad_(True))  # (a forward of its own: which autograd nodes does the fused module build?)
    names, todo = set(), [probe.grad_fn]
    while todo:
        fn = todo.pop()
        if fn is not None and fn not in names:
            names.add(fn)
            todo += [f for f, _ in fn.next_functions]
    names = {type(f).__name__ for f in names}
    want = {"bottleneck": {"_BottleneckTrainBackward"}, "csp": {"_BottleneckTrainBackward", "_PairTrainBackward", "_JoinSlicesBackward"},
            "csp_noshortcut": {"_PairTrainBackward", "_JoinSlicesBackward"}}[block]
    assert want <= names and not any("Cat" in n or "Add" in n for n in names), names
    del probe
    m.load_state_dict(ref.state_dict())  # undo the running-statistics update of the probe
    y1, dx1, g1, b1 = _grads_of(m, x, dy_wide, 16)
    monkeypatch.setenv("FRLW_TRAIN_FUSE", "0")
    y0, dx0, g0, b0 = _grads_of(ref, x, dy_wide, 16)
    assert torch.equal(y1, y0) and torch.equal(dx1, dx0)
    for n in g0:
        assert torch.equal(g1[n], g0[n]), n
    for n in b0:
        assert torch.equal(b1[n], b0[n]), n
    # and both agree with torch autograd of the same module in float64
    monkeypatch.setenv("FRLW_NATIVE_TRAIN", "0")
    x64 = x.double().clone().requires_grad_(True)
    fresh = Bottleneck(C, C, shortcut=True, expansion=1.0) if block == "bottleneck" else CSPLayer(C, C, n=2, shortcut=block == "csp")
    fresh = fresh.cuda().double().train()
    fresh.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in m.state_dict().items()}, strict=True)
    y64 = fresh(x64)
    y64.backward(dy_wide[:, 16:16 + C].double())
    assert rel(y1.double(), y64.detach()) < TOL and rel(dx1.double(), x64.grad) < TOL
    for n, p in fresh.named_parameters():
        assert rel(g1[n].double(), p.grad) < TOL, n
    if block == "bottleneck":
        return
    # conv1 | conv2 as ONE stacked block (_PairStackTrain: one convolution, one BatchNorm pass each way, one data and one weight
    # gradient for both): the same numbers to rounding -- the contraction and the float64 statistics sums run over other tiles
    monkeypatch.setenv("FRLW_NATIVE_TRAIN", "1")
    monkeypatch.setenv("FRLW_TRAIN_FUSE", "1")
    monkeypatch.setenv("FRLW_TRAIN_STACK", "1")
    st = copy.deepcopy(ref)
    st.load_state_dict(start)
    probe = st(x.clone().requires_grad_(True))
    names, todo = set(), [probe.grad_fn]
    while todo:
        fn = todo.pop()
        if fn is not None and fn not in names:
            names.add(fn)
            todo += [f for f, _ in fn.next_functions]
    assert "_PairStackTrainBackward" in {type(f).__name__ for f in names}
    del probe
    st.load_state_dict(start)
    y2, dx2, g2, b2 = _grads_of(st, x, dy_wide, 16)
    assert rel(y2, y1) < 1e-5 and rel(dx2, dx1) < 1e-5
    for n in g1:
        assert rel(g2[n], g1[n]) < 2e-5, n
    for n in b1:
        if b1[n].is_floating_point():
            assert rel(b2[n], b1[n]) < 1e-5, n
        else:
            assert torch.equal(b2[n], b1[n]), n
    for n, p in fresh.named_parameters():
        assert rel(g2[n].double(), p.grad) < TOL, n


def test_fused_train_step_equals_unfused(monkeypatch):
    """The whole detector: loss and every parameter gradient of one step with the fused blocks (Bottleneck nodes, CSP / head pairs)
    equal the unfused step's bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from frlw_evd_amd.yolox import build_yolox
    from frlw_evd_amd.yolox.model import recipe_state_dict
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.integers(0, 256, size=(2, 16, 128, 160, 1, 1)).astype(np.float32) / np.float32(255)).cuda()
    lab = torch.zeros(2, 80, 5, dtype=torch.float64)
    lab[:, 0] = torch.tensor([0, 60.0, 50.0, 40.0, 30.0])
    lab[:, 1] = torch.tensor([1, 100.0, 90.0, 30.0, 50.0])
    lab = lab.cuda()
    out = {}
    monkeypatch.setenv("FRLW_TRAIN_STACK", "0")
    for fuse in ("1", "0", "stack"):
        monkeypatch.setenv("FRLW_TRAIN_FUSE", "0" if fuse == "0" else "1")
        monkeypatch.setenv("FRLW_TRAIN_STACK", "1" if fuse == "stack" else "0")
        m = build_yolox(16, 2)
        m.load_state_dict(recipe_state_dict(m, seed=12))
        m = m.cuda().train()
        loss = m(x, lab, None, None)
        loss.backward()
        torch.cuda.synchronize()
        out[fuse] = (float(loss.detach()), {n: p.grad.clone() for n, p in m.named_parameters()},
                     {n: b.clone() for n, b in m.named_buffers()})
    assert out["1"][0] == out["0"][0]
    for n, g in out["0"][1].items():
        assert torch.equal(out["1"][1][n], g), n
    for n, b in out["0"][2].items():
        assert torch.equal(out["1"][2][n], b), n
    # ... and with the pairs as stacked blocks: the same step to rounding
    assert out["stack"][0] == pytest.approx(out["1"][0], rel=1e-6)
    for n, g in out["1"][1].items():
        assert rel(out["stack"][1][n], g) < 1e-3, n   # (single gradients of a 65536x scaled loss pass through SimOTA's choices: 1e-3 like every train test)
    for n, b in out["1"][2].items():
        if b.is_floating_point():
            assert rel(out["stack"][2][n], b) < 1e-5, n
