"""Fuzz of the train step's block nodes (yolox/train_ops.py: _BottleneckTrain, _PairTrain, destination slices) on random shapes:
a Bottleneck / CSPLayer with FRLW_TRAIN_FUSE=1 against the same module with every BaseConv as its own autograd node
(FRLW_TRAIN_FUSE=0) -- output, input gradient, every parameter gradient and every running statistic must be EQUAL (the fused
additions are the same IEEE additions); and a CSPLayer whose conv1 | conv2 run as ONE stacked block (FRLW_TRAIN_STACK=1) against
the two-block pair: equal to rounding (1e-4 of the largest value; the sums run over other tiles).   python tools/fuzz_train_blocks.py [cases] [seed]"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from frlw_evd_amd.yolox.network_blocks import Bottleneck, CSPLayer

def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def run(module, x, dy_wide, lo, fuse, stack="0"):
    os.environ["FRLW_TRAIN_FUSE"] = fuse
    os.environ["FRLW_TRAIN_STACK"] = stack
    xl = x.clone().requires_grad_(True)
    y = module(xl)
    y.backward(dy_wide[:, lo:lo + y.shape[1]])
    torch.cuda.synchronize()
    return (y.detach().clone(), xl.grad.clone(), {n: p.grad.clone() for n, p in module.named_parameters()},
            {n: b.clone() for n, b in module.named_buffers()})

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for it in range(cases):
    C = int(rng.choice([8, 16, 24, 40, 64, 96, 128, 256, 512]))
    Cout = C if rng.random() < 0.6 else int(rng.choice([16, 32, 64, 128, 256]))
    B = int(rng.integers(1, 9)) if C <= 128 else int(rng.integers(1, 4))
    H, W = int(rng.integers(3, 41)), int(rng.integers(3, 49))
    if rng.random() < 0.2:  # a large map: the wide tiles and the unsplit kernels
        B, H, W = int(rng.integers(8, 33)), 32, 40
        C = min(C, 128); Cout = min(Cout, 128)
    kind = rng.choice(["bottleneck", "csp", "csp_plain"])
    torch.manual_seed(int(rng.integers(1 << 30)))
    if kind == "bottleneck":
        m = Bottleneck(C, C, shortcut=True, expansion=float(rng.choice([0.5, 1.0])))
        Cout = C
    else:
        m = CSPLayer(C, Cout, n=int(rng.integers(1, 4)), shortcut=kind == "csp")
    for p in m.parameters():
        if p.dim() == 1:
            torch.nn.init.uniform_(p, 0.5, 1.5)
    m = m.cuda().train()
    ref = copy.deepcopy(m)
    st = copy.deepcopy(m)
    x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    pad = int(rng.choice([0, 4, 32]))
    dy_wide = torch.randn(B, Cout + pad, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    lo = pad // 2 // 4 * 4
    a = run(m, x, dy_wide, lo, "1")
    b = run(ref, x, dy_wide, lo, "0")
    ok = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and all(torch.equal(a[2][n], b[2][n]) for n in b[2]) \
        and all(torch.equal(a[3][n], b[3][n]) for n in b[3])
    if not ok:
        bad += 1
        print(f"MISMATCH case {it}: {kind} C={C} Cout={Cout} B={B} H={H} W={W} pad={pad}")
    if kind != "bottleneck":  # conv1 | conv2 as ONE stacked block: the same numbers to rounding (other summation tiles)
        c = run(st, x, dy_wide, lo, "1", "1")
        worst = max([rel(c[0], a[0]), rel(c[1], a[1])] + [rel(c[2][n], a[2][n]) for n in a[2]]
                    + [rel(c[3][n], a[3][n]) for n in a[3] if a[3][n].is_floating_point()])
        if not worst < 1e-4:
            bad += 1
            print(f"STACK MISMATCH case {it}: {kind} C={C} Cout={Cout} B={B} H={H} W={W} pad={pad}: {worst:.3g}")
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
