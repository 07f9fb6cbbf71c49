"""The split-K reduction inside the convolution kernel (k_conv_mfma_sk: the split that arrives last at its tile's counter sums all
splits in split order and finishes the tile) against the two-launch form (partials + k_splitk_reduce): the SAME BITS.  The
two-launch form is reachable only through a knob of the developer build (FRLW_CONV_SK_INKERNEL=0): two child processes load
libfrlw_evd_dev.so and save the head tensor of the same forward."""
import os
import subprocess
import sys

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_in_kernel_reduction_equals_the_two_launch_form_bit_for_bit(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from frlw_evd_amd import _build
    dev_lib = _build.DEV_LIB if os.path.exists(_build.DEV_LIB) else _build.build_dev()  # (__graft_entry__.build() keeps it current)
    outs = []
    for knob in ("0", "1"):
        out = str(tmp_path / f"sk{knob}.pt")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sk_ab.py"), out],
                           # (the same split decisions in both forms: with the reduction in the kernel the library also splits
                           # contractions of 512 .. 1023, which the two-launch form leaves whole -- another summation order)
                           env=dict(os.environ, FRLW_LIB_PATH=dev_lib, FRLW_CONV_SK_INKERNEL=knob, FRLW_CONV_SPLIT_MIN_NK="32"),
                           capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]  # (the tool itself asserts run-to-run equality)
        outs.append(torch.load(out))
    assert outs[0].shape[0] == 32 and float(outs[0].abs().max()) > 0
    assert torch.equal(outs[0], outs[1])


def test_train_step_with_in_kernel_reduction_is_reproducible():
    """The train step's thin convolutions (forward and data gradient) reduce their splits inside the kernel too, through ONE
    counter buffer for all layers: two steps from the same state give the same loss and gradients bit for bit, and the counters
    are back at zero afterwards."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import copy
    import numpy as np
    from frlw_evd_amd import e2e
    from frlw_evd_amd.trainer import Trainer
    from frlw_evd_amd.yolox import train_ops
    m0 = e2e.build_model(in_channels=16, num_classes=2)
    rng = np.random.default_rng(7)
    x = torch.from_numpy(rng.integers(0, 256, size=(8, 16, 256, 320, 1, 1), dtype=np.uint8)).float().div(255).cuda()
    lab = torch.zeros(8, 80, 5, dtype=torch.float64)
    lab[:, 0] = torch.tensor([0, 100, 90, 60, 40.0])
    lab = lab.cuda()
    res = []
    for _ in range(2):
        m = copy.deepcopy(m0)
        tr = Trainer(m, global_batch=8, nodes=1, iters_per_epoch=10)
        loss, _ = tr.train_step(x, lab, 0)
        res.append((loss, [p.detach().clone() for p in m.parameters()]))
    assert res[0][0] == res[1][0]
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)
    cnt = train_ops._sk_counters(x.device)
    assert int(cnt.abs().sum()) == 0
