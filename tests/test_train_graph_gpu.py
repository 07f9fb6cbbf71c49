"""The train step replayed as one HIP graph (Trainer(graph=True)) against the eager step: same kernels in the same order,
so losses and parameters must agree bit for bit; caches keyed on parameter versions must notice the replayed updates."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu


def _inputs(B, seed):
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.integers(0, 256, size=(B, 16, 128, 160, 1, 1)).astype(np.float32) / np.float32(255))
    lab = torch.zeros(B, 80, 5, dtype=torch.float64)
    lab[:, 0] = torch.tensor([0, 60.0 + seed, 50.0, 40.0, 30.0])
    lab[:, 1] = torch.tensor([1, 100.0, 90.0 - seed, 30.0, 50.0])
    return x.cuda(), lab.cuda()


def _trainer(graph, warmup_epochs=1):
    from frlw_evd_amd.trainer import Trainer
    from frlw_evd_amd.yolox import build_yolox
    from frlw_evd_amd.yolox.model import recipe_state_dict
    m = build_yolox(16, 2)
    m.load_state_dict(recipe_state_dict(m, seed=31))
    return Trainer(m.cuda(), global_batch=4, nodes=1, iters_per_epoch=4, max_epoch=10, warmup_epochs=warmup_epochs, graph=graph)


def test_implicit_capture_follows_the_eager_trajectory_at_full_rate():
    """``train_step`` on a ``graph=True`` trainer captures on first use; with ``warmup_epochs=0`` the very first batch trains at
    the full rate -- exactly once, as in the eager loop (it used to get four updates)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    batches = [_inputs(4, s) for s in range(3)]
    eager, graphed = _trainer(False, 0), _trainer(True, 0)
    le = [eager.train_step(x, lab, i)[0] for i, (x, lab) in enumerate(batches)]
    lg = [graphed.train_step(x, lab, i)[0] for i, (x, lab) in enumerate(batches)]
    assert graphed._graph is not None and le == lg, (le, lg)
    for (n, a), b in zip(eager.model.state_dict().items(), graphed.model.state_dict().values()):
        assert torch.equal(a, b), n
    for pe, pg in zip(eager.optimizer.param_groups[0]["params"], graphed.optimizer.param_groups[0]["params"]):
        se, sg = eager.optimizer.state[pe], graphed.optimizer.state[pg]
        assert float(se["step"]) == float(sg["step"]) == len(batches)
        assert torch.equal(se["exp_avg"], sg["exp_avg"]) and torch.equal(se["exp_avg_sq"], sg["exp_avg_sq"])


def test_graph_replay_equals_eager_steps():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    batches = [_inputs(4, s) for s in range(5)]
    eager, graphed = _trainer(False), _trainer(True)
    # the capture's warm-up steps do not train: parameters, BatchNorm buffers and Adam's state are restored in place, so the
    # graphed trainer follows the eager trajectory (and core/exp.py:292-303) on the same data stream from the first batch on
    assert graphed.capture(*batches[0], warmup=3) == 0
    eager.model.train()
    le, lg = [], []
    for i, (x, lab) in enumerate(batches):
        le.append(eager.train_step(x, lab, i)[0])
        lg.append(graphed.train_step(x, lab, i)[0])
    assert graphed._graph is not None
    assert le == lg, (le, lg)
    for (n, a), b in zip(eager.model.named_parameters(), graphed.model.parameters()):
        assert torch.equal(a, b), n
    for (n, a), b in zip(eager.model.named_buffers(), graphed.model.buffers()):
        assert torch.equal(a, b), n
    # deferred read-back: a device tensor that survives the next replay
    l0, _ = graphed.train_step(*batches[0], 5, sync=False)
    l1, _ = graphed.train_step(*batches[1], 6, sync=False)
    assert l0.is_cuda and float(l0) != float(l1)


def test_replayed_updates_reach_the_inference_engine():
    """The folded inference weights are cached per parameter version: a replay must invalidate them."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    tr = _trainer(True)
    x, lab = _inputs(4, 9)
    for i in range(6):  # capture + replays past the zero-rate first step of the warm-up schedule
        tr.train_step(x, lab, i)
    m = tr.model
    m.eval()
    with torch.no_grad():
        got = m.engine().raw_outputs(x[..., 0])
        m2 = _trainer(False).model
        m2.load_state_dict(m.state_dict())
        m2.eval()
        want = m2.engine().raw_outputs(x[..., 0])
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-6)


def test_batched_weight_layouts_equal_per_layer(monkeypatch):
    """layout_all_weights (one launch for all BaseConv weights, from the second step on) against the per-layer layout
    kernels: same operands, so the same parameters bit for bit after a few steps."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from frlw_evd_amd.yolox import train_ops
    batches = [_inputs(4, s) for s in range(4)]
    a = _trainer(False)
    calls = []
    real = train_ops.layout_all_weights
    monkeypatch.setattr(train_ops, "layout_all_weights", lambda m: calls.append(real(m)) or calls[-1])
    for i, (x, lab) in enumerate(batches):
        a.train_step(x, lab, i)
    assert calls[0] is False and all(calls[1:])  # no caches before the first forward, one batched launch per step after it
    monkeypatch.setattr(train_ops, "layout_all_weights", lambda m: False)
    b = _trainer(False)
    for i, (x, lab) in enumerate(batches):
        b.train_step(x, lab, i)
    for (n, p), q in zip(a.model.named_parameters(), b.model.parameters()):
        assert torch.equal(p, q), n
