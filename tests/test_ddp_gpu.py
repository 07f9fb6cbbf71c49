"""Train step under DistributedDataParallel with the native BaseConv kernels: two ranks, one GPU each over RCCL when
the box has two GPUs, otherwise both on the one GPU of the test box over gloo (same code path above the backend).
Every rank is a freshly spawned child process (nothing re-execs a process that has touched the GPU).
Checks that the custom autograd Function fires DDP's gradient hooks: after one backward both ranks hold the same
gradients = the mean of the per-rank gradients, and those equal a single-process run of the two half batches."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(rank, B=2):
    rng = np.random.default_rng(500 + rank)
    x = torch.from_numpy(rng.integers(0, 256, size=(B, 16, 128, 160, 1, 1)).astype(np.float32) / np.float32(255))
    lab = torch.zeros(B, 80, 5, dtype=torch.float64)
    lab[:, 0] = torch.tensor([0, 60.0 + 10 * rank, 50.0, 40.0, 30.0])
    lab[:, 1] = torch.tensor([1, 100.0, 90.0 - 5 * rank, 30.0, 50.0])
    return x, lab


def _build():
    from frlw_evd_amd.yolox import build_yolox
    from frlw_evd_amd.yolox.model import recipe_state_dict
    m = build_yolox(16, 2)
    m.load_state_dict(recipe_state_dict(m, seed=77))
    return m.cuda().train()


def _worker(rank, world, port, out_dir, hook="default"):
    two_gpus = torch.cuda.device_count() >= world  # device_count() does not initialise the GPU
    dev = rank if two_gpus else 0
    backend = "nccl" if two_gpus else "gloo"
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(dev), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), FRLW_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    from frlw_evd_amd import dist as fd
    from torch.nn.parallel import DistributedDataParallel
    fd.init_from_env(backend)
    torch.cuda.set_device(dev)
    ddp = DistributedDataParallel(_build(), device_ids=[dev], broadcast_buffers=False, **fd.ddp_kwargs())  # core/exp.py:391
    timed = fd.install_comm_hook(ddp, hook)
    x, lab = _inputs(rank)
    loss = ddp(x.cuda(), lab.cuda(), None, None)
    loss.backward()
    torch.cuda.synchronize()
    g = torch.cat([p.grad.flatten().double().cpu() for p in ddp.module.parameters()])
    np.save(os.path.join(out_dir, f"g{rank}.npy"), g.numpy())
    np.save(os.path.join(out_dir, f"l{rank}.npy"), np.array(float(loss.detach())))
    if hook == "timed":  # the timing hook is the default all-reduce plus device events around every bucket
        summary = timed.summary()
        assert summary["buckets"] >= 1 and summary["total_ms"] > 0 and sum(summary["bucket_bytes"]) > 0, summary
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("hook", ["default", "timed", "rs_ag"])
def test_two_ddp_ranks_native_train_ops(tmp_path, hook):
    """hook: DDP's own all-reduce, the same with per-bucket timing (what bench.py --gpus N reports), and the
    reduce-scatter + all-gather hook -- all three must deliver the averaged gradient."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), hook), nprocs=world, join=True)
    g0, g1 = (np.load(tmp_path / f"g{r}.npy") for r in range(world))
    assert np.array_equal(g0, g1)  # the all-reduced (averaged) gradient is the same on both ranks
    # single process: mean of the two per-rank gradients (each rank normalises its loss by its own foreground count)
    m = _build()
    want = None
    for r in range(world):
        m.zero_grad(set_to_none=True)
        x, lab = _inputs(r)
        loss = m(x.cuda(), lab.cuda(), None, None)
        assert float(loss.detach()) == pytest.approx(float(np.load(tmp_path / f"l{r}.npy")), rel=1e-5)
        loss.backward()
        g = torch.cat([p.grad.flatten().double().cpu() for p in m.parameters()]).numpy()
        want = g if want is None else want + g
        m.load_state_dict(_build().state_dict())  # undo the running-statistics update of this pass
    want = want / world
    assert np.abs(g0 - want).max() <= 1e-3 * np.abs(want).max()


def _trainer_worker(rank, world, port, out_dir, graph):
    two_gpus = torch.cuda.device_count() >= world
    dev = rank if two_gpus else 0
    backend = "nccl" if two_gpus else "gloo"
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(dev), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), FRLW_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    from frlw_evd_amd import dist as fd
    from frlw_evd_amd.trainer import Trainer
    fd.init_from_env(backend)
    torch.cuda.set_device(dev)
    m = _build()
    if rank == 1:  # the ranks build their models independently: rank 0's state must win
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.01)
    tr = Trainer(m, global_batch=2 * world, nodes=world, iters_per_epoch=4, max_epoch=10, warmup_epochs=0, local_rank=dev,
                 ddp=True, graph=graph)
    losses = []
    for i in range(3):
        x, lab = _inputs(rank + 10 * i)
        losses.append(tr.train_step(x.cuda(), lab.cuda(), i)[0])
    x, lab = _inputs(rank + 40, B=1)  # another shape: the eager form of the same step
    losses.append(tr.train_step(x.cuda(), lab.cuda(), 3)[0])
    torch.cuda.synchronize()
    mod = tr.model.module if hasattr(tr.model, "module") else tr.model
    if graph:
        assert not hasattr(tr.model, "module") and tr._graph is not None and len(tr._graph) == 5
    tag = "g" if graph else "e"
    np.save(os.path.join(out_dir, f"p{tag}{rank}.npy"), torch.cat([p.detach().flatten().cpu() for p in mod.parameters()]).numpy())
    np.save(os.path.join(out_dir, f"l{tag}{rank}.npy"), np.array(losses))
    torch.distributed.destroy_process_group()


def test_two_graph_ddp_step_equals_the_eager_ddp_step(tmp_path):
    """Trainer(ddp=True, graph=True) -- two HIP graphs around ONE all-reduce of a flat gradient buffer, no
    DistributedDataParallel wrapper -- against the DistributedDataParallel trainer on the same two ranks and batches:
    rank 0's initial state reaches rank 1, the ranks stay identical, and both forms land on the same parameters
    (two addends commute: divide-then-SUM gives the same bits whatever the message layout)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world = 2
    for graph in (False, True):
        mp.spawn(_trainer_worker, args=(world, _free_port(), str(tmp_path), graph), nprocs=world, join=True)
    pe = [np.load(tmp_path / f"pe{r}.npy") for r in range(world)]
    pg = [np.load(tmp_path / f"pg{r}.npy") for r in range(world)]
    assert np.array_equal(pe[0], pe[1]) and np.array_equal(pg[0], pg[1])
    for r in range(world):
        le, lg = np.load(tmp_path / f"le{r}.npy"), np.load(tmp_path / f"lg{r}.npy")
        assert np.allclose(le, lg, rtol=1e-6, atol=0), (r, le, lg)
    assert np.abs(pe[0] - pg[0]).max() <= 1e-6 * np.abs(pe[0]).max()
    fresh = torch.cat([p.detach().flatten().cpu() for p in _build().parameters()]).numpy()
    assert np.abs(pg[0] - fresh).max() > 0  # it trained


def _stripe_worker(rank, world, port, out_dir):
    two_gpus = torch.cuda.device_count() >= world
    dev = rank if two_gpus else 0
    backend = "nccl" if two_gpus else "gloo"
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(dev), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), FRLW_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    from frlw_evd_amd import dist as fd, event_representation as er, synth
    fd.init_from_env(backend)
    torch.cuda.set_device(dev)
    H, W, K, win, n_win = 240, 304, 8, 10_000, 8
    ev = synth.synth_events(77, 400_000, W, H, n_win * win)
    w_idx = np.minimum(ev["t"] // win, n_win - 1)
    keep = ~((w_idx == 3) & (ev["y"] < H // 2)) & (w_idx != 6)   # window 3 only in the lower half, window 6 nowhere
    rec = synth.to_dat8({k: v[keep] for k, v in ev.items()})
    dat = torch.from_numpy(np.ascontiguousarray(rec).view(np.uint8).reshape(-1, 8).copy()).cuda()
    lo, hi = fd.shard_range(H, rank, world)                     # this rank's rows
    st = torch.full((1, hi - lo, W, 2, K), -6000.0, device="cuda")
    if backend == "gloo":  # gloo reduces host tensors: the 8-byte masks make the round trip (RCCL: on the device)
        def exchange(m):
            h = m.cpu()
            parts = [torch.zeros_like(h) for _ in range(world)]
            torch.distributed.all_gather(parts, h)
            for p in parts:
                h |= p
            m.copy_(h)
        er.encode_taf_stripe(dat, [0, len(rec)], (H, W), (lo, hi), st, 0, win, n_win, K, exchange=exchange)
    else:
        er.encode_taf_stripe(dat, [0, len(rec)], (H, W), (lo, hi), st, 0, win, n_win, K)
    np.save(os.path.join(out_dir, f"s{rank}.npy"), st.cpu().numpy())
    torch.distributed.destroy_process_group()


def test_row_stripe_sharding_two_ranks(tmp_path):
    """One stream, two ranks, each encodes its half of the rows; the window masks are OR-reduced between the two halves of
    the encode (RCCL when the box has two GPUs, else both ranks on GPU 0 over gloo): halves together == whole frame."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from frlw_evd_amd import event_representation as er, synth
    world = 2
    mp.spawn(_stripe_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    H, W, K, win, n_win = 240, 304, 8, 10_000, 8
    ev = synth.synth_events(77, 400_000, W, H, n_win * win)
    w_idx = np.minimum(ev["t"] // win, n_win - 1)
    keep = ~((w_idx == 3) & (ev["y"] < H // 2)) & (w_idx != 6)
    rec = synth.to_dat8({k: v[keep] for k, v in ev.items()})
    dat = torch.from_numpy(np.ascontiguousarray(rec).view(np.uint8).reshape(-1, 8).copy()).cuda()
    full = torch.full((1, H, W, 2, K), -6000.0, device="cuda")
    er.encode_taf_batch(dat, [0, len(rec)], (H, W), full, 0, win, n_win, K)
    got = np.concatenate([np.load(tmp_path / f"s{r}.npy") for r in range(world)], axis=1)
    assert got.shape == (1, H, W, 2, K) and got.tobytes() == full.cpu().numpy().tobytes()
