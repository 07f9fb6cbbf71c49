"""The N > 1 path on CPU: two processes, gloo backend, 127.0.0.1 rendezvous.  Covers the sharding rules and the
reductions bench.py uses (barrier-bracketed timing, MAX over ranks, whole-job throughput)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from frlw_evd_amd import dist as fd


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, lr = fd.init_from_env(backend="gloo")
    assert (r, w, lr) == (rank, world, rank)
    seqs = list(range(11))  # 11 sequences, every rank encodes its own (stand-in "encode" = a checksum)
    mine = fd.shard_round_robin(seqs, r, w)
    lo, hi = fd.shard_range(1001, r, w)
    fd.barrier_sync()
    seconds = 0.5 + 0.25 * r  # rank 1 is the slow one
    tmax, = fd.max_over_ranks([seconds])
    rate = fd.job_throughput(len(mine) * 1000, seconds)
    checksum = fd.sum_over_ranks([sum(s * s for s in mine)])[0]
    fd.barrier_sync()
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([len(mine), lo, hi, tmax, rate, checksum]))
    torch.distributed.destroy_process_group()


def test_two_process_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (np.load(tmp_path / f"r{i}.npy") for i in range(world))
    assert r0[0] + r1[0] == 11 and abs(r0[0] - r1[0]) <= 1           # every sequence exactly once
    assert (r0[1], r0[2], r1[1], r1[2]) == (0, 501, 501, 1001)       # contiguous, balanced, no gap
    assert r0[3] == r1[3] == 0.75                                    # MAX over ranks
    assert r0[4] == r1[4] == pytest.approx(11000 / 0.75)             # whole-job units / slowest rank
    assert r0[5] == r1[5] == sum(s * s for s in range(11))           # nothing lost, nothing twice


def _agree_worker(rank, world, port, out_dir, failing_rank):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    fd.init_from_env(backend="gloo")
    from frlw_evd_amd import event_representation as er
    assert er.FAST_PATH_ENABLED is True
    verdict = fd.agree_fast_path(local_ok=rank != failing_rank)
    np.save(os.path.join(out_dir, f"a{rank}.npy"), np.array([int(verdict), int(er.FAST_PATH_ENABLED)]))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("failing_rank", [-1, 1])
def test_ranks_agree_on_the_fast_path(tmp_path, failing_rank):
    """The LDS lane-order self-test runs per process and device; one rank on the general path beside fast ranks changes step
    times (everybody waits for it at the next collective), not results.  dist.agree_fast_path takes the MINIMUM of the ranks'
    verdicts: all ranks pass -> everybody keeps the fast paths; rank 1 fails -> NOBODY takes them (module switch on every rank)."""
    world = 2
    mp.spawn(_agree_worker, args=(world, _free_port(), str(tmp_path), failing_rank), nprocs=world, join=True)
    want = 1 if failing_rank < 0 else 0
    for r in range(world):
        assert np.load(tmp_path / f"a{r}.npy").tolist() == [want, want]


def test_fast_path_switch_sends_every_encoder_to_the_general_path(monkeypatch):
    """With the job-wide switch off the batched entry points refuse like a device that failed the self-test (before any device
    work: this runs without a GPU) and the single-stream wrappers stop asking for the fast path."""
    from frlw_evd_amd import event_representation as er
    monkeypatch.setattr(er, "FAST_PATH_ENABLED", False)
    st = torch.zeros((1, 8, 8, 2, 8))
    with pytest.raises(NotImplementedError):
        er.encode_taf_batch(torch.zeros((4, 8), dtype=torch.uint8), [0, 4], (8, 8), st, 0)
    with pytest.raises(NotImplementedError):
        er.encode_ev_batch(torch.zeros((4, 8), dtype=torch.uint8), [0, 4], (8, 8), 10, 10)
    assert fd.agree_fast_path(local_ok=True) is True and er.FAST_PATH_ENABLED is True  # one rank: its own verdict
    monkeypatch.setattr(er, "FAST_PATH_ENABLED", True)


def test_single_process_helpers():
    assert fd.shard_round_robin(range(5), 0, 1) == [0, 1, 2, 3, 4]
    assert fd.shard_range(10, 0, 1) == (0, 10)
    assert [fd.shard_range(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    assert fd.max_over_ranks([1.5, 2.0]) == [1.5, 2.0]
    assert fd.job_throughput(100, 2.0) == 50.0


def _ddp_worker(rank, world, port, out_dir, hook="default"):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), FRLW_DDP_HOOK=hook)
    torch.set_num_threads(3)
    fd.init_from_env(backend="gloo")
    from frlw_evd_amd.trainer import Trainer
    from frlw_evd_amd.yolox import build_yolox
    from frlw_evd_amd.yolox.model import recipe_state_dict
    from test_detector_cpu import detector_input, train_labels
    m = build_yolox(10, 2)
    m.load_state_dict(recipe_state_dict(m, seed=1004 + rank))  # different init: DDP broadcasts rank 0's weights
    tr = Trainer(m, global_batch=4, nodes=world, iters_per_epoch=10, ddp="flat" if hook == "flat" else True)   # per-GPU batch = 4 / 2
    assert tr.per_gpu_batch == 2
    if hook == "flat":  # no wrapper: rank 0's state broadcast by hand, one all-reduce of the flat gradient per step
        assert tr.model is m and tr._flat_ddp and tr._world == world
    else:
        assert tr.model.gradient_as_bucket_view and tr.model.static_graph  # dist.ddp_kwargs
    x, lab = detector_input(1005, 4, H=128, W=160), train_labels()
    lab[..., 1:] *= 0.5  # boxes for the 128 x 160 input
    lo, hi = fd.shard_range(4, rank, world)
    losses = [tr.train_step(x[lo:hi], lab[lo:hi], i)[0] for i in range(2)]
    w = m.head.cls_preds[0].bias.detach().double()
    rm = m.backbone.stem.conv.bn.running_mean.detach().double()
    np.save(os.path.join(out_dir, f"d{rank}_{hook}.npy"),
            np.array(losses + [float(w.sum()), float(w.abs().sum()), float(rm.sum())]))
    torch.distributed.destroy_process_group()


def test_ddp_train_step_two_ranks(tmp_path):
    """core/exp.py:386-391 + 292-303 with world_size 2: weights stay identical across ranks (broadcast at wrap,
    all-reduced gradients), BatchNorm statistics stay per rank (broadcast_buffers=False)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    world = 2
    mp.spawn(_ddp_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    d0, d1 = (np.load(tmp_path / f"d{i}_default.npy") for i in range(world))
    assert np.all(np.isfinite(d0)) and np.all(np.isfinite(d1))
    assert d0[2] == d1[2] and d0[3] == d1[3]      # parameters identical after two steps
    assert d0[4] != d1[4]                         # running_mean differs: each rank saw its own shard
    assert d0[0] != d1[0]                         # per-rank losses differ (different images)
    # the reduce-scatter + all-gather communication hook gives the same trajectory as DDP's all-reduce
    mp.spawn(_ddp_worker, args=(world, _free_port(), str(tmp_path), "rs_ag"), nprocs=world, join=True)
    h0, h1 = (np.load(tmp_path / f"d{i}_rs_ag.npy") for i in range(world))
    assert h0[2] == h1[2] and h0[3] == h1[3]
    assert np.allclose(h0, d0, rtol=1e-6, atol=0) and np.allclose(h1, d1, rtol=1e-6, atol=0)
    # the wrapper-free exchange of the two-graph DDP step (Trainer(ddp="flat"): broadcast_module_state + one all-reduce of the
    # flat gradient buffer; the GPU captures the halves around it as HIP graphs): the same trajectory again
    mp.spawn(_ddp_worker, args=(world, _free_port(), str(tmp_path), "flat"), nprocs=world, join=True)
    f0, f1 = (np.load(tmp_path / f"d{i}_flat.npy") for i in range(world))
    assert f0[2] == f1[2] and f0[3] == f1[3] and f0[4] != f1[4]
    assert np.allclose(f0, d0, rtol=1e-6, atol=0) and np.allclose(f1, d1, rtol=1e-6, atol=0)


# ---- bench.py --gpus N is its own launcher (the flag used to be dead) --------------------------------
def _run_bench(args, env_extra, timeout=240):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


def test_bench_gpus_flag_spawns_the_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts 2 fresh ranks that rendezvous (gloo here, RCCL on
    the GPU box) and rank 0's single JSON line comes back through the parent."""
    rc, line, err = _run_bench(["--gpus", "2", "--launch-only"], {"FRLW_DIST_BACKEND": "gloo"})
    assert rc == 0, err
    assert line["n_gpus"] == 2 and line["rank_sum"] == 3.0 and line["max_over_ranks"] == 1.5
    assert line["launched_by"] == "bench.py" and line["backend"] == "gloo"


def test_bench_eight_ranks_dry_launch():
    """The driver's SCALE run goes to 8 ranks (README.md:160-170: one process per GPU of an 8-GPU node): the launcher, the
    rendezvous on 127.0.0.1, the SUM / MAX reductions and rank 0's single JSON line with eight fresh ranks over gloo -- both
    as bench.py's own launcher and under torch.distributed.run, the form the driver uses -- so that a first real 8-GPU run
    cannot die before it measures anything."""
    import subprocess
    import sys
    rc, line, err = _run_bench(["--gpus", "8", "--launch-only"], {"FRLW_DIST_BACKEND": "gloo"}, timeout=600)
    assert rc == 0, err
    assert line["n_gpus"] == 8 and line["rank_sum"] == 36.0 and line["max_over_ranks"] == 7.5
    assert line["launched_by"] == "bench.py" and line["backend"] == "gloo"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FRLW_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "3",
                        "--warmup", "1", "--launch-only"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert '"n_gpus": 8' in p.stdout and '"rank_sum": 36.0' in p.stdout and '"launched_by": "external launcher"' in p.stdout
    assert sum(1 for ln in p.stdout.splitlines() if ln.startswith("{")) == 1  # ONE line, from rank 0


def test_bench_single_rank_line_has_one_gpu():
    rc, line, err = _run_bench(["--launch-only"], {})
    assert rc == 0, err
    assert line["n_gpus"] == 1 and line["launched_by"] == "external launcher"


def test_bench_launcher_reports_a_failed_rank():
    rc, line, err = _run_bench(["--gpus", "2", "--launch-only"], {"FRLW_DIST_BACKEND": "gloo", "FRLW_BENCH_TEST_FAIL_RANK": "1"})
    assert rc != 0 and "rank 1 exited with code 3" in err


def test_bench_under_an_external_launcher():
    """The driver's form: torch.distributed.run starts the ranks, bench.py --gpus N joins them (and refuses a mismatch)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FRLW_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2",
                        "--launch-only"], env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr
    assert '"n_gpus": 2' in p.stdout and '"launched_by": "external launcher"' in p.stdout
    rc, _, err = _run_bench(["--gpus", "2", "--launch-only"], {"WORLD_SIZE": "3", "RANK": "0"})
    assert rc != 0


def _mask_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    fd.init_from_env(backend="gloo")
    from frlw_evd_amd.event_representation import _or_reduce_window_masks
    masks = torch.from_numpy(np.array([[0b1011, 1 << 40, 0], [0b0100, 1 << 63, 0]][rank], dtype=np.uint64).view(np.int64))  # bit 63: the sign bit travels too
    _or_reduce_window_masks(masks, None)
    np.save(os.path.join(out_dir, f"m{rank}.npy"), masks.numpy())
    torch.distributed.destroy_process_group()


def test_window_masks_or_reduce_two_ranks(tmp_path):
    """The one exchange of the row-stripe sharding (event_representation.encode_taf_stripe): 64-bit window masks OR-ed over the
    ranks through a MAX all-reduce of their bits (RCCL has no bitwise reductions)."""
    world = 2
    mp.spawn(_mask_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want = np.array([0b1111, (1 << 40) | (1 << 63), 0], dtype=np.uint64).view(np.int64)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"m{r}.npy"), want)
