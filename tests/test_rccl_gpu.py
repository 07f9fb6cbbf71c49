"""RCCL on the box we have: a one-rank ``nccl`` process group executes every collective call an N-rank job makes
(until this test nothing in the repo had ever loaded librccl -- the 2-rank tests fall back to gloo on
a one-GPU box).  Fresh child processes only; nothing re-execs a process that touched the GPU.

Reference: train.py:31 (``init_process_group('nccl')``, also with one process), core/exp.py:391 (DDP)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("FRLW_DIST_BACKEND", None)
    return env


def test_every_collective_on_rccl_world_1(tmp_path):
    torch = pytest.importorskip("torch")
    if torch.cuda.device_count() < 1:
        pytest.skip("no GPU")
    out = tmp_path / "rccl.json"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_child.py"), str(out)], env=_env(),
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    res = json.loads(out.read_text())
    assert res["max"] == [1.5, 2.5] and res["sum"] == [3.0] and res["rate"] == 5.0
    assert res["rs_ag_roundtrip"]
    for hook in ("default", "timed", "rs_ag"):
        h = res["hooks"][hook]
        assert h["losses_equal"], (hook, h["losses"], res["losses_bare"])   # world 1: sum / 1 -- bit for bit
        assert h["params_equal"], hook
    assert res["hooks"]["timed"]["buckets"] >= 1 and res["hooks"]["timed"]["total_ms"] > 0
    g2 = res["graph2"]  # two HIP graphs around one RCCL all-reduce, no DistributedDataParallel wrapper
    assert g2["two_graphs"] and not g2["wrapped"] and g2["losses_equal"] and g2["params_equal"] and g2["odd_batch_loss_equal"], g2
    assert res["stripe_equals_whole"] and res["mask_reduce_identity"]
    assert res["destroyed"]
    assert any("rccl" in lib.lower() for lib in res["libs"]), res["libs"]       # the collectives ran in librccl
    assert any("libfrlw_evd" in lib for lib in res["libs"]), res["libs"]


def test_bench_launch_only_on_rccl_world_1():
    """``bench.py --gpus 1 --launch-only`` with a launcher's environment around it and FRLW_DIST_FORCE=1: the rendezvous and
    the barrier / MAX / SUM reductions of the timed-region bracket on backend nccl."""
    torch = pytest.importorskip("torch")
    if torch.cuda.device_count() < 1:
        pytest.skip("no GPU")
    env = dict(_env(), FRLW_DIST_FORCE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--launch-only"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["launch_only"] and line["n_gpus"] == 1 and line["backend"] == "nccl"
    assert line["rank_sum"] == 1.0 and line["max_over_ranks"] == 0.5
