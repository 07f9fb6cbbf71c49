"""One-process-per-GPU helpers (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).

The hot path shards without a data-path collective: encoders by sequence (the FIFO state is per sequence,
generate_taf.py:143-160), the detector forward by batch.  The only collectives are the barrier that brackets a
timed region and the MAX over ranks of the measured times (bench.py); the train step adds the gradient
all-reduce through DistributedDataParallel (core/exp.py:391).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, local_rank_arg=None):
    """(rank, world, local_rank) from RANK / WORLD_SIZE / LOCAL_RANK; accepts the launcher's --local_rank too
    (train.py:10; torch >= 2 passes --local-rank / LOCAL_RANK)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", local_rank_arg if local_rank_arg is not None else 0))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("FRLW_DIST_BACKEND", backend)  # e.g. gloo: several ranks on one GPU in tests
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, init_method="env://", rank=rank, world_size=world, **kw)
    if torch.cuda.is_available() and local_rank >= torch.cuda.device_count():
        local_rank = local_rank % torch.cuda.device_count()  # only reachable with FRLW_DIST_BACKEND=gloo
    return rank, world, local_rank


def shard_round_robin(items, rank, world):
    """Sequences (or samples) of this rank: item i goes to rank i % world."""
    return list(items)[rank::world]


def shard_range(n, rank, world):
    """Contiguous balanced [lo, hi) of n units for this rank (sizes differ by at most one)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def barrier_sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()


def max_over_ranks(values):
    """Element-wise MAX of a list of floats over all ranks (the time a job takes is its slowest rank's)."""
    if not dist.is_initialized():
        return [float(v) for v in values]
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]


def sum_over_ranks(values):
    if not dist.is_initialized():
        return [float(v) for v in values]
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t]


def job_throughput(units_this_rank, seconds_this_rank):
    """Whole-job rate: units all ranks processed / the slowest rank's time."""
    total = sum_over_ranks([units_this_rank])[0]
    return total / max_over_ranks([seconds_this_rank])[0]
