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


def init_from_env(backend=None, local_rank_arg=None, force=False):
    """(rank, world, local_rank) from RANK / WORLD_SIZE / LOCAL_RANK; accepts the launcher's --local_rank too
    (train.py:10; torch >= 2 passes --local-rank / LOCAL_RANK).  A one-rank job joins no process
    group unless ``force`` (or ``FRLW_DIST_FORCE=1``) asks for it -- the reference's train.py:31 always does, and a
    one-rank ``nccl`` group is how the RCCL code paths are exercised on a one-GPU box (tests/test_rccl_gpu.py)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", local_rank_arg if local_rank_arg is not None else 0))
    force = force or os.environ.get("FRLW_DIST_FORCE") == "1"
    if (world > 1 or force) and not dist.is_initialized():
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


def agree_fast_path(group=None, local_ok=None):
    """One verdict for the whole job on the encoders' fast paths: the MINIMUM over the ranks of each rank's LDS lane-order
    self-test (``event_representation.fast_path_ok``: once per process and device, 0.7 ms -- eight times on an 8-GPU node).  A
    rank whose device fails the test would quietly take the general path -- same bits, several times the encode time -- and
    every other rank would wait for it at the next collective; after this call either every rank runs the fast kernels or none
    does (``event_representation.FAST_PATH_ENABLED``).  Returns the job's verdict.  ``local_ok``: this rank's own verdict when the
    caller has it already (tests on CPU).  Call it once, after ``init_from_env`` and before the first encode."""
    from . import event_representation as er
    if local_ok is None:
        local_ok = er.fast_path_ok() if torch.cuda.is_available() else True
    ok = bool(local_ok)
    if dist.is_initialized():
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        ok = bool(int(t.item()))
    er.FAST_PATH_ENABLED = ok
    return ok


def job_throughput(units_this_rank, seconds_this_rank):
    """Whole-job rate: units all ranks processed / the slowest rank's time."""
    total = sum_over_ranks([units_this_rank])[0]
    return total / max_over_ranks([seconds_this_rank])[0]


# ---- DistributedDataParallel wiring of the train step (core/exp.py:391) -------------------------------------------
def ddp_kwargs(bucket_cap_mb=None):
    """Keyword arguments for ``DistributedDataParallel(model, device_ids=[local_rank], broadcast_buffers=False, ...)``.

    * ``gradient_as_bucket_view``: gradients live inside the all-reduce buckets (no copy in, no copy out);
    * ``static_graph``: the autograd graph is the same every step (every parameter gets a gradient), so the buckets are
      rebuilt once in the order the gradients become ready and the reducer skips its unused-parameter search;
    * ``bucket_cap_mb``: the 57.5 MB of gradients travel in buckets of this size (default 25 -> 3 collectives per step,
      each large enough to run at link bandwidth over xGMI, with the head gradients -- ready first -- already in flight
      while the backbone's backward runs).  ``FRLW_DDP_BUCKET_MB`` overrides (host-side launch configuration).
    """
    cap = bucket_cap_mb if bucket_cap_mb is not None else int(os.environ.get("FRLW_DDP_BUCKET_MB", "25"))
    return {"gradient_as_bucket_view": True, "static_graph": True, "bucket_cap_mb": cap}


def broadcast_module_state(module, src=0, group=None, buffers=False):
    """Every rank takes rank ``src``'s parameters, in place -- what ``DistributedDataParallel.__init__`` does before the first
    step (core/exp.py:391 relies on it: the ranks build their models independently).  Buffers (the BatchNorm statistics) stay
    per rank unless asked for, as under ``broadcast_buffers=False``.  One broadcast per dtype: the float32 parameters travel
    as one flat message.  Returns the number of collectives issued."""
    if not dist.is_initialized():
        return 0
    tensors = [p.data for p in module.parameters()]
    if buffers:
        tensors += list(module.buffers())
    by_dtype = {}
    for t in tensors:
        if t.numel():
            by_dtype.setdefault((t.dtype, t.device), []).append(t)
    n = 0
    with torch.no_grad():
        for group_tensors in by_dtype.values():
            flat = torch.cat([t.reshape(-1) for t in group_tensors])
            dist.broadcast(flat, src, group=group)
            for t, v in zip(group_tensors, flat.split([t.numel() for t in group_tensors])):
                t.copy_(v.view_as(t))
            n += 1
    return n


def reduce_scatter_allgather_hook(group, bucket):
    """DDP communication hook: average a gradient bucket with a reduce-scatter followed by an all-gather instead of one
    all-reduce.  On the fully connected xGMI mesh of an 8-GPU MI355X node each phase sends 1/world of the bucket to every
    peer over its own link (7 links busy), where a ring all-reduce pushes 2 * (world - 1) / world of the bucket through
    one link per direction.  Opt in with ``model.register_comm_hook(None, ...)`` or
    ``FRLW_DDP_HOOK=rs_ag`` (``install_comm_hook``).  Result = DDP's default (sum / world), bit for bit on the same
    reduction order; tests/test_dist_cpu.py checks it against the default hook on gloo."""
    group = group if group is not None else dist.group.WORLD
    world = dist.get_world_size(group)
    buf = bucket.buffer()
    n = buf.numel()
    per = (n + world - 1) // world
    if per * world != n:
        padded = torch.zeros(per * world, dtype=buf.dtype, device=buf.device)
        padded[:n].copy_(buf)
    else:
        padded = buf
    shard = torch.empty(per, dtype=buf.dtype, device=buf.device)
    if dist.get_backend(group) == "gloo":  # gloo has no reduce_scatter_tensor: same data flow with its primitives
        dist.all_reduce(padded, group=group)
        shard.copy_(padded[dist.get_rank(group) * per:(dist.get_rank(group) + 1) * per])
        fut = torch.futures.Future()
        fut.set_result(shard)
    else:
        fut = dist.reduce_scatter_tensor(shard, padded, group=group, async_op=True).get_future()

    def gather(f):
        shard.div_(world)
        if dist.get_backend(group) == "gloo":
            parts = [torch.empty_like(shard) for _ in range(world)]
            dist.all_gather(parts, shard, group=group)
            padded.copy_(torch.cat(parts))
        else:
            dist.all_gather_into_tensor(padded, shard, group=group, async_op=True).get_future().wait()
        if padded is not buf:
            buf.copy_(padded[:n])
        return buf

    return fut.then(gather)


class TimedAllreduce:
    """DDP communication hook that is the default all-reduce (sum / world) plus device-side timing of every bucket:
    ``.summary()`` -> bucket sizes and the mean time a bucket's collective took.  How much of that time is EXPOSED (not
    hidden behind the rest of the backward) is measured by the caller as step time with DDP minus step time inside
    ``model.no_sync()`` (bench.py reports both)."""

    def __init__(self, group=None):
        self.group = group
        self.events = []  # (bytes, start event, end event)

    def record(self, bucket):
        group = self.group if self.group is not None else dist.group.WORLD
        world = dist.get_world_size(group)
        buf = bucket.buffer()
        cuda = buf.is_cuda
        if cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        fut = dist.all_reduce(buf, group=group, async_op=True).get_future()

        def done(f):
            buf.div_(world)
            if cuda:
                e1.record()
                self.events.append((buf.numel() * buf.element_size(), e0, e1))
            return buf

        return fut.then(done)

    def summary(self, reset=True):
        if not self.events:
            return {"buckets": 0}
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        sizes = sorted({b for b, _, _ in self.events}, reverse=True)
        ms = [a.elapsed_time(b) for _, a, b in self.events]
        out = {"buckets": len(self.events), "bucket_bytes": sizes, "mean_bucket_ms": sum(ms) / len(ms), "total_ms": sum(ms)}
        if reset:
            self.events = []
        return out


def timed_allreduce_hook(state, bucket):
    """The function DDP calls (it wants a plain function with a ``__name__``); ``state`` is the TimedAllreduce."""
    return state.record(bucket)


def install_comm_hook(ddp_model, kind=None):
    """``kind``: None / "default" (DDP's all-reduce), "rs_ag" (reduce-scatter + all-gather), "timed" (default + timing;
    returns the TimedAllreduce object).  ``FRLW_DDP_HOOK`` supplies the default."""
    kind = kind or os.environ.get("FRLW_DDP_HOOK", "default")
    if kind == "rs_ag":
        ddp_model.register_comm_hook(None, reduce_scatter_allgather_hook)
        return None
    if kind == "timed":
        hook = TimedAllreduce()
        ddp_model.register_comm_hook(hook, timed_allreduce_hook)
        return hook
    return None
