#!/usr/bin/env python3
"""bench.py -- the hot path of BASELINE.json on MI355X: TAF encode (Mevents/s) + roofline + CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload taf_mpx|taf_gen1|ev_gen1]

A step = one pass of the fused TAF encoder (libfrlw_evd.so, `frlw_taf_encode`) over one batch of
synthetic DAT records already resident in HBM: SURVEY.md section 8d cfg 3 -- seed 1003, 10 M events,
1280x720 native, 8 windows x 10 ms, K = 8, leaky transform + uint8 output, FIFO state carried from
step to step like consecutive labels of one sequence (generate_taf.py:175-186).

N > 1: one process per GPU (torch.distributed.run), every rank encodes its own independent stream
(the path shards by sequence, no data-path collective) -> weak scaling; the timed region is
bracketed by barrier + synchronize and the MAX over ranks is reported.

The JSON line also carries
  roofline     algorithmic bytes of the encode (8 B/event + FIFO state read + write + uint8 out) over
               the device time of one encode measured with HIP events on the launch stream;
  cpu_baseline the CPU oracle (oracle/frlw_oracle.c, a port of the reference's algorithm, 1 thread)
               timed on this host on the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (seed, n_events, H, W, t_span, n_windows, window_us, K)
    "taf_mpx": (1003, 10_000_000, 720, 1280, 80_000, 8, 10_000, 8),
    "taf_gen1": (1005, 1_000_000, 240, 304, 80_000, 8, 10_000, 8),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def taf_algorithmic_bytes(n, H, W, K):
    """events once (8 B DAT record), FIFO state read once + written once, uint8 out once."""
    return 8 * n + 2 * (4 * 2 * K * H * W) + 2 * K * H * W


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="taf_mpx", choices=sorted(WORKLOADS))
    ap.add_argument("--hotspot", action="store_true", help="25 %% of the events in a sigma-8px blob")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--local_rank", "--local-rank", type=int, default=None)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", args.local_rank if args.local_rank is not None else 0))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", init_method="env://", device_id=torch.device("cuda", local_rank))
    n_gpus = world

    from frlw_evd_amd import _lib, synth
    from frlw_evd_amd import event_representation as er
    _lib.load()

    seed, n, H, W, t_span, n_win, win_us, K = WORKLOADS[args.workload]
    ev = synth.synth_events(seed + 7919 * rank, n, W, H, t_span, hotspot=args.hotspot)
    dat_h = synth.to_dat8(ev)
    dat = torch.from_numpy(dat_h.view(np.uint8).reshape(-1, 8)).cuda()
    state = torch.full((H, W, 2, K), -6000.0, device="cuda")

    def step():
        return er.encode_taf_dat(dat, (H, W), state, 0, win_us, n_win, K, want_view=False, want_u8=True,
                                 flip_k=True, check=False)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # correctness guard: data-dependent status of the first encode must be clean
    u8, _ = er.encode_taf_dat(dat, (H, W), state, 0, win_us, n_win, K, check=True)
    for _ in range(args.warmup):
        step()
    sync_all()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    sync_all()
    elapsed = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1) / args.steps  # HIP events on the launch stream (torch's current stream)
    if world > 1:
        t = torch.tensor([elapsed, dev_ms], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, dev_ms = float(t[0]), float(t[1])

    ms_per_step = elapsed / args.steps * 1e3
    value = n_gpus * n / (elapsed / args.steps) / 1e6
    alg_bytes = taf_algorithmic_bytes(n, H, W, K)
    achieved = alg_bytes / (dev_ms * 1e-3) / 1e9

    result = {
        "metric": "TAF encode throughput (Mevents/s)",
        "value": round(value, 2),
        "unit": "Mevents/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: TAF K={K} encode + leaky transform + uint8, {n} events, {W}x{H}, "
                        f"{n_win} windows x {win_us} us, raw 8-byte DAT records resident in HBM"
                        + (", hotspot" if args.hotspot else ""),
            "events_per_step_per_gpu": n,
            "parallelism": f"sequence-sharded x{n_gpus} (no collective)",
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "frlw_taf_encode = k_hist + k_colscan + k_tilescan + k_scatter + k_taf_tile",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "algorithmic_bytes": alg_bytes,
            "device_ms_per_encode": round(dev_ms, 4),
            "traffic": None,
        },
    }
    traffic_file = os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")
    if os.path.exists(traffic_file):  # PMC passes are separate runs (tools/profile.sh); per-encode HBM bytes
        with open(traffic_file) as f:
            result["roofline"]["traffic"] = json.load(f).get("hbm_bytes_per_encode")

    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(dat_h, n, H, W, K, n_win, win_us)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(dat_h, n, H, W, K, n_win, win_us, budget_s=20.0):
    """The CPU oracle (a 1-thread C port of generate_taf.py:19-76 + harness) on the same stream."""
    from oracle import oracle as orc
    orc.build()
    st0 = np.full((H, W, 2, K), -6000, np.float32)
    best = None
    spent = 0.0
    runs = 0
    while runs < 3 and spent < budget_s:
        t0 = time.perf_counter()
        view, st = orc.taf_stream_dat8(dat_h, (H, W), (H, W), K, 0, win_us, n_win, st0)
        u8 = orc.quantize_u8(orc.leaky_transform(view))
        dt = time.perf_counter() - t0
        spent += dt
        runs += 1
        best = dt if best is None else min(best, dt)
    return {"value": round(n / best / 1e6, 3), "unit": "Mevents/s", "cores": 1, "kind": "port",
            "sample": f"the full workload ({n} events, {n_win} windows), best of {runs} runs, {best:.3f} s each",
            "host_cpus": os.cpu_count()}


if __name__ == "__main__":
    main()
