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
    ap.add_argument("--no-detector", action="store_true", help="skip the detector forward leg")
    ap.add_argument("--det-batch", type=int, default=32)
    ap.add_argument("--no-train", action="store_true", help="skip the train-step leg")
    ap.add_argument("--no-also", action="store_true", help="skip the GEN1-shaped TAF leg (clean per-kernel profiles)")
    ap.add_argument("--train-batch", type=int, default=64, help="per-GPU batch of the train-step leg")
    ap.add_argument("--local_rank", "--local-rank", type=int, default=None)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    from frlw_evd_amd import _lib, synth
    from frlw_evd_amd import dist as fd
    from frlw_evd_amd import event_representation as er
    rank, world, local_rank = fd.init_from_env("nccl", args.local_rank)  # one process per GPU, RCCL
    torch.cuda.set_device(local_rank)
    n_gpus = world
    _lib.load()

    seed, n, H, W, t_span, n_win, win_us, K = WORKLOADS[args.workload]
    ev = synth.synth_events(seed + 7919 * rank, n, W, H, t_span, hotspot=args.hotspot)
    dat_h = synth.to_dat8(ev)
    dat = torch.from_numpy(dat_h.view(np.uint8).reshape(-1, 8)).cuda()
    state = torch.full((H, W, 2, K), -6000.0, device="cuda")

    def step():
        return er.encode_taf_dat(dat, (H, W), state, 0, win_us, n_win, K, want_view=False, want_u8=True,
                                 flip_k=True, check=False)

    sync_all = fd.barrier_sync

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
    elapsed, dev_ms = fd.max_over_ranks([elapsed, dev_ms])

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
            "workload": f"{args.workload} (BASELINE.json configs[2]{' at the GEN1 shape' if args.workload == 'taf_gen1' else ''}): "
                        f"TAF K={K} encode + leaky transform + uint8, {n} events, {W}x{H}, "
                        f"{n_win} windows x {win_us} us, raw 8-byte DAT records resident in HBM"
                        + (", hotspot" if args.hotspot else ""),
            "events_per_step_per_gpu": n,
            "parallelism": f"sequence-sharded x{n_gpus} (no collective)",
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "frlw_taf_encode = k_hist + k_slabscan + k_tilescan + k_scatter + k_taf_tile (dominant: k_taf_tile)",
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

    if args.workload == "taf_mpx" and not args.no_also:
        # the same encoder at the GEN1 sensor shape BASELINE.json's metric names (304x240, 1 M events): launch /
        # latency bound at this size, reported next to the headline number
        s2, n2, H2, W2, t2, nw2, wu2, K2 = WORKLOADS["taf_gen1"]
        ev2 = synth.synth_events(s2 + 7919 * rank, n2, W2, H2, t2)
        dat2 = torch.from_numpy(synth.to_dat8(ev2).view(np.uint8).reshape(-1, 8)).cuda()
        st2 = torch.full((H2, W2, 2, K2), -6000.0, device="cuda")
        for _ in range(3):
            er.encode_taf_dat(dat2, (H2, W2), st2, 0, wu2, nw2, K2, check=False)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            er.encode_taf_dat(dat2, (H2, W2), st2, 0, wu2, nw2, K2, check=False)
        sync_all()
        dt2, = fd.max_over_ranks([time.perf_counter() - t0])
        result["also"] = [{"workload": "taf_gen1 (the GEN1 304x240 shape BASELINE.json's metric names): TAF K=8 encode + leaky + "
                                       "uint8, 1000000 events, 304x240, 8 windows",
                           "value": round(n_gpus * n2 / (dt2 / args.steps) / 1e6, 2), "unit": "Mevents/s",
                           "ms_per_step": round(dt2 / args.steps * 1e3, 4)}]
        # BASELINE.json configs[1]: Event Volume, 1 M events, 304x240, 5 bins (bit-exactness is the tests' job; this is its rate)
        ev4 = synth.synth_events(1002 + 7919 * rank, 1_000_000, W2, H2, 250_000)
        dat4 = torch.from_numpy(synth.to_dat8(ev4).view(np.uint8).reshape(-1, 8)).cuda()
        for _ in range(3):
            er.encode_ev_dat(dat4, (H2, W2), 250_000, 250_000, volume_bins=5, check=False)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            er.encode_ev_dat(dat4, (H2, W2), 250_000, 250_000, volume_bins=5, check=False)
        sync_all()
        dt4, = fd.max_over_ranks([time.perf_counter() - t0])
        result["also"].append({"workload": "ev_gen1 (BASELINE.json configs[1]): Event Volume 5 bins, 1000000 events, 304x240",
                               "value": round(n_gpus * 1_000_000 / (dt4 / args.steps) / 1e6, 2), "unit": "Mevents/s",
                               "ms_per_step": round(dt4 / args.steps * 1e3, 4)})
        if not args.hotspot:
            # SURVEY.md section 8d "report both": the contention variant of the headline workload (25 % of the events in
            # a sigma = 8 px blob -> a few tiles hold most of them; hot tiles are split over workgroups)
            ev3 = synth.synth_events(seed + 7919 * rank, n, W, H, t_span, hotspot=True)
            dat3 = torch.from_numpy(synth.to_dat8(ev3).view(np.uint8).reshape(-1, 8)).cuda()
            st3 = torch.full((H, W, 2, K), -6000.0, device="cuda")
            hs = max(3, min(10, args.steps))
            for _ in range(2):
                er.encode_taf_dat(dat3, (H, W), st3, 0, win_us, n_win, K, check=False)
            sync_all()
            t0 = time.perf_counter()
            for _ in range(hs):
                er.encode_taf_dat(dat3, (H, W), st3, 0, win_us, n_win, K, check=False)
            sync_all()
            dt3, = fd.max_over_ranks([time.perf_counter() - t0])
            result["also"].append({"workload": f"{args.workload}, hotspot variant (25 % of the events in a sigma = 8 px blob)",
                                   "value": round(n_gpus * n / (dt3 / hs) / 1e6, 2), "unit": "Mevents/s",
                                   "ms_per_step": round(dt3 / hs * 1e3, 4)})
            del dat3, st3, ev3
    if not args.no_detector:
        result["detector"] = bench_detector(args, torch, dist, world, rank, sync_all)
    if not args.no_train:
        try:
            result["train"] = bench_train(args, torch, world, rank, local_rank, sync_all)
        except Exception as e:  # never lose the headline line to the extra leg
            result["train"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(dat_h, n, H, W, K, n_win, win_us)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def bench_train(args, torch, world, rank, local_rank, sync_all):
    """SURVEY.md section 8d cfg 5 without the encode: YOLOX (16-channel TAF input) train step -- forward, batched
    SimOTA assignment (frlw_simota_assign) + losses, backward, Adam -- under DDP over RCCL when N > 1, per-GPU batch
    fixed (weak scaling).  Convolution forward/backward of this leg are torch/MIOpen autograd; the assignment is ours."""
    from frlw_evd_amd import dist as fd
    from frlw_evd_amd import e2e
    from frlw_evd_amd.trainer import Trainer
    B = args.train_batch
    m = e2e.build_model(in_channels=16, num_classes=2)
    tr = Trainer(m, global_batch=B * world, nodes=world, iters_per_epoch=100, local_rank=local_rank, ddp=world > 1)
    rng = np.random.default_rng(1005 + rank)
    x = torch.from_numpy(rng.integers(0, 256, size=(B, 16, 256, 320, 1, 1), dtype=np.uint8)).float().div(255).cuda()
    lab = torch.zeros(B, 80, 5, dtype=torch.float64)
    lab[:, 0] = torch.tensor([0, 100, 90, 60, 40.0])
    lab[:, 1] = torch.tensor([1, 220, 150, 50, 80.0])
    lab = lab.cuda()
    steps = 5

    def timed():
        for i in range(3):
            tr.train_step(x, lab, i)
        sync_all()
        t0 = time.perf_counter()
        for i in range(steps):
            loss, _ = tr.train_step(x, lab, 3 + i)
        sync_all()
        dt, = fd.max_over_ranks([time.perf_counter() - t0])
        return dt, loss

    dt, loss = timed()  # BaseConv forward / backward in the gfx950 kernels of csrc/train_ops.hip (the default)
    # BASELINE.json configs[4]: the same step fed by the TAF encode of its batch (B GEN1-shaped streams of 8 x 125 000
    # events -> TAF K=8 -> leaky -> uint8 -> nearest 256x320 -> /255), everything on this GPU
    src = e2e.SyntheticTafSource(B, seed=1005 + 1000 * rank)
    idx = list(range(B))
    for i in range(2):
        tr.train_step(src.encode_batch(idx), lab, i)
    sync_all()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.train_step(src.encode_batch(idx), lab, 2 + i)
    sync_all()
    dt_e2e, = fd.max_over_ranks([time.perf_counter() - t0])
    t0 = time.perf_counter()
    for i in range(steps):
        src.encode_batch(idx)
    sync_all()
    dt_enc, = fd.max_over_ranks([time.perf_counter() - t0])
    prev = os.environ.get("FRLW_NATIVE_TRAIN")
    os.environ["FRLW_NATIVE_TRAIN"] = "0"  # the same step with torch autograd / MIOpen convolutions, for comparison
    try:
        dt_t, _ = timed()
    finally:
        if prev is None:
            os.environ.pop("FRLW_NATIVE_TRAIN", None)
        else:
            os.environ["FRLW_NATIVE_TRAIN"] = prev
    return {"metric": "YOLOX train step (frames/s)", "value": round(world * B * steps / dt, 1), "unit": "frames/s",
            "ms_per_step": round(dt / steps * 1e3, 3), "per_gpu_batch": B, "steps": steps, "loss": round(loss, 4),
            "parallelism": f"ddp{world}" if world > 1 else "single", "scaling": "weak",
            "convolutions": "csrc/train_ops.hip (fp32 MFMA fwd / dgrad / wgrad, BatchNorm + SiLU fwd / bwd), SimOTA csrc/simota.hip",
            "same_step_with_miopen_convs": {"value": round(world * B * steps / dt_t, 1), "ms_per_step": round(dt_t / steps * 1e3, 3)},
            "encode_plus_train_step": {"workload": "BASELINE.json configs[4]: TAF encode of the batch (8 x 125 000 events per "
                                                   "304x240 sample) + train step", "value": round(world * B * steps / dt_e2e, 1),
                                       "unit": "frames/s", "ms_per_step": round(dt_e2e / steps * 1e3, 3),
                                       "encode_ms_per_batch": round(dt_enc / steps * 1e3, 3)}}


FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, f32 in / f32 accumulate


def bench_detector(args, torch, dist, world, rank, sync_all):
    """Second half of BASELINE.json's metric: YOLOX forward frames/s (SURVEY.md section 8d cfg 4):
    B = 32, (10, 256, 320) f32 input (the detector shape of a 304x240 sensor), recipe weights, eval mode,
    forward to the pre-NMS tensor (B, 1680, 7); decode + NMS timed separately."""
    from frlw_evd_amd.yolox import build_yolox
    from frlw_evd_amd.yolox.model import recipe_state_dict
    B = args.det_batch
    net = build_yolox(10, 2)
    net.load_state_dict(recipe_state_dict(net, seed=1004))
    net.eval()
    rng = np.random.default_rng(1004 + rank)
    x_h = torch.from_numpy(rng.integers(0, 256, size=(B, 10, 256, 320)).astype(np.float32) / np.float32(255))
    x = x_h.cuda()
    eng = net.engine()
    steps = max(5, min(args.steps, 30))
    for _ in range(3):
        eng.raw_outputs(x)
    sync_all()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        eng.raw_outputs(x)
    e1.record()
    sync_all()
    elapsed = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1) / steps
    for _ in range(2):
        eng.detect(x)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        eng.detect(x)
    torch.cuda.synchronize()
    full_ms = (time.perf_counter() - t1) / 5 * 1e3
    from frlw_evd_amd import dist as fd
    elapsed, dev_ms = fd.max_over_ranks([elapsed, dev_ms])
    tflops = eng.flops_per_image * B / (dev_ms * 1e-3) / 1e12
    out = {
        "metric": "YOLOX-S (CSPDarknet + PAFPN + decoupled head) eval forward to the pre-NMS tensor (BASELINE.json configs[3])",
        "value": round(world * B / (elapsed / steps), 1), "unit": "frames/s", "batch_per_gpu": B,
        "input": "(B, 10, 256, 320) f32, recipe weights", "steps": steps, "ms_per_batch": round(elapsed / steps * 1e3, 3),
        "dtype": "f32", "fwd_plus_decode_nms_ms": round(full_ms, 3),
        "roofline": {"bound": "mfma", "kernel": "k_conv_mfma (80 launches per forward)", "achieved": round(tflops, 2),
                     "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tflops / FP32_MFMA_PEAK_TFLOPS, 4),
                     "flops_per_image": eng.flops_per_image, "device_ms_per_batch": round(dev_ms, 3),
                     "mfma": "v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate)"},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # PyTorch-CPU forward of the same module definition
        nthreads = min(os.cpu_count() or 1, 64)
        torch.set_num_threads(nthreads)
        xb = x_h[:8, ..., None]
        with torch.no_grad():
            net.reference_outputs(xb)
            best = None
            for _ in range(3):
                t2 = time.perf_counter()
                net.reference_outputs(xb)
                dt = time.perf_counter() - t2
                best = dt if best is None else min(best, dt)
        out["cpu_baseline"] = {"value": round(8 / best, 1), "unit": "frames/s", "cores": nthreads, "kind": "port",
                               "sample": f"PyTorch-CPU fp32 forward of the same modules, batch 8, best of 3 ({best:.3f} s)"}
    return out


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
