#!/usr/bin/env python3
"""bench.py -- the hot path of BASELINE.json on MI355X: TAF encode (Mevents/s) + roofline + CPU baseline, then the other
rows of SURVEY.md section 8(d): GEN1-shaped and batched encodes, Event Volume, detector forward, train step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload taf_mpx|taf_gen1] [--hotspot]

A step = one pass of the TAF encoder over one batch of synthetic DAT records already resident in HBM: SURVEY.md 8(d)
cfg 3 -- seed 1003, 10 M events, 1280x720 native, 8 windows x 10 ms, K = 8, leaky transform + uint8 output, FIFO state
carried from step to step like consecutive labels of one sequence (generate_taf.py:175-186) -- through
``frlw_taf_encode_batch`` (csrc/taf_fast.hip; the general path ``frlw_taf_encode`` is timed beside it).

N > 1: one process per GPU (torch.distributed.run), every rank encodes its own independent stream (the path shards by
sequence, no data-path collective) -> weak scaling; the timed region is bracketed by barrier + synchronize and the MAX
over ranks is reported.  The train leg runs under DistributedDataParallel (RCCL) and reports how much of the gradient
all-reduce is exposed.

The JSON line carries
  roofline     algorithmic bytes of the encode (8 B/event + FIFO state read + write + uint8 out) over the device time
               of one encode measured with HIP events on the launch stream; ``frac`` against the 8 TB/s of the data
               sheet, ``frac_of_copy`` against the float4 copy rate measured in this run; ``traffic`` = HBM bytes per
               encode from a separate rocprofv3 --pmc pass (profiles/traffic_<workload>.json), only while the kernel
               sources are the ones that pass measured;
  cpu_baseline the CPU oracle (oracle/frlw_oracle.c, a port of the reference's algorithm) timed on this host on the same
               workload with 1 thread and with all cores (one independent stream per thread), rank 0 at N = 1 only.
"""
import argparse
import hashlib
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (seed, n_events, H, W, t_span, n_windows, window_us, K)
    "taf_mpx": (1003, 10_000_000, 720, 1280, 80_000, 8, 10_000, 8),
    "taf_gen1": (1005, 1_000_000, 240, 304, 80_000, 8, 10_000, 8),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, f32 in / f32 accumulate
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16: 32 cycles per SIMD)


def mfma_roofline(tflops, precision, kernel, **extra):
    """The ``roofline`` object of a convolution workload.  ``tflops`` = ALGORITHMIC rate (2 x MACs of the float32 convolutions
    / device time).  precision "f32" (the default): one v_mfma_f32_32x32x2_f32 per product, peak 157.3.  precision "bf16x3"
    (opt-in, FRLW_CONV_PRECISION): every float32 product is three bf16 MFMAs (hi/lo split operands, f32 accumulate), so the peak of the
    ALGORITHMIC rate is a third of the dense bf16 peak; the executed matrix rate and the ratio to the float32-MFMA peak -- what the
    same float32 work could reach at best on the float32 instruction -- are listed beside it."""
    if precision == "bf16x3":
        peak = BF16_MFMA_PEAK_TFLOPS / 3.0
        r = {"bound": "mfma", "kernel": kernel, "achieved": round(tflops, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
             "frac": round(tflops / peak, 4), "executed_bf16_TFLOPs": round(3 * tflops, 1), "executed_peak": BF16_MFMA_PEAK_TFLOPS,
             "x_f32_mfma_peak": round(tflops / FP32_MFMA_PEAK_TFLOPS, 3),
             "mfma": "3 x v_mfma_f32_32x32x16_bf16 per 16 k: x = hi + lo (bf16 each), a*b ~ a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, f32 "
                     "accumulate -- float32 products to <= 2^-16 relative (observed 3e-6 of max |y| per layer; tolerance 1e-3); "
                     "peak = dense bf16 peak / 3; opt-in (FRLW_CONV_PRECISION=bf16x3) beside the default v_mfma_f32_32x32x2_f32 (exact products)"}
    else:
        r = {"bound": "mfma", "kernel": kernel, "achieved": round(tflops, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
             "frac": round(tflops / FP32_MFMA_PEAK_TFLOPS, 4), "mfma": "v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate)"}
    r.update(extra)
    return r


def taf_algorithmic_bytes(n, H, W, K):
    """events once (8 B DAT record), FIFO state read once + written once, uint8 out once."""
    return 8 * n + 2 * (4 * 2 * K * H * W) + 2 * K * H * W


def ev_algorithmic_bytes(n, H, W, bins):
    """SURVEY.md 8(d) cfg 2: 8 B per event + the f32 (2 * bins, H, W) output."""
    return 8 * n + 4 * 2 * bins * H * W


def sae_algorithmic_bytes(n, H, W, n_lam):
    """8 B per event + the (2, H, W) f32 memory read and written + the f32 (2 * n_lam, H, W) output."""
    return 8 * n + 2 * (4 * 2 * H * W) + 4 * 2 * n_lam * H * W


def eci_algorithmic_bytes(n, H, W):
    """8 B per event + the f32 (2, H, W) output."""
    return 8 * n + 4 * 2 * H * W


def kernel_source_sha():
    """Identity of the encoder kernels a PMC traffic figure belongs to."""
    h = hashlib.sha256()
    for name in ("taf_fast.hip", "partition.hip", "encoders.hip", "frlw_common.h"):
        with open(os.path.join(ROOT, "frlw-evd_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def one_region(timer, fn, steps, warmup, extra=0):
    """ONE timed region of `steps` steps (the first one after the warm-up) is returned as the value -- never a best-of.  `extra`
    further regions are returned third; the encoder and train rows ignore them, the DETECTOR rows report the MEDIAN of their three
    regions (stated in their `value_is`; DESIGN.md 4: a fresh box now and then spends tens of ms of one region on a clock transition)."""
    first = timer.run(fn, steps, warmup)
    more = [timer.run(fn, steps, 1) for _ in range(extra)]
    return first[0], first[1], [first] + more


def physical_cores():
    """(physical cores this process may run on, hardware threads per core): the CPU baselines use one thread per PHYSICAL
    core (n = 1 and n = all cores), not one per logical CPU and not an arbitrary cap."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    cores = set()
    for c in cpus:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as f:
                cores.add(f.read().strip())
        except OSError:
            cores.add(str(c))
    n = max(1, len(cores))
    return n, max(1, len(cpus) // n)


class Timer:
    """K timed steps between barrier + synchronize on both sides; device time by HIP events on the launch stream."""

    def __init__(self, torch, fd, er=None):
        self.torch, self.fd, self.er = torch, fd, er

    def run(self, fn, steps, warmup):
        torch = self.torch
        for _ in range(warmup):
            fn()
        self.fd.barrier_sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        self.fd.barrier_sync()
        wall = time.perf_counter() - t0
        wall, dev_ms = self.fd.max_over_ranks([wall, e0.elapsed_time(e1) / steps])
        if self.er is not None:
            self.er.raise_deferred("bench.py timed region")  # unchecked encoder calls: their status, outside the timed region
        return wall / steps, dev_ms


def graphed_row(row, units, per_graph, eager):
    """A small single-stream encode: `value` / `ms_per_step` = K EAGER calls (what rounds 1-4 reported under these keys and what a
    Python caller in a loop gets); `graph_value` / `graph_ms_per_step` = the same K steps replayed as ONE HIP graph (the kernels
    without ~20 us of Python per call); roofline.device_ms is the graph replay's (kernel time, the roofline's denominator)."""
    row["value"] = round(units / eager / 1e6, 2)
    row["ms_per_step"] = round(eager * 1e3, 4)
    row["graph_value"] = round(units / per_graph / 1e6, 2)
    row["graph_ms_per_step"] = round(per_graph * 1e3, 4)
    row["launch"] = "value, ms_per_step: K eager calls; graph_*: the K steps as ONE HIP graph replay; roofline: graph device time"
    return row


def run_graphed(timer, fn, steps, warmup):
    """The K steps of a SMALL encode captured into ONE HIP graph and replayed once inside the timed region (barrier + synchronize
    on both sides as everywhere): a 100 000-event Event Count Image is 13 us of kernels behind ~20 us of Python per call, and a
    launch-bound inner loop belongs in a graph.  Returns (seconds per step, device ms per step, eager seconds per step)."""
    torch = timer.torch
    eager, _ = timer.run(fn, steps, warmup)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()  # (workspace, tables and the lane-order self-test exist on this stream before the capture)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(steps):
                fn()
    torch.cuda.synchronize()
    graph.replay()  # warm-up replay
    torch.cuda.synchronize()
    timer.fd.barrier_sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    graph.replay()
    e1.record()
    timer.fd.barrier_sync()
    wall = time.perf_counter() - t0
    wall, dev_ms = timer.fd.max_over_ranks([wall, e0.elapsed_time(e1) / steps])
    with torch.cuda.stream(side):
        if timer.er is not None:
            timer.er.raise_deferred("bench.py graphed region")
    return wall / steps, dev_ms, eager


def roofline(alg_bytes, dev_ms, kernel, copy_gbs, units):
    achieved = alg_bytes / (dev_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_of_copy": round(achieved / copy_gbs, 4) if copy_gbs else None,
            "copy_GBs_measured": round(copy_gbs, 1) if copy_gbs else None, "algorithmic_bytes": alg_bytes,
            "units_per_launch": units, "device_ms": round(dev_ms, 4), "traffic": None}


# The driver reads back a bounded number of bytes and its parser keeps the FIRST 24 keys of `roofline`: the line rank 0 prints is
# the COMPACT line (<= 6 000 bytes, tests/test_bench_line.py) -- the 12 top-level scalars, `config` (four short strings), `roofline`
# = these 24 keys in this order + at most 30 further scalars (ROOFLINE_MORE), `cpu_baseline` = CPU_BASELINE_KEYS.  Every other
# block (`also`, `gen1`, `detector`, `train`, `general_path`, `stripe_sharding`, `allreduce`, every descriptive string) goes to
# the side file bench_detail.json next to this script (tools/refresh_r06.sh copies it into profiles/).
ROOFLINE_FIRST_24 = (
    "bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "device_ms", "algorithmic_bytes", "traffic_x_algorithmic",
    "frac_of_copy", "copy_GBs_measured",
    "detector_frames_per_s", "detector_ms_per_batch", "detector_frac", "detector_1mpx_frac",
    "train_ms", "train_frac", "encode_plus_train_ms",
    "gen1_taf_single_graph_ms", "gen1_taf_single_frac", "gen1_taf_x64_frac", "gen1_ev_single_graph_ms", "gen1_ev_x64_frac",
)
# (`*_graph_ms`: K steps replayed as ONE HIP graph; `*_eager_ms`: the same K steps as K eager calls -- round 4 and before reported
#  the eager time under `gen1_*_single_ms`, round 5 the graph replay under the same name; the two now have a key each)
ROOFLINE_MORE = (
    "taf_mpx_hotspot_ms", "taf_mpx_hotspot_frac", "taf_mpx_hotspot_traffic_x", "general_path_ms",
    "gen1_taf_single_eager_ms", "gen1_taf_single_traffic_x", "gen1_taf_x64_ms", "gen1_taf_x64_traffic_x",
    "gen1_ev_single_eager_ms", "gen1_ev_single_default_call_ms", "gen1_ev_single_frac", "gen1_ev_single_traffic_x",
    "gen1_ev_x64_ms", "gen1_ev_x64_traffic_x",
    "sae_gen1_graph_ms", "sae_gen1_frac", "eci_gen1_graph_ms", "eci_gen1_frac",
    "detector_TFLOPs", "detector_fwd_nms_ms", "detector_fwd_nms_typical_ms", "detector_1mpx_ms_per_batch", "detector_1mpx_fwd_nms_ms",
    "train_TFLOPs", "train_eager_ms", "train_miopen_ms", "encode_ahead_plus_train_ms",
    "allreduce_exposed_ms", "global64_ms", "stripe_ms",
)
CPU_BASELINE_KEYS = ("value", "unit", "cores", "kind", "cpu", "all_cores_value", "all_cores_threads", "sample")
TOP_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
CONFIG_KEYS = ("workload", "path", "parallelism", "detail")
LINE_LIMIT = 6000
DETAIL_FILE = "bench_detail.json"


def _short(v, n):
    return v if not isinstance(v, str) or len(v) <= n else v[:n - 1] + "~"


def order_roofline(result):
    """Re-key result['roofline'] (the DETAIL form): ROOFLINE_FIRST_24 first (None where a leg did not run: the position is what
    counts), ROOFLINE_MORE behind them, then whatever else the legs wrote; descriptive strings move to result['config']."""
    roof = result["roofline"]
    if roof.get("traffic") and roof.get("algorithmic_bytes"):
        roof["traffic_x_algorithmic"] = round(roof["traffic"] / roof["algorithmic_bytes"], 3)
    cfg = result.setdefault("config", {})
    for k in ("units_per_launch", "traffic_source", "traffic_note"):
        if k in roof:
            cfg[k] = roof.pop(k)
    ordered = {k: roof.get(k) for k in ROOFLINE_FIRST_24}
    for k in ROOFLINE_MORE:
        if k in roof:
            ordered[k] = roof[k]
    for k, v in roof.items():
        if k not in ordered:
            ordered[k] = v
    result["roofline"] = ordered
    return result


def compact_line(result):
    """The ONE line rank 0 prints, from the detail form `order_roofline` left: scalars only inside `roofline` (the two strings
    `bound`, `unit` and a short `kernel` aside), no prose, nothing nested."""
    line = {k: result.get(k) for k in TOP_KEYS}
    cfg = result.get("config", {})
    line["config"] = {"workload": _short(cfg.get("workload", ""), 160), "path": _short(cfg.get("path", ""), 48),
                      "parallelism": _short(cfg.get("parallelism", ""), 48), "detail": DETAIL_FILE}
    roof = result["roofline"]
    out = {}
    for k in ROOFLINE_FIRST_24 + ROOFLINE_MORE:
        v = roof.get(k)
        if k in ROOFLINE_FIRST_24 or v is not None:
            if isinstance(v, (dict, list)):
                continue
            out[k] = _short(v, 140) if k == "kernel" else _short(v, 16)
    line["roofline"] = out
    cb = result.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: _short(cb[k], 96 if k == "sample" else 48) for k in CPU_BASELINE_KEYS if k in cb}
    return line


def emit(result, out, write_detail=True):
    """Detail -> bench_detail.json (and stderr), the compact line -> the real stdout.  The line is refused when it outgrows
    LINE_LIMIT: a line the driver cannot parse is worth nothing (BENCH_r05.json)."""
    order_roofline(result)
    line = compact_line(result)
    text = json.dumps(line)
    if len(text) > LINE_LIMIT or "\n" in text or not text.isascii():
        raise SystemExit(f"bench.py: the JSON line is {len(text)} bytes (limit {LINE_LIMIT}) or not one ASCII line")
    if write_detail:
        detail = dict(result, line=line)
        try:
            with open(os.path.join(ROOT, DETAIL_FILE), "w") as f:
                json.dump(detail, f, indent=1)
        except OSError as e:  # (a read-only checkout: the line still goes out)
            print(f"bench.py: {DETAIL_FILE} not written: {e}", file=sys.stderr)
        print("bench.py detail: " + json.dumps(result), file=sys.stderr, flush=True)
    print(text, file=out, flush=True)
    return line


def fast_kernel_label(er=None):
    """The launch sequence `frlw_taf_encode_batch` runs on the headline workload (chunk-major partition with large chunks, DESIGN.md
    3.6; profiles/r05_bench_kernel_stats.csv shows exactly these names)."""
    return "frlw_taf_encode_batch = kf_scatter_cm + kf_split_whole + kf_segcount_cm + kf_split_place + kf_taf_walk (dominant)"


def attach_traffic(roof, tag):
    """roofline.traffic = HBM bytes per encode from the separate rocprofv3 --pmc passes kept under profiles/ (tools/refresh_r03.sh),
    reported only while the kernel sources are the ones those passes measured."""
    path = os.path.join(ROOT, "profiles", f"traffic_{tag}.json")
    if not os.path.exists(path):
        return roof
    with open(path) as f:
        tr = json.load(f)
    if tr.get("kernel_source_sha") == kernel_source_sha():
        roof["traffic"] = tr.get("hbm_bytes_per_encode")
        roof["traffic_source"] = f"{os.path.relpath(path, ROOT)} (separate rocprofv3 --pmc passes, {tr.get('source')})"
    else:
        roof["traffic_note"] = "kernel sources changed since the PMC pass in profiles/: not reported"
    return roof


def copy_bandwidth(torch):
    """float32 copy of 1 GiB (read + write = 2 GiB of traffic) with torch's vectorised copy kernel."""
    a = torch.empty(256 << 20, dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2 * a.numel() * 4 / (e0.elapsed_time(e1) / 5 * 1e-3) / 1e9


def dev_records(torch, synth, ev):
    return torch.from_numpy(synth.to_dat8(ev).view(np.uint8).reshape(-1, 8)).cuda()


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n, argv):
    """``python bench.py --gpus N`` without a launcher around it (no WORLD_SIZE in the environment): start N FRESH child
    processes of this script, one rank per GPU (the reference's launch, README.md:160-170: one process per GPU with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), forward rank 0's JSON line, and return non-zero when any
    rank fails.  This parent never touches HIP (it does not even import torch): nothing re-execs or forks a process
    that has initialised the GPU."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=os.environ.get("MASTER_PORT") or str(free_port()), FRLW_BENCH_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    rc, out0 = 0, b""
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                p = procs[r]
                if r == 0 and p.stdout is not None:  # rank 0 prints one line at the very end; communicate() drains it
                    try:
                        o, _ = p.communicate(timeout=0.2)
                        out0 += o or b""
                    except subprocess.TimeoutExpired:
                        continue
                elif p.poll() is None:
                    continue
                pending.discard(r)
                if p.returncode != 0:
                    rc = rc or p.returncode or 1
                    print(f"bench.py: rank {r} exited with code {p.returncode}", file=sys.stderr)
            if rc:
                break
            time.sleep(0.05)
    finally:
        for p in procs:  # exactly the children started here, by PID
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    return rc


def claim_stdout():
    """The JSON line must be the ONLY thing this process writes to stdout: RCCL prints ``Librccl path : ...`` there from native
    code when its library goes away -- after the line, and from every rank.  From here on file descriptor 1 is stderr for
    everybody (native libraries included); the returned file is the real stdout, for rank 0's one line."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    return os.fdopen(real, "w")


def launch_only(args):
    """--launch-only: the rendezvous of a --gpus N run and nothing else (tests/test_dist_cpu.py runs it with
    FRLW_DIST_BACKEND=gloo on CPU): every rank joins the process group, the ranks are summed and the barrier-bracketed
    MAX-over-ranks reduction of bench.py is exercised; rank 0 prints one JSON line."""
    import torch.distributed as dist
    from frlw_evd_amd import dist as fd
    if os.environ.get("FRLW_BENCH_TEST_FAIL_RANK") == os.environ.get("RANK", "0"):  # the launcher's failure path, under test
        raise SystemExit(3)
    out = claim_stdout()
    rank, world, local_rank = fd.init_from_env(None, args.local_rank)
    fd.barrier_sync()
    ranks = fd.sum_over_ranks([rank + 1])[0]
    slowest = fd.max_over_ranks([0.5 + rank])[0]
    fd.barrier_sync()
    if rank == 0:
        print(file=out, flush=True, *[json.dumps({"launch_only": True, "n_gpus": world, "gpus_flag": args.gpus, "rank_sum": ranks,
                          "max_over_ranks": slowest, "backend": dist.get_backend() if dist.is_initialized() else None,
                          "launched_by": "bench.py" if os.environ.get("FRLW_BENCH_LAUNCHED") else "external launcher"})])
    if dist.is_initialized():
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node (default: WORLD_SIZE, else 1)")
    ap.add_argument("--launch-only", action="store_true", help="rendezvous of the N ranks only (CPU test of the launcher)")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="taf_mpx", choices=sorted(WORKLOADS))
    ap.add_argument("--hotspot", action="store_true", help="25 %% of the events in a sigma-8px blob")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-detector", action="store_true", help="skip the detector forward leg")
    ap.add_argument("--det-batch", type=int, default=32)
    ap.add_argument("--no-train", action="store_true", help="skip the train-step leg")
    ap.add_argument("--no-also", action="store_true", help="skip the other encoder workloads (clean per-kernel profiles)")
    ap.add_argument("--train-batch", type=int, default=64, help="per-GPU batch of the train-step leg")
    ap.add_argument("--local_rank", "--local-rank", type=int, default=None)
    args = ap.parse_args()
    if args.gpus is None:
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: be the launcher (before anything imports torch or touches HIP in this process)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")
    if args.launch_only:
        return launch_only(args)

    out = claim_stdout()
    import torch
    import torch.distributed as dist

    if os.environ.get("FRLW_DIST_BACKEND", "nccl") == "nccl" and args.gpus > torch.cuda.device_count():
        raise SystemExit(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs, this box has {torch.cuda.device_count()} "
                         "(one process per GPU over RCCL)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    from frlw_evd_amd import _lib, synth
    from frlw_evd_amd import dist as fd
    from frlw_evd_amd import event_representation as er
    rank, world, local_rank = fd.init_from_env("nccl", args.local_rank)  # one process per GPU, RCCL
    torch.cuda.set_device(local_rank)
    n_gpus = world
    _lib.load()
    # every rank runs the lane-order self-test of the fast paths on its own GPU; the job takes them only if ALL ranks passed
    # (dist.agree_fast_path: a mixed world would have one rank on the general path and everybody else waiting for it)
    fast_ok = fd.agree_fast_path()
    if not fast_ok:  # (never seen on gfx950; the batched rows have no general-path form)
        args.no_also = True
    timer = Timer(torch, fd, er)
    copy_gbs = copy_bandwidth(torch)

    seed, n, H, W, t_span, n_win, win_us, K = WORKLOADS[args.workload]
    ev = synth.synth_events(seed + 7919 * rank, n, W, H, t_span, hotspot=args.hotspot)
    dat_h = synth.to_dat8(ev)
    dat = torch.from_numpy(dat_h.view(np.uint8).reshape(-1, 8)).cuda()
    state = torch.full((H, W, 2, K), -6000.0, device="cuda")
    use_fast = n >= er.FAST_MIN_EVENTS and fast_ok

    def step(fast=use_fast):
        return er.encode_taf_dat(dat, (H, W), state, 0, win_us, n_win, K, want_view=False, want_u8=True, flip_k=True,
                                 check=False, fast=fast)

    # correctness guard: data-dependent status of the first encode must be clean (and the fast path must have taken it)
    if use_fast:
        er.encode_taf_batch(dat, [0, n], (H, W), state.view(1, H, W, 2, K), 0, win_us, n_win, K, check=True)
    else:
        er.encode_taf_dat(dat, (H, W), state, 0, win_us, n_win, K, check=True, fast=False)
    per_step, dev_ms = timer.run(step, args.steps, args.warmup)
    alg_bytes = taf_algorithmic_bytes(n, H, W, K)
    kernels = fast_kernel_label(er) if use_fast else \
        "frlw_taf_encode = k_hist + k_slabscan + k_tilescan + k_scatter + k_taf_tile (dominant: k_taf_tile)"
    result = {
        "metric": "TAF encode throughput (Mevents/s)",
        "value": round(n_gpus * n / per_step / 1e6, 2),
        "unit": "Mevents/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(per_step * 1e3, 4),
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
            "path": "fast (csrc/taf_fast.hip)" if use_fast else "general (csrc/encoders.hip)",
            "events_per_step_per_gpu": n,
            "parallelism": f"sequence-sharded x{n_gpus} (no collective)",
        },
        "roofline": roofline(alg_bytes, dev_ms, kernels, copy_gbs, f"{n} events"),
    }
    attach_traffic(result["roofline"], f"{args.workload}{'_hotspot' if args.hotspot else ''}")
    if use_fast:  # the general path on the same workload, for comparison
        st2 = state.clone()
        per2, dev2 = timer.run(lambda: er.encode_taf_dat(dat, (H, W), st2, 0, win_us, n_win, K, check=False, fast=False),
                               max(5, args.steps // 5), 2)
        result["general_path"] = {"value": round(n_gpus * n / per2 / 1e6, 2), "unit": "Mevents/s", "device_ms": round(dev2, 4)}

    if world > 1 and args.workload == "taf_mpx":
        # SURVEY.md 8(e), second form: ONE stream sharded spatially -- every rank holds the same 10 M-event stream and encodes its
        # stripe of rows (events routed by y on the device, own stripe of state / output); the only exchange is the OR of the
        # 64-bit window masks (generate_taf.py:40-41 is a per-frame rule) between the two halves of the encode.  Strong scaling
        # of one encode: the partition still reads the whole stream on every rank.
        ev_s = synth.synth_events(seed, n, W, H, t_span)  # the SAME stream on every rank
        dat_s = torch.from_numpy(synth.to_dat8(ev_s).view(np.uint8).reshape(-1, 8)).cuda()
        lo, hi = fd.shard_range(H, rank, world)
        st_s = torch.full((1, hi - lo, W, 2, K), -6000.0, device="cuda")
        er.encode_taf_stripe(dat_s, [0, n], (H, W), (lo, hi), st_s, 0, win_us, n_win, K, check=True)
        per_s, dev_s = timer.run(lambda: er.encode_taf_stripe(dat_s, [0, n], (H, W), (lo, hi), st_s, 0, win_us, n_win, K, check=False),
                                 max(5, args.steps // 2), 2)
        result["stripe_sharding"] = {"workload": f"ONE {n}-event {W}x{H} stream, rows sharded over {world} GPUs (encode_taf_stripe: window "
                                                 "masks OR-reduced over RCCL between partition and walk)", "scaling": "strong",
                                     "value": round(n / per_s / 1e6, 2), "unit": "Mevents/s", "ms_per_step": round(per_s * 1e3, 4),
                                     "rows_of_rank0": [lo, hi]}
        del dat_s, st_s
    if args.workload == "taf_mpx" and not args.no_also:
        result["also"] = bench_also(args, torch, synth, er, timer, rank, n_gpus, copy_gbs)
        # the shape BASELINE.json's metric names (GEN1 304x240), promoted: a block of its own and a compact copy inside
        # `roofline` (the keys the driver keeps when it parses the line)
        for row, tag in zip(result["also"], ("taf_gen1", "taf_gen1_x64", "ev_gen1", "ev_gen1_x64", "sae_gen1", "eci_gen1",
                                             "taf_mpx_hotspot")):
            attach_traffic(row["roofline"], tag)
        names = ("taf_single", "taf_x64", "ev_single", "ev_x64")
        result["gen1"] = {k: row for k, row in zip(names, result["also"][:4])}
        # the driver's parser keeps SCALAR keys of `roofline` only: one flat key per number
        flat = result["roofline"]

        def flat_row(prefix, row):
            rf = row["roofline"]
            if "graph_ms_per_step" in row:  # small single-stream encodes: the graph replay and the eager calls, a key each
                flat[f"{prefix}_graph_ms"] = row["graph_ms_per_step"]
                flat[f"{prefix}_eager_ms"] = row["ms_per_step"]
            else:
                flat[f"{prefix}_ms"] = row["ms_per_step"]
            if "default_call_ms" in row:
                flat[f"{prefix}_default_call_ms"] = row["default_call_ms"]
            flat[f"{prefix}_frac"] = rf["frac"]
            if rf.get("traffic"):
                flat[f"{prefix}_traffic_x"] = round(rf["traffic"] / rf["algorithmic_bytes"], 3)
        for k, row in result["gen1"].items():
            flat_row(f"gen1_{k}", row)
        for row in result["also"][4:]:
            if row.get("tag"):
                flat_row(row["tag"], row)
    if "general_path" in result:
        result["roofline"]["general_path_ms"] = result["general_path"]["device_ms"]
    if "stripe_sharding" in result:
        result["roofline"]["stripe_ms"] = result["stripe_sharding"]["ms_per_step"]
    if not args.no_detector:
        result["detector"] = bench_detector(args, torch, world, rank, timer)
        d = result["detector"]  # the second half of BASELINE.json's metric, in the keys the driver keeps (scalars)
        flat = result["roofline"]
        flat["detector_frames_per_s"] = d["value"]
        flat["detector_ms_per_batch"] = d["ms_per_batch"]
        flat["detector_batch_per_gpu"] = d["batch_per_gpu"]
        flat["detector_TFLOPs"] = d["roofline"]["achieved"]
        flat["detector_frac"] = d["roofline"]["frac"]
        flat["detector_dtype"] = d["dtype"]
        if "other_arithmetic" in d:  # the opt-in bf16x3 arithmetic (or, when that was made the default, the float32 MFMA)
            o = d["other_arithmetic"]
            tagp = "bf16x3" if o["precision"] == "bf16x3" else "f32mfma"
            flat[f"detector_{tagp}_frames_per_s"] = o["value"]
            flat[f"detector_{tagp}_ms_per_batch"] = o["ms_per_batch"]
            flat[f"detector_{tagp}_frac"] = o["roofline"]["frac"]
            if "x_f32_mfma_peak" in o["roofline"]:
                flat[f"detector_{tagp}_x_f32_mfma_peak"] = o["roofline"]["x_f32_mfma_peak"]
            flat["detector_bf16x3_vs_f32_max_rel_diff"] = o["max_rel_diff_bf16x3_vs_f32"]
        flat["detector_fwd_nms_ms"] = d.get("fwd_plus_decode_nms_ms")
        flat["detector_fwd_nms_typical_ms"] = d.get("fwd_plus_decode_nms_typical_ms")
        if "shape_1mpx" in d:
            flat["detector_1mpx_frames_per_s"] = d["shape_1mpx"]["value"]
            flat["detector_1mpx_frac"] = d["shape_1mpx"]["roofline"]["frac"]
            flat["detector_1mpx_ms_per_batch"] = d["shape_1mpx"]["ms_per_batch"]
            flat["detector_1mpx_fwd_nms_ms"] = d["shape_1mpx"].get("fwd_plus_decode_nms_ms")
    if not args.no_train:
        try:
            result["train"] = bench_train(args, torch, world, rank, local_rank, timer)
            t = result["train"]
            flat = result["roofline"]
            flat["train_frames_per_s"] = t["value"]
            flat["train_ms"] = t["ms_per_step"]
            flat["train_TFLOPs"] = t["roofline"]["achieved"]
            flat["train_frac"] = t["roofline"]["frac"]
            flat["train_dtype"] = t["dtype"]
            if "other_arithmetic" in t:
                o = t["other_arithmetic"]
                tagp = "bf16x3" if o["precision"] == "bf16x3" else "f32mfma"
                flat[f"train_{tagp}_ms"] = o["ms_per_step"]
                flat[f"train_{tagp}_frames_per_s"] = o["value"]
                flat[f"train_{tagp}_frac"] = o["roofline"]["frac"]
                if "x_f32_mfma_peak" in o["roofline"]:
                    flat[f"train_{tagp}_x_f32_mfma_peak"] = o["roofline"]["x_f32_mfma_peak"]
            if "encode_plus_train_step" in t:
                flat["encode_plus_train_ms"] = t["encode_plus_train_step"]["ms_per_step"]
                flat["encode_plus_train_frames_per_s"] = t["encode_plus_train_step"]["value"]
                flat["encode_ahead_plus_train_ms"] = t["encode_plus_train_step"]["encode_ahead"]["ms_per_step"]
                flat["encode_ahead_plus_train_frames_per_s"] = t["encode_plus_train_step"]["encode_ahead"]["value"]
            if "eager" in t:
                flat["train_eager_ms"] = t["eager"]["ms_per_step"]
            if "same_step_with_miopen_convs" in t:
                flat["train_miopen_ms"] = t["same_step_with_miopen_convs"]["ms_per_step"]
            if "allreduce" in t:
                flat["allreduce_exposed_ms"] = t["allreduce"]["allreduce_exposed_ms"]
            if "global64" in t:
                flat["global64_ms"] = t["global64"]["ms_per_step"]
        except Exception as e:  # never lose the headline line to the extra leg
            result["train"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline_taf(dat_h, n, H, W, K, n_win, win_us)
    if rank == 0:
        emit(result, out)
    if dist.is_initialized():
        dist.destroy_process_group()


def bench_also(args, torch, synth, er, timer, rank, n_gpus, copy_gbs):
    """The other encoder rows of SURVEY.md 8(d): the GEN1 shape BASELINE.json's metric names (single stream and 64
    sequences per launch), Event Volume cfg 2 (single and batched), and the contention variant of the headline."""
    out = []
    steps = max(5, min(args.steps, 20))
    s2, n2, H2, W2, t2, nw2, wu2, K2 = WORKLOADS["taf_gen1"]
    # ---- TAF, GEN1 shape, one stream (launch / latency bound at this size)
    ev2 = synth.synth_events(s2 + 7919 * rank, n2, W2, H2, t2)
    rec2 = synth.to_dat8(ev2)
    dat2 = torch.from_numpy(rec2.view(np.uint8).reshape(-1, 8)).cuda()
    st2 = torch.full((H2, W2, 2, K2), -6000.0, device="cuda")
    per, dev, eager = run_graphed(timer, lambda: er.encode_taf_dat(dat2, (H2, W2), st2, 0, wu2, nw2, K2, check=False,
                                                                    fast=n2 >= er.FAST_MIN_EVENTS), steps, 3)
    row = {"workload": "taf_gen1 (the GEN1 304x240 shape BASELINE.json's metric names): TAF K=8 encode + leaky + uint8, "
                       "1000000 events, 304x240, 8 windows, ONE stream per launch sequence",
           "value": round(n_gpus * n2 / per / 1e6, 2), "unit": "Mevents/s", "ms_per_step": round(per * 1e3, 4),
           "roofline": roofline(taf_algorithmic_bytes(n2, H2, W2, K2), dev, "frlw_taf_encode_batch, one sequence, direct mode = kf_scatter_cm + kf_taf_walk<K8, direct> (two launches)" if n2 >= er.FAST_MIN_EVENTS else "frlw_taf_encode (k_taf_tile dominant)", copy_gbs,
                                f"{n2} events")}
    graphed_row(row, n_gpus * n2, per, eager)
    row["path"] = "encode_taf_dat(fast=True, check=False): an OPT-IN for unchecked calls (fast='auto' takes it for checked calls only)"
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        row["cpu_baseline"] = cpu_baseline_taf(rec2, n2, H2, W2, K2, nw2, wu2, all_cores=False)
    out.append(row)
    # ---- TAF, GEN1 shape, 64 sequences per launch sequence (frlw_taf_encode_batch)
    B = 64
    recs = [synth.to_dat8(synth.synth_events(s2 + 100 + j + 7919 * rank, n2, W2, H2, t2)) for j in range(B)]
    offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
    datb = torch.from_numpy(np.concatenate(recs).view(np.uint8).reshape(-1, 8)).cuda()
    del recs
    stb = torch.full((B, H2, W2, 2, K2), -6000.0, device="cuda")
    per, dev = timer.run(lambda: er.encode_taf_batch(datb, offs, (H2, W2), stb, 0, wu2, nw2, K2, check=False), steps, 3)
    out.append({"workload": f"taf_gen1 x{B}: {B} independent GEN1-shaped streams of 1000000 events in ONE launch sequence "
                            "(frlw_taf_encode_batch: own FIFO state and window rule per sequence)",
                "value": round(n_gpus * B * n2 / per / 1e6, 2), "unit": "Mevents/s", "ms_per_step": round(per * 1e3, 4),
                "roofline": roofline(B * taf_algorithmic_bytes(n2, H2, W2, K2), dev, "frlw_taf_encode_batch (kf_taf_walk dominant)",
                                     copy_gbs, f"{B} x {n2} events")})
    del datb, stb
    # ---- Event Volume cfg 2: 1 M events, 304x240, 5 bins -- one stream, then 64 streams as two launches of 32 stacked
    ev4 = synth.synth_events(1002 + 7919 * rank, 1_000_000, W2, H2, 250_000)
    rec4 = synth.to_dat8(ev4)
    dat4 = torch.from_numpy(rec4.view(np.uint8).reshape(-1, 8)).cuda()
    er.encode_ev_dat(dat4, (H2, W2), 250_000, 250_000, volume_bins=5, check=True, fast=True)  # data-dependent status: clean
    per, dev, eager = run_graphed(timer, lambda: er.encode_ev_dat(dat4, (H2, W2), 250_000, 250_000, volume_bins=5, check=False, fast=True), steps, 3)
    row = {"workload": "ev_gen1 (BASELINE.json configs[1]): Event Volume 5 bins, 1000000 events, 304x240, ONE stream",
           "value": round(n_gpus * 1_000_000 / per / 1e6, 2), "unit": "Mevents/s", "ms_per_step": round(per * 1e3, 4),
           "roofline": roofline(ev_algorithmic_bytes(1_000_000, H2, W2, 5), dev,
                                "frlw_ev_encode_batch, one window, direct mode = kf_scatter_cm + kf_ev_fadd (two launches)", copy_gbs,
                                "1000000 events")}
    graphed_row(row, n_gpus * 1_000_000, per, eager)
    row["path"] = "encode_ev_dat(fast=True, check=False): the two-launch form is an OPT-IN for unchecked calls (fast='auto' takes it for checked calls only)"
    dflt, _d = timer.run(lambda: er.encode_ev_dat(dat4, (H2, W2), 250_000, 250_000, volume_bins=5), steps, 2)
    row["default_call_ms"] = round(dflt * 1e3, 4)  # encode_ev_dat with its defaults: fast="auto", check=True (one host sync per call)
    pg, dg, _e = run_graphed(timer, lambda: er.encode_ev_dat(dat4, (H2, W2), 250_000, 250_000, volume_bins=5, check=False, fast=False), steps, 3)
    row["general_path"] = {"ms_per_step": round(pg * 1e3, 4), "device_ms": round(dg, 4), "kernel": "frlw_ev_encode (five launches)"}
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        row["cpu_baseline"] = cpu_baseline_ev(rec4, H2, W2)
    out.append(row)
    # 64 independent label windows in ONE frlw_ev_encode_batch call (csrc/taf_fast.hip: 4-byte records, the tile walk splits
    # every tile in LDS): own t_end per sequence, bit-identical to 64 frlw_ev_encode calls (tests/test_ev_batch_gpu.py)
    B = 64
    recs = [synth.to_dat8(synth.synth_events(1002 + 50 + j + 7919 * rank, 1_000_000, W2, H2, 250_000, t_offset=1)) for j in range(B)]
    offs5 = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
    dat5 = torch.from_numpy(np.concatenate(recs).view(np.uint8).reshape(-1, 8)).cuda()
    del recs
    er.encode_ev_batch(dat5, offs5, (H2, W2), 250_000, 250_000, 5, check=True)  # data-dependent status of the workload: clean
    per, dev = timer.run(lambda: er.encode_ev_batch(dat5, offs5, (H2, W2), 250_000, 250_000, 5, check=False), steps, 2)
    out.append({"workload": f"ev_gen1 x{B}: {B} independent GEN1-shaped label windows of 1000000 events in ONE launch sequence "
                            "(frlw_ev_encode_batch)",
                "value": round(n_gpus * B * 1_000_000 / per / 1e6, 2), "unit": "Mevents/s", "ms_per_step": round(per * 1e3, 4),
                "roofline": roofline(B * ev_algorithmic_bytes(1_000_000, H2, W2, 5), dev, "frlw_ev_encode_batch = kf_scatter_cm + kf_split_whole<true> + kf_segcount_cm + kf_split_place<true> + kf_ev_sub (dominant: kf_ev_sub)",
                                     copy_gbs, f"{B} x 1000000 events")})
    del dat5
    # ---- Surface of Active Events (generate_surfaceofactiveevents.py:44-80): 1 M events over 5 s, 304x240, three lambdas, the
    #      per-pixel memory carried from call to call like consecutive labels (:176-190)
    LAM = [0.00001, 0.0000025, 0.000001]
    ev6 = synth.synth_events(1006 + 7919 * rank, 1_000_000, W2, H2, 5_000_000, t_offset=30_000_000)
    rec6 = synth.to_dat8(ev6)
    dat6 = torch.from_numpy(rec6.view(np.uint8).reshape(-1, 8)).cuda()
    now6, win6 = 35_000_000, 5_541_263
    mem = {"m": er.encode_sae_dat(dat6, (H2, W2), LAM, None, now6, win6, check=True)[2]}

    def sae_step():
        mem["m"] = er.encode_sae_dat(dat6, (H2, W2), LAM, mem["m"], now6, win6, check=False)[2]
    per, dev, eager = run_graphed(timer, sae_step, steps, 3)
    row = {"tag": "sae_gen1", "workload": "sae_gen1: Surface of Active Events, 3 lambdas, 1000000 events over 5 s, 304x240, memory carried "
                                        "(frlw_sae_encode)",
           "value": round(n_gpus * 1_000_000 / per / 1e6, 2), "unit": "Mevents/s", "ms_per_step": round(per * 1e3, 4),
           "roofline": roofline(sae_algorithmic_bytes(1_000_000, H2, W2, len(LAM)), dev,
                                "frlw_sae_encode = kf_scatter_cm<SAE> + kf_sae_sub (two launches)", copy_gbs, "1000000 events")}
    graphed_row(row, n_gpus * 1_000_000, per, eager)
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        row["cpu_baseline"] = cpu_baseline_generic(
            lambda orc: orc.sae_stream_dat8(rec6, (H2, W2), (H2, W2), LAM, None, now6, win6), len(rec6), "sae_stream_dat8")
    out.append(row)
    del dat6
    # ---- Event Count Image (BASELINE.json configs[0]; generate_eventcountimage.py:19-41): 100 k events, 304x240
    ev7 = synth.synth_events(1001 + 7919 * rank, 100_000, W2, H2, 50_000)
    rec7 = synth.to_dat8(ev7)
    dat7 = torch.from_numpy(rec7.view(np.uint8).reshape(-1, 8)).cuda()
    er.encode_eci_dat(dat7, (H2, W2), check=True)
    per, dev, eager7 = run_graphed(timer, lambda: er.encode_eci_dat(dat7, (H2, W2), check=False), steps, 3)
    row = {"tag": "eci_gen1", "workload": "eci_gen1 (BASELINE.json configs[0]): Event Count Image, 100000 events, 304x240 (frlw_eci_encode)",
           "value": round(n_gpus * 100_000 / per / 1e6, 2), "unit": "Mevents/s", "ms_per_step": round(per * 1e3, 4),
           "roofline": roofline(eci_algorithmic_bytes(100_000, H2, W2), dev,
                                "frlw_eci_encode = kf_scatter_cm<ECI> + kf_sae_sub<count> (two launches)", copy_gbs, "100000 events")}
    graphed_row(row, n_gpus * 100_000, per, eager7)
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        row["cpu_baseline"] = cpu_baseline_generic(lambda orc: orc.eci_stream_dat8(rec7, (H2, W2), (H2, W2)), len(rec7),
                                                   "eci_stream_dat8", reps=40)
    out.append(row)
    del dat7
    if not args.hotspot:
        # SURVEY.md 8(d) "report both": the contention variant of the headline workload (25 % of the events in a
        # sigma = 8 px blob -> a few tiles hold most of them; skewed tiles are split by segments)
        seed, n, H, W, t_span, n_win, win_us, K = WORKLOADS["taf_mpx"]
        ev3 = synth.synth_events(seed + 7919 * rank, n, W, H, t_span, hotspot=True)
        dat3 = torch.from_numpy(synth.to_dat8(ev3).view(np.uint8).reshape(-1, 8)).cuda()
        st3 = torch.full((H, W, 2, K), -6000.0, device="cuda")
        per, dev = timer.run(lambda: er.encode_taf_dat(dat3, (H, W), st3, 0, win_us, n_win, K, check=False, fast=True), steps, 2)
        out.append({"tag": "taf_mpx_hotspot", "workload": "taf_mpx, hotspot variant (25 % of the events in a sigma = 8 px blob)",
                    "value": round(n_gpus * n / per / 1e6, 2), "unit": "Mevents/s", "ms_per_step": round(per * 1e3, 4),
                    "roofline": roofline(taf_algorithmic_bytes(n, H, W, K), dev, "frlw_taf_encode_batch", copy_gbs, f"{n} events")})
    return out


def mfma_sustained(torch):
    """What a bare v_mfma_f32_32x32x2_f32 loop sustains on this chip (random operands), TFLOP/s."""
    import ctypes as C
    from frlw_evd_amd import _lib
    lib = _lib.load()
    seed = torch.randn(256, device="cuda")
    sink = torch.zeros(4, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    blocks, iters = 256 * 5, 2000
    lib.frlw_selftest_mfma_f32_rate(blocks, iters, seed.data_ptr(), sink.data_ptr(), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        lib.frlw_selftest_mfma_f32_rate(blocks, iters, seed.data_ptr(), sink.data_ptr(), st)
    e1.record()
    torch.cuda.synchronize()
    return blocks * 4 * iters * 32 * 4096 / (e0.elapsed_time(e1) / 3) / 1e9


def bench_detector(args, torch, world, rank, timer):
    """Second half of BASELINE.json's metric: YOLOX forward frames/s (SURVEY.md 8(d) cfg 4): B = 32, (10, 256, 320) f32
    input (the detector shape of a 304x240 sensor), recipe weights, eval mode, forward to the pre-NMS tensor (B, 1680, 7);
    decode + NMS timed separately; then the 1 Mpx detector shape (10, 512, 640), 4x the FLOPs per image."""
    from frlw_evd_amd.yolox import build_yolox
    from frlw_evd_amd.yolox.model import recipe_state_dict
    out = None
    for tag, B, Hd, Wd in (("gen1", args.det_batch, 256, 320), ("1mpx", max(1, args.det_batch // 4), 512, 640)):
        net = build_yolox(10, 2 if tag == "gen1" else 7, radius=5.0 if tag == "gen1" else 2.5)
        net.load_state_dict(recipe_state_dict(net, seed=1004))
        net.eval()
        rng = np.random.default_rng(1004 + rank)
        x_h = torch.from_numpy(rng.integers(0, 256, size=(B, 10, Hd, Wd)).astype(np.float32) / np.float32(255))
        x = x_h.cuda()
        eng = net.engine()
        steps = max(5, min(args.steps, 30))
        # three timed regions of `steps` forwards each; the value is their MEDIAN (not the best: a fresh box now and then spends
        # tens of ms of one region on a clock / power transition, and the first region after the warm-up runs ~3 % slow)
        _p, _d, runs = one_region(timer, lambda: eng.raw_outputs(x), steps, 20, extra=2)
        per, dev_ms = sorted(runs)[1]
        tflops = eng.flops_per_image * B / (dev_ms * 1e-3) / 1e12
        row = {
            "value": round(world * B / per, 1), "unit": "frames/s", "batch_per_gpu": B,
            "input": f"(B, 10, {Hd}, {Wd}) f32, recipe weights", "steps": steps, "ms_per_batch": round(per * 1e3, 3),
            "dtype": "f32" if eng.precision == "f32" else "f32 storage and accumulation, products from 3 bf16 MFMAs (bf16x3)",
            "ms_per_batch_runs": [round(r[0] * 1e3, 3) for r in runs], "value_is": "the median of the three timed regions listed",
            "roofline": mfma_roofline(tflops, eng.precision, f"k_conv_mfma ({eng.n_conv} launches per forward)",
                                      flops_per_image=eng.flops_per_image, device_ms_per_batch=round(dev_ms, 3)),
        }
        if tag == "gen1":  # the same forward in the OTHER arithmetic (the opt-in bf16x3 beside the float32 default), one region
            from frlw_evd_amd.detector import DetectorEngine
            other = "bf16x3" if eng.precision == "f32" else "f32"
            e2 = DetectorEngine(net, precision=other)
            per2, dev2, _r = one_region(timer, lambda: e2.raw_outputs(x), steps, 10)
            tf2 = e2.flops_per_image * B / (dev2 * 1e-3) / 1e12
            ref32, alt = (eng, e2) if other == "bf16x3" else (e2, eng)
            row["other_arithmetic"] = {"precision": other, "value": round(world * B / per2, 1), "unit": "frames/s",
                                       "ms_per_batch": round(per2 * 1e3, 3),
                                       "roofline": mfma_roofline(tf2, other, "k_conv_mfma", device_ms_per_batch=round(dev2, 3)),
                                       "max_rel_diff_bf16x3_vs_f32": float((alt.raw_outputs(x) - ref32.raw_outputs(x)).abs().max()
                                                                           / ref32.raw_outputs(x).abs().max()),
                                       "speedup_vs_default": round(per / per2, 3)}
            del e2
        def time_detect(engine):
            """(wall ms of the full eval forward incl. decode + NMS and the host's read of the counts, device ms of the decode +
            NMS launches alone)."""
            for _ in range(2):
                engine.detect(x)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                engine.detect(x)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t1) / 5 * 1e3
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                engine._run(x, engine.n_forward_ops, -1)
            e1.record()
            torch.cuda.synchronize()
            return round(wall, 3), round(e0.elapsed_time(e1) / 5, 3)
        raw = eng.raw_outputs(x)
        row["nms_candidates_per_image"] = round(float((raw[:, :, 4] > net.head.obj_threshold).sum(1).float().mean()), 1)
        row["fwd_plus_decode_nms_ms"], row["decode_nms_device_ms"] = time_detect(eng)  # recipe weights: close to the worst case
        # ... and with a realistic candidate count: the recipe's objectness bias shifted so that ~100 anchors per image pass the
        # threshold (a trained network's regime; random weights put most anchors above 0.3)
        with torch.no_grad():
            obj = raw[:, :, 4].clamp(1e-6, 1 - 1e-6)
            kth = torch.logit(obj).topk(min(100, obj.shape[1]), dim=1).values[:, -1].median()
            shift = float(kth - torch.logit(torch.tensor(float(net.head.obj_threshold))))
            saved = [p_.bias.clone() for p_ in net.head.obj_preds]
            for p_ in net.head.obj_preds:
                p_.bias.sub_(shift)
        eng_t = net.engine()
        raw_t = eng_t.raw_outputs(x)
        row["nms_candidates_per_image_typical"] = round(float((raw_t[:, :, 4] > net.head.obj_threshold).sum(1).float().mean()), 1)
        row["fwd_plus_decode_nms_typical_ms"], row["decode_nms_typical_device_ms"] = time_detect(eng_t)
        with torch.no_grad():
            for p_, v_ in zip(net.head.obj_preds, saved):
                p_.bias.copy_(v_)
        del eng_t
        if tag == "gen1":
            out = dict({"metric": "YOLOX-S (CSPDarknet + PAFPN + decoupled head) eval forward to the pre-NMS tensor "
                                  "(BASELINE.json configs[3])"}, **row)
            if rank == 0 and world == 1 and not args.no_cpu_baseline:
                # PyTorch-CPU forward of the same module definition: all cores and 1
                cpu = {}
                xb = x_h[..., None]
                many_t = sorted({physical_cores()[0], min(32, physical_cores()[0])}, reverse=True)
                for nthreads in many_t + [1]:  # (all physical cores; 32 threads: torch's CPU convolutions stop scaling early)
                    torch.set_num_threads(nthreads)
                    xs = xb if nthreads > 1 else xb[:1]
                    with torch.no_grad():
                        net.reference_outputs(xs)
                        best = None
                        for _ in range(3 if nthreads > 1 else 1):
                            t2 = time.perf_counter()
                            net.reference_outputs(xs)
                            dt = time.perf_counter() - t2
                            best = dt if best is None else min(best, dt)
                    cpu[nthreads] = (len(xs) / best, best, len(xs))
                many = max((k for k in cpu if k > 1), key=lambda k: cpu[k][0], default=1)  # the thread count that did best
                out["cpu_baseline"] = {"value": round(cpu[many][0], 1), "unit": "frames/s", "cores": many, "kind": "port",
                                       "frames_per_s_by_threads": {str(k): round(v[0], 2) for k, v in cpu.items()},
                                       "host_physical_cores": physical_cores()[0],
                                       "one_thread_frames_per_s": round(cpu[1][0], 2), "cpu": cpu_model(),
                                       "sample": f"PyTorch-CPU fp32 forward of the same modules, batch {cpu[many][2]} on {many} threads "
                                                 f"(best of 3, {cpu[many][1]:.3f} s), batch 1 on 1 thread ({cpu[1][1]:.3f} s)"}
        else:
            out["shape_1mpx"] = dict({"workload": "the 1 Mpx detector shape of SURVEY.md 8(d) cfg 4: (B, 10, 512, 640), 7 classes, "
                                                  "6720 anchors, 4x the FLOPs per image"}, **row)
        del eng, net, x
    # the bare v_mfma_f32_32x32x2_f32 loop of this box (measured AFTER the forwards: it is the chip's highest power draw)
    sustained = mfma_sustained(torch)
    for r in (out["roofline"], out["shape_1mpx"]["roofline"]):
        r["bare_f32_mfma_loop_TFLOPs"] = round(sustained, 1)  # what the float32 instruction sustains on this box
        r["x_bare_f32_mfma_loop"] = round(r["achieved"] / sustained, 4)
    return out


def _train_inputs(torch, B, rank):
    rng = np.random.default_rng(1005 + rank)
    x = torch.from_numpy(rng.integers(0, 256, size=(B, 16, 256, 320, 1, 1), dtype=np.uint8)).float().div(255).cuda()
    lab = torch.zeros(B, 80, 5, dtype=torch.float64)
    lab[:, 0] = torch.tensor([0, 100, 90, 60, 40.0])
    lab[:, 1] = torch.tensor([1, 220, 150, 50, 80.0])
    return x, lab.cuda()


def _train_variant(torch, timer, world, rank, local_rank, per_gpu_batch, ddp, hook, steps=5, warmup=3, graph=False):
    """One configuration of the train step on a fresh model: (seconds per step, device ms per step, Trainer, last loss).
    graph: the whole step replayed as one HIP graph (one rank without DDP; captured during the warm-up calls)."""
    from frlw_evd_amd import e2e
    from frlw_evd_amd.trainer import Trainer
    m = e2e.build_model(in_channels=16, num_classes=2)
    tr = Trainer(m, global_batch=per_gpu_batch * world, nodes=world, iters_per_epoch=100, local_rank=local_rank,
                 ddp=ddp, comm_hook=hook, graph=graph)
    x, lab = _train_inputs(torch, per_gpu_batch, rank)
    state = {"i": 0, "loss": None}

    def one():
        state["loss"], _ = tr.train_step(x, lab, state["i"])
        state["i"] += 1

    if graph:  # capture, then feed the step from the graph's own input buffers like the eager step reads its batch in place
        one()
        bx, bl = tr.input_buffers()
        bx.copy_(x)
        bl.copy_(lab)
        x, lab = bx, bl
    per, dev_ms, _runs = one_region(timer, one, steps, warmup)
    return per, dev_ms, tr, state, one, (x, lab)


def _allreduce_rows(torch, timer, world, rank, local_rank, per_gpu_batch, per_main, main_is_two_graph, steps):
    """How much of the gradient exchange is exposed (not hidden behind the backward), per form of the step: the step under
    DDP with each communication hook minus the same eagerly launched step on the bare module (same kernels, no gradient
    hooks, no collective); the two-graph step (``per_main`` when the main row ran it) minus the one-graph step of a lone rank."""
    per_ns, _, tr_ns, *_ = _train_variant(torch, timer, world, rank, local_rank, per_gpu_batch, False, None, steps, 2)
    grad_bytes = int(sum(p.numel() for p in tr_ns.model.parameters()) * 4)
    del tr_ns
    if main_is_two_graph:
        per_default, _, tr_d, *_ = _train_variant(torch, timer, world, rank, local_rank, per_gpu_batch, True, None, steps, 3)
        del tr_d
        per_ng, _, tr_ng, *_ = _train_variant(torch, timer, world, rank, local_rank, per_gpu_batch, False, None, steps, graph=True)
        del tr_ng
    else:
        per_default, per_ng = per_main, None
    per_rs, _, tr_rs, *_ = _train_variant(torch, timer, world, rank, local_rank, per_gpu_batch, True, "rs_ag", steps, 3)
    del tr_rs
    per_t, _, tr_t, *_ = _train_variant(torch, timer, world, rank, local_rank, per_gpu_batch, True, "timed", steps, 3)
    hook = tr_t.comm_hook.summary() if tr_t.comm_hook is not None else {}
    del tr_t
    torch.cuda.empty_cache()
    exposed = {"default": round(max(0.0, (per_default - per_ns) * 1e3), 3), "rs_ag": round(max(0.0, (per_rs - per_ns) * 1e3), 3)}
    by_hook = {"default": round(per_default * 1e3, 3), "rs_ag": round(per_rs * 1e3, 3), "timed": round(per_t * 1e3, 3)}
    if main_is_two_graph:
        exposed["two_graph"] = round(max(0.0, (per_main - per_ng) * 1e3), 3)
        by_hook["two_graph"] = round(per_main * 1e3, 3)
    out = {"gradient_bytes": grad_bytes, "step_ms_without_collective": round(per_ns * 1e3, 3),
           "graph_step_ms_without_collective": None if per_ng is None else round(per_ng * 1e3, 3),
           "allreduce_exposed_ms": exposed["two_graph" if main_is_two_graph else "default"],
           "exposed_ms_by_hook": exposed, "step_ms_by_hook": by_hook,
           "bucket_bytes": hook.get("bucket_bytes"), "mean_bucket_ms": round(hook.get("mean_bucket_ms", 0.0), 3),
           "buckets_per_step": (hook["buckets"] // (steps + 3)) if hook.get("buckets") else None,
           "collective_ms_per_step": round(hook.get("total_ms", 0.0) / max(steps + 3, 1), 3),
           "hooks": "two_graph = Trainer(ddp=True, graph=True): forward + backward as one HIP graph, ONE all-reduce of the flat "
                    "gradient buffer launched eagerly, Adam as a second graph (nothing of the exchange is hidden: the whole of it "
                    "is exposed, the ~1 200 launches of the step are not); default = DistributedDataParallel's bucketed RCCL "
                    "all-reduce behind eager launches; rs_ag = reduce-scatter + all-gather per bucket (frlw_evd_amd.dist); "
                    "timed = default + HIP events around each bucket's collective",
           "ddp": "gradient_as_bucket_view, static_graph, 25 MB buckets (frlw_evd_amd.dist.ddp_kwargs)"}
    return out


def bench_train(args, torch, world, rank, local_rank, timer):
    """SURVEY.md 8(d) cfg 5: YOLOX (16-channel TAF input) train step -- forward, batched SimOTA assignment
    (frlw_simota_assign) + losses, backward, Adam -- under DDP over RCCL when N > 1.  Two rows: per-GPU batch fixed at 64
    (weak scaling, global 64 * N) and the reference's own semantics, global batch 64 = 64 / N per GPU (settings.py:41;
    strong scaling, ``global64``).  Every BaseConv runs forward and backward in the gfx950 kernels of csrc/train_ops.hip;
    the same step with torch autograd / MIOpen convolutions is timed beside it at N = 1."""
    from frlw_evd_amd import e2e
    B = args.train_batch
    steps = 10
    # one rank: forward + SimOTA + losses + backward + Adam replayed as ONE HIP graph; DDP ranks: two graphs around one
    # eagerly launched all-reduce of the flat gradient buffer (Trainer(ddp=True, graph=True)); FRLW_BENCH_DDP_EAGER=1 (or a
    # capture that fails) keeps the DistributedDataParallel step with eager launches
    use_graph = world == 1 or os.environ.get("FRLW_BENCH_DDP_EAGER") != "1"
    capture_error = None
    try:
        per, dev_ms, tr, state, one, (x, lab) = _train_variant(torch, timer, world, rank, local_rank, B, world > 1, None, steps,
                                                               graph=use_graph)
    except Exception as e:  # noqa: BLE001 -- (no collective runs inside the capture: a failure there is the same on every rank)
        if world == 1 or not use_graph:
            raise
        capture_error = f"{type(e).__name__}: {e}"[:300]
        use_graph = False
        torch.cuda.empty_cache()
        per, dev_ms, tr, state, one, (x, lab) = _train_variant(torch, timer, world, rank, local_rank, B, True, None, steps)
    loss = state["loss"]
    from frlw_evd_amd.detector import DetectorEngine
    from frlw_evd_amd.yolox import train_ops as _tops
    precision = {v: k for k, v in _tops.PRECISIONS.items()}[_tops.conv_precision()]
    probe = DetectorEngine(e2e.build_model(16, 2, device="cpu").eval(), device="cpu")
    probe.build((16, 256, 320))  # the plan builder counts the convolution MACs of the 16-channel network
    # forward + data gradient + weight gradient of every convolution, EXCEPT the data gradient of the stem: the network input
    # needs no gradient (yolox/train_ops.py: needs_input_grad[0] is false there) and none is computed
    stem_fl = next(m[4] for m in probe.ops_meta if m[0] in ("fstem", "conv"))
    flops_img = 3 * probe.flops_per_image - stem_fl
    flops = flops_img * B
    tflops = flops / (dev_ms * 1e-3) / 1e12
    out = {"metric": "YOLOX train step (frames/s)", "value": round(world * B / per, 1), "unit": "frames/s",
           "ms_per_step": round(per * 1e3, 3), "per_gpu_batch": B, "global_batch": B * world, "steps": steps,
           "loss": round(loss, 4), "parallelism": f"ddp{world}" if world > 1 else "single", "scaling": "weak",
           "launch": ("one HIP graph per step (Trainer(graph=True): ~1 200 kernel nodes, the loss read back after every replay "
                      "like core/exp.py:303)" if world == 1 else
                      "two HIP graphs per step around one eagerly launched RCCL all-reduce of the flat gradient buffer "
                      "(Trainer(ddp=True, graph=True))") if use_graph else "eager launches (DistributedDataParallel ranks)",
           "convolutions": "csrc/train_ops.hip (MFMA fwd / dgrad / wgrad, BatchNorm + SiLU fwd / bwd), SimOTA + losses csrc/simota.hip",
           "dtype": "f32" if precision == "f32" else "f32 storage and accumulation, products from 3 bf16 MFMAs (bf16x3)",
           "roofline": mfma_roofline(tflops, precision, "k_conv_mfma (fwd + dgrad) + k_wgrad_mfma", flops_per_step=flops,
                                     flops_model="(3 x conv MACs x 2 of the 16-channel forward - the stem's data gradient, which "
                                     "is never computed: the network input needs none) x batch = forward + data gradient + weight gradient",
                                     stem_fraction_of_forward=round(stem_fl / probe.flops_per_image, 4), device_ms_per_step=round(dev_ms, 3))}
    if capture_error:
        out["graph_capture_error"] = capture_error
    if world > 1:
        out["allreduce"] = _allreduce_rows(torch, timer, world, rank, local_rank, B, per, use_graph, steps)
        if use_graph:  # the DistributedDataParallel step with eager launches beside it (what rounds 1-5 ran on N > 1 ranks)
            e_ms = out["allreduce"]["step_ms_by_hook"]["default"]
            out["eager"] = {"value": round(world * B / (e_ms * 1e-3), 1), "ms_per_step": e_ms, "graph_speedup": round(e_ms / (per * 1e3), 3)}
    # BASELINE.json configs[4]: the same step fed by the TAF encode of its batch (B GEN1-shaped streams of 8 x 125 000
    # events -> frlw_taf_encode_batch -> uint8 -> nearest 256x320 -> /255), everything on this GPU
    src = e2e.SyntheticTafSource(B, seed=1005 + 1000 * rank)
    idx = list(range(B))

    def e2e_step():
        state["loss"], _ = tr.train_step(src.encode_batch(idx), lab, state["i"])
        state["i"] += 1
    per_e2e, _, _r = one_region(timer, e2e_step, steps, 2)
    per_enc, _ = timer.run(lambda: src.encode_batch(idx), steps, 1)
    # the same with the encode of batch i + 1 on its own stream while step i runs (e2e.EncodeAhead): every step still
    # consumes a batch that was encoded for it inside the timed region
    ahead = e2e.EncodeAhead(src)
    ahead.start(idx)

    def e2e_overlapped():
        state["loss"], _ = tr.train_step(ahead.take(), lab, state["i"], after_launch=lambda: ahead.start(idx))
        state["i"] += 1
    per_ovl, _, _r = one_region(timer, e2e_overlapped, steps, 2)
    del ahead
    out["encode_plus_train_step"] = {"workload": "BASELINE.json configs[4]: TAF encode of the batch (8 x 125 000 events per "
                                                 "304x240 sample, one frlw_taf_encode_batch call) + train step",
                                     "value": round(world * B / per_e2e, 1), "unit": "frames/s",
                                     "ms_per_step": round(per_e2e * 1e3, 3), "encode_ms_per_batch": round(per_enc * 1e3, 3),
                                     "encode_ahead": {"value": round(world * B / per_ovl, 1), "ms_per_step": round(per_ovl * 1e3, 3),
                                                      "what": "batch i + 1 encoded on a second HIP stream while step i runs "
                                                              "(frlw_evd_amd.e2e.EncodeAhead)"}}
    del src
    if world == 1:
        # the same step launched eagerly (what the DDP ranks do), and eagerly with torch autograd / MIOpen convolutions
        per_e, dev_e, tr_e, _st, one_e, _xy = _train_variant(torch, timer, world, rank, local_rank, B, False, None, steps)
        out["eager"] = {"value": round(B / per_e, 1), "ms_per_step": round(per_e * 1e3, 3), "device_ms_per_step": round(dev_e, 3),
                        "graph_speedup": round(per_e / per, 3)}
        prev = os.environ.get("FRLW_NATIVE_TRAIN")
        os.environ["FRLW_NATIVE_TRAIN"] = "0"
        try:
            per_t, _, _r = one_region(timer, one_e, steps, 3)
        finally:
            if prev is None:
                os.environ.pop("FRLW_NATIVE_TRAIN", None)
            else:
                os.environ["FRLW_NATIVE_TRAIN"] = prev
        out["same_step_with_miopen_convs"] = {"value": round(world * B / per_t, 1), "ms_per_step": round(per_t * 1e3, 3),
                                              "launch": "eager", "native_speedup": round(per_t / per_e, 3),
                                              "native_graph_speedup": round(per_t / per, 3)}
        del tr_e, one_e, _xy
        # the same graph-replayed step in the OTHER arithmetic (the opt-in bf16x3 beside the float32 default)
        other = "bf16x3" if precision == "f32" else "f32"
        prev = os.environ.get("FRLW_CONV_PRECISION")
        os.environ["FRLW_CONV_PRECISION"] = other
        try:
            per2, dev2, tr2, st2, *_ = _train_variant(torch, timer, world, rank, local_rank, B, False, None, steps, graph=True)
        finally:
            if prev is None:
                os.environ.pop("FRLW_CONV_PRECISION", None)
            else:
                os.environ["FRLW_CONV_PRECISION"] = prev
        out["other_arithmetic"] = {"precision": other, "value": round(B / per2, 1), "unit": "frames/s", "ms_per_step": round(per2 * 1e3, 3),
                                   "loss": round(st2["loss"], 4),
                                   "roofline": mfma_roofline(flops / (dev2 * 1e-3) / 1e12, other, "k_conv_mfma + k_wgrad_mfma",
                                                             device_ms_per_step=round(dev2, 3)),
                                   "speedup_vs_default": round(per / per2, 3)}
        del tr2
    del tr, one, x, lab
    torch.cuda.empty_cache()
    # ---- the reference's semantics: GLOBAL batch 64 (settings.py:41: 64 / nodes per GPU) -> strong scaling over N
    G = 64
    if G % world == 0:
        b = G // world
        if world == 1 and b == B:
            g64 = {"value": out["value"], "ms_per_step": out["ms_per_step"], "device_ms_per_step": round(dev_ms, 3),
                   "note": "N = 1: the same step as the weak row"}
        else:
            per_g, dev_g, tr_g, *_ = _train_variant(torch, timer, world, rank, local_rank, b, world > 1, None, steps,
                                                    graph=use_graph)
            del tr_g
            g64 = {"value": round(G / per_g, 1), "ms_per_step": round(per_g * 1e3, 3), "device_ms_per_step": round(dev_g, 3)}
            tf = flops_img * b / (dev_g * 1e-3) / 1e12
            g64["roofline"] = mfma_roofline(tf, precision, "k_conv_mfma + k_wgrad_mfma")
            if world > 1:
                g64["allreduce"] = _allreduce_rows(torch, timer, world, rank, local_rank, b, per_g, use_graph, steps)
        out["global64"] = dict({"workload": f"global batch {G} over {world} GPU(s) = {b} per GPU (settings.py:41), DDP "
                                            "(core/exp.py:391)", "unit": "frames/s", "per_gpu_batch": b, "global_batch": G,
                                "scaling": "strong"}, **g64)
    return out


def _best_of(fn, budget_s, max_runs=3):
    best, spent, runs = None, 0.0, 0
    while runs < max_runs and spent < budget_s:
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        spent += dt
        runs += 1
        best = dt if best is None else min(best, dt)
    return best, runs


def cpu_baseline_taf(dat_h, n, H, W, K, n_win, win_us, all_cores=True, budget_s=12.0):
    """The CPU oracle (a C port of generate_taf.py:19-76 + harness) on the same stream: one thread, and -- the path
    shards by sequence -- one independent copy of the stream per host thread on all cores (ctypes releases the GIL)."""
    from oracle import oracle as orc
    orc.build()
    st0 = np.full((H, W, 2, K), -6000, np.float32)

    def one():
        view, _ = orc.taf_stream_dat8(dat_h, (H, W), (H, W), K, 0, win_us, n_win, st0)
        orc.quantize_u8(orc.leaky_transform(view))

    best, runs = _best_of(one, budget_s)
    out = {"value": round(n / best / 1e6, 3), "unit": "Mevents/s", "cores": 1, "kind": "port",
           "sample": f"the full workload ({n} events, {n_win} windows), best of {runs} runs, {best:.3f} s each",
           "host_cpus": os.cpu_count(), "host_physical_cores": physical_cores()[0], "threads_per_core": physical_cores()[1],
           "cpu": cpu_model()}
    if all_cores:
        threads = physical_cores()[0]  # one thread per physical core of this host
        with ThreadPoolExecutor(threads) as pool:
            t0 = time.perf_counter()
            list(pool.map(lambda _i: one(), range(threads)))
            dt = time.perf_counter() - t0
        out["all_cores"] = {"value": round(threads * n / dt / 1e6, 3), "unit": "Mevents/s", "cores": threads,
                            "sample": f"{threads} threads, each the full workload on its own copy of the state ({dt:.3f} s)"}
        out["all_cores_value"] = out["all_cores"]["value"]  # (scalar copies: the driver's parser drops nested objects)
        out["all_cores_threads"] = threads
    return out


def cpu_baseline_ev(rec, H, W, budget_s=6.0):
    """The CPU oracle's Event Volume (a C port of generate_eventvolume.py:15-42 + harness) on the same stream: one thread, and
    one independent copy per host thread on all cores (n = 1 and n = all cores)."""
    from oracle import oracle as orc
    orc.build()

    def one():
        orc.ev_stream_dat8(rec, (H, W), (H, W), 5, 250_000, 250_000)

    best, runs = _best_of(one, budget_s, 5)
    out = {"value": round(len(rec) / best / 1e6, 3), "unit": "Mevents/s", "cores": 1, "kind": "port",
           "sample": f"the full workload ({len(rec)} events), best of {runs} runs, {best:.4f} s each", "cpu": cpu_model()}
    threads = physical_cores()[0]  # one thread per physical core of this host
    reps = 8  # ~10 ms per call: several calls per thread so that the pool's start-up does not dominate
    with ThreadPoolExecutor(threads) as pool:
        t0 = time.perf_counter()
        list(pool.map(lambda _i: [one() for _ in range(reps)], range(threads)))
        dt = time.perf_counter() - t0
    out["all_cores"] = {"value": round(threads * reps * len(rec) / dt / 1e6, 3), "unit": "Mevents/s", "cores": threads,
                        "sample": f"{threads} threads x {reps} encodes of the full workload each ({dt:.3f} s)"}
    return out


def cpu_baseline_generic(call, n_events, what, reps=8, budget_s=4.0):
    """One of the smaller oracle encoders (oracle/frlw_oracle.c) on the same stream: one thread (best of a few runs), and one
    independent copy per physical core (`reps` encodes per thread so that the pool's start-up does not dominate)."""
    from oracle import oracle as orc
    orc.build()
    best, runs = _best_of(lambda: call(orc), budget_s, 5)
    out = {"value": round(n_events / best / 1e6, 3), "unit": "Mevents/s", "cores": 1, "kind": "port",
           "sample": f"the full workload ({n_events} events) through oracle.{what}, best of {runs} runs, {best:.4f} s each",
           "cpu": cpu_model()}
    threads = physical_cores()[0]
    with ThreadPoolExecutor(threads) as pool:
        t0 = time.perf_counter()
        list(pool.map(lambda _i: [call(orc) for _ in range(reps)], range(threads)))
        dt = time.perf_counter() - t0
    out["all_cores"] = {"value": round(threads * reps * n_events / dt / 1e6, 3), "unit": "Mevents/s", "cores": threads,
                        "sample": f"{threads} threads x {reps} encodes of the full workload each ({dt:.3f} s)"}
    out["all_cores_value"] = out["all_cores"]["value"]
    out["all_cores_threads"] = threads
    return out


if __name__ == "__main__":
    main()
