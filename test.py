#!/usr/bin/env python3
"""Entry point with the reference's flags (test.py:9-22) for the ``yolox`` experiment on SYNTHETIC streams:
TAF encode on the GPU -> YOLOX eval forward + decode + NMS on the gfx950 engine, batches sharded over GPUs."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    p = argparse.ArgumentParser(description="Test network (synthetic harness).")
    p.add_argument("--local_rank", "--local-rank", type=int, default=None)
    p.add_argument("--exp_type", default="yolox")
    p.add_argument("--dataset", default="gen1")
    p.add_argument("--event_volume_bins", type=float, default=8)
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--nodes", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    args = p.parse_args()
    import torch
    from frlw_evd_amd import dist as fd
    from frlw_evd_amd import e2e

    rank, world, local_rank = fd.init_from_env("nccl", args.local_rank)
    torch.cuda.set_device(local_rank)
    net = e2e.build_model(int(2 * args.event_volume_bins), 2).eval()
    B = int(args.batch_size / args.nodes)
    src = e2e.SyntheticTafSource(B, seed=1005 + 1000 * rank)
    idx = list(range(B))
    with torch.no_grad():
        for _ in range(3):  # plan build, allocator warm-up
            net(src.encode_batch(idx))
        fd.barrier_sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            dets = net(src.encode_batch(idx))  # core/model.py:58 eval branch -> list of (n, 6)
        fd.barrier_sync()
    dt, = fd.max_over_ranks([time.perf_counter() - t0])
    if rank == 0:
        print(json.dumps({"metric": "E2E TAF encode + YOLOX eval (decode + NMS)", "value": round(world * B * args.steps / dt, 1),
                          "unit": "frames/s", "n_gpus": world, "detections_first_image": int(dets[0].shape[0])}))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
