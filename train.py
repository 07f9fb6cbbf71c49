#!/usr/bin/env python3
"""Entry point with the reference's flags (train.py:9-23) for the ``yolox`` experiment on SYNTHETIC streams:
TAF encode on the GPU -> YOLOX train step under DistributedDataParallel (RCCL).

    python -m torch.distributed.run --nproc-per-node G train.py --exp_type yolox --nodes G --batch_size 64

Datasets, checkpoints, tensorboard and the evaluator of the reference are out of scope;
this harness exists to run BASELINE.json's config 5 and to keep the launch contract (``--local_rank`` /
``--local-rank`` / LOCAL_RANK, ``init_process_group('nccl', 'env://')``, ``--nodes`` = GPU count dividing the
global batch, settings.py:41).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    p = argparse.ArgumentParser(description="Train network (synthetic harness).")
    p.add_argument("--local_rank", "--local-rank", type=int, default=None)
    p.add_argument("--exp_name", default="synthetic")
    p.add_argument("--exp_type", default="yolox")
    p.add_argument("--dataset", default="gen1")
    p.add_argument("--event_volume_bins", type=float, default=8)
    p.add_argument("--batch_size", type=int, default=64)  # GLOBAL batch, divided by --nodes (settings.py:41)
    p.add_argument("--nodes", type=int, default=1)        # number of GPUs
    p.add_argument("--steps", type=int, default=5)
    args = p.parse_args()
    if args.exp_type != "yolox" or args.dataset != "gen1":
        raise SystemExit("only --exp_type yolox on gen1-shaped synthetic streams is on the hot path")

    import torch
    from frlw_evd_amd import dist as fd
    from frlw_evd_amd import e2e
    from frlw_evd_amd.trainer import Trainer

    torch.manual_seed(0)
    rank, world, local_rank = fd.init_from_env("nccl", args.local_rank)
    torch.cuda.set_device(local_rank)
    assert world == args.nodes, "--nodes must equal the number of launched processes (settings.py:41)"
    net = e2e.build_model(int(2 * args.event_volume_bins), 2)
    tr = Trainer(net, global_batch=args.batch_size, nodes=args.nodes, iters_per_epoch=100, local_rank=local_rank,
                 ddp=world > 1)
    B = tr.per_gpu_batch
    src = e2e.SyntheticTafSource(B, seed=1005 + 1000 * rank)
    labels = src.labels(B)
    idx = list(range(B))
    for _ in range(2):  # warm-up (allocator, MIOpen find)
        tr.train_step(src.encode_batch(idx), labels, 0)
    fd.barrier_sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss, lr = tr.train_step(src.encode_batch(idx), labels, i + 1)
    fd.barrier_sync()
    dt, = fd.max_over_ranks([time.perf_counter() - t0])
    if rank == 0:
        print(json.dumps({"metric": "E2E TAF encode + YOLOX train step", "value": round(world * B * args.steps / dt, 1),
                          "unit": "frames/s", "n_gpus": world, "global_batch": B * world, "loss": loss, "lr": lr,
                          "backward": "csrc/train_ops.hip (fp32 MFMA dgrad / wgrad, BatchNorm + SiLU backward), SimOTA csrc/simota.hip"}))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
