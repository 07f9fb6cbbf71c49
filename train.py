#!/usr/bin/env python3
"""Training entry point with the reference's command line (train.py:9-71): same flags, same
``init_process_group('nccl', 'env://')`` (RCCL on ROCm), same ``Setting_train_val`` -> ``yolox(settings).train()`` /
``yoloxtafBFM(settings).train()`` dispatch, one process per GPU, ``--nodes`` = number of GPUs dividing the global batch.

    python -m torch.distributed.run --nproc-per-node G train.py --dataset gen1 --batch_size 64 --exp_name E \\
        --exp_type yolox --event_volume_bins 8 --nodes G

Without ``--data_path`` / ``--bbox_path`` the run is SYNTHETIC: event streams are generated and TAF-encoded on the GPU
every step (BASELINE.json config 5); with them it stops with a note -- reading the pre-encoded dataset files is outside
this build's hot path.  ``--local_rank`` / ``--local-rank`` / ``LOCAL_RANK`` are all accepted
(torch.distributed.launch vs torchrun).  ``FRLW_MAX_EPOCHS`` / ``FRLW_SYNTHETIC_BATCHES`` shorten synthetic runs.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def build_parser():
    parser = argparse.ArgumentParser(description="Train network.")
    parser.add_argument("--local_rank", "--local-rank", type=int, default=None,
                        help="local rank for DistributedDataParallel")  # no need to set
    parser.add_argument("--resume_exp")  # name of the experiment to resume
    parser.add_argument("--exp_name")    # name of a new experiment (an existing one of that name is overwritten)
    parser.add_argument("--exp_type", default="basic")
    parser.add_argument("--log_path", default="log/")  # logs and checkpoints
    parser.add_argument("--dataset", default="gen1")   # gen1 / gen4
    parser.add_argument("--bbox_path")  # annotations ("train, val, test" level directory)
    parser.add_argument("--data_path")  # pre-encoded data
    parser.add_argument("--event_volume_bins", type=float, default=5)  # x 2 (polarity) = input channels
    parser.add_argument("--batch_size", type=int, default=30)  # GLOBAL batch
    parser.add_argument("--num_cpu_workers", type=int, default=-1)
    parser.add_argument("--nodes", type=int, default=1)  # number of GPUs
    parser.add_argument("--augmentation", type=bool, default=True)  # (any non-empty string is True, like the reference)
    return parser


def init_distributed(args):
    """train.py:28-31: seed, device, process group over env:// -- a plain ``python train.py`` becomes a 1-rank job."""
    import torch
    if args.local_rank is None:
        args.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    torch.manual_seed(0)
    backend = os.environ.get("FRLW_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        torch.cuda.set_device(args.local_rank)
    if not torch.distributed.is_initialized():
        torch.distributed.init_process_group(backend=backend, init_method="env://")
    return torch.distributed.get_rank(), torch.distributed.get_world_size()


def pick_experiment(exp_type):
    from frlw_evd_amd import exp
    if exp_type in exp.EXPERIMENTS:
        return exp.EXPERIMENTS[exp_type]
    if exp_type in exp.OTHER_RECIPES:
        raise SystemExit(f"--exp_type {exp_type}: the AED / YOLOv3 detectors are outside this build's scope "
                         f"(SURVEY.md section 2 #15, #16); available: {', '.join(exp.EXPERIMENTS)}")
    raise SystemExit(f"unknown --exp_type {exp_type}")


def main(argv=None):
    args = build_parser().parse_args(argv)
    cls = pick_experiment(args.exp_type)
    rank, world = init_distributed(args)
    assert world == args.nodes, "--nodes must equal the number of launched processes (settings.py:41)"
    from frlw_evd_amd.settings import Setting_train_val
    settings = Setting_train_val(args)
    trainer = cls(settings)
    trainer.train()
    import torch
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    return trainer


if __name__ == "__main__":
    main()
