#!/usr/bin/env python3
"""Evaluation entry point with the reference's command line (test.py:9-52): same flags, process-group init and
``Setting_test`` -> ``yolox(settings).test()`` dispatch; ``--resume_exp E`` loads ``<log_path>E/checkpoints/best_epoch.pth``
(a checkpoint written by this build or by the reference: same dict layout and parameter names), ``--record True`` writes
``summarise.npz``.  Without ``--data_path`` / ``--bbox_path`` the run is synthetic (see train.py); the detections go
through the gfx950 engine (decode + NMS on device) and the evaluator hand-off."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def build_parser():
    parser = argparse.ArgumentParser(description="Test network.")
    parser.add_argument("--local_rank", "--local-rank", type=int, default=None,
                        help="local rank for DistributedDataParallel")
    parser.add_argument("--resume_exp")  # experiment whose best_epoch checkpoint is evaluated
    parser.add_argument("--exp_type", default="basic")
    parser.add_argument("--log_path", default="log/")
    parser.add_argument("--record", type=bool)  # summarise.npz under log_path for visualisation / motion-level evaluation
    parser.add_argument("--dataset", default="gen1")
    parser.add_argument("--bbox_path")
    parser.add_argument("--data_path")
    parser.add_argument("--event_volume_bins", type=int, default=5)
    parser.add_argument("--batch_size", type=int, default=1)
    parser.add_argument("--num_cpu_workers", type=int, default=-1)
    parser.add_argument("--nodes", type=int, default=1)
    return parser


def main(argv=None):
    import train as train_entry
    args = build_parser().parse_args(argv)
    cls = train_entry.pick_experiment(args.exp_type)
    train_entry.init_distributed(args)
    from frlw_evd_amd.settings import Setting_test
    settings = Setting_test(args)
    tester = cls(settings)
    result = tester.test()
    import torch
    if torch.distributed.get_rank() == 0 and isinstance(result, dict):
        n_dt = sum(len(d) for d in result["dt_boxes_list"])
        print({"images": len(result["gt_boxes_list"]), "detections": n_dt, "avg_infer_ms": round(result["avg_infer_ms"], 3)})
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
