"""Child process of tests/test_rccl_gpu.py: ONE rank on backend ``nccl`` (= RCCL on ROCm), started fresh with
WORLD_SIZE=1 RANK=0 LOCAL_RANK=0.  Runs every collective call of the product on RCCL -- the calls an N-rank job makes,
with a group of one -- and writes what it observed to the JSON file named on the command line:

  * ``dist.init_from_env("nccl", force=True)`` with ``device_id`` (train.py:31 of the reference always initialises);
  * ``barrier_sync``, ``max_over_ranks``, ``sum_over_ranks``, ``job_throughput`` (bench.py's timed-region bracket);
  * ``reduce_scatter_tensor`` / ``all_gather_into_tensor`` directly;
  * a DDP-wrapped ``Trainer.train_step`` (core/exp.py:391) with every communication hook: default, timed, rs_ag --
    losses and parameters after two steps must equal the bare (non-DDP) trainer's bit for bit at world 1;
  * the same step as two HIP graphs around one all-reduce of the flat gradient buffer (``Trainer(ddp=True, graph=True)``);
  * ``encode_taf_stripe`` with its own collective (the MAX-reduce of the window-mask bits);
  * ``destroy_process_group``.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path):
    import numpy as np
    import torch
    import torch.distributed as dist
    from frlw_evd_amd import dist as fd, event_representation as er, synth
    from frlw_evd_amd.trainer import Trainer
    from frlw_evd_amd.yolox import build_yolox
    from frlw_evd_amd.yolox.model import recipe_state_dict

    res = {}
    rank, world, local_rank = fd.init_from_env("nccl", force=True)
    assert (rank, world, local_rank) == (0, 1, 0)
    assert dist.is_initialized() and dist.get_backend() == "nccl"
    torch.cuda.set_device(local_rank)
    fd.barrier_sync()
    res["max"] = fd.max_over_ranks([1.5, 2.5])
    res["sum"] = fd.sum_over_ranks([3.0])
    res["rate"] = fd.job_throughput(10.0, 2.0)

    # the two halves of the rs_ag hook, called directly
    src = torch.arange(1024, dtype=torch.float32, device="cuda")
    shard = torch.empty(1024, dtype=torch.float32, device="cuda")
    dist.reduce_scatter_tensor(shard, src)
    back = torch.empty_like(src)
    dist.all_gather_into_tensor(back, shard)
    torch.cuda.synchronize()
    res["rs_ag_roundtrip"] = bool(torch.equal(back, src))

    # ---- train step under DDP with every hook vs the bare trainer
    def inputs(seed, B=2):
        rng = np.random.default_rng(seed)
        x = torch.from_numpy(rng.integers(0, 256, size=(B, 16, 128, 160, 1, 1)).astype(np.float32) / np.float32(255))
        lab = torch.zeros(B, 80, 5, dtype=torch.float64)
        lab[:, 0] = torch.tensor([0, 60.0 + seed, 50.0, 40.0, 30.0])
        lab[:, 1] = torch.tensor([1, 100.0, 90.0 - seed, 30.0, 50.0])
        return x.cuda(), lab.cuda()

    def trainer(ddp, hook, graph=False):
        m = build_yolox(16, 2)
        m.load_state_dict(recipe_state_dict(m, seed=41))
        return Trainer(m.cuda(), global_batch=2, nodes=1, iters_per_epoch=4, max_epoch=10, warmup_epochs=0,
                       local_rank=0, ddp=ddp, comm_hook=hook, graph=graph)

    batches = [inputs(s) for s in range(2)]

    def run(tr):
        losses = [tr.train_step(x, lab, i)[0] for i, (x, lab) in enumerate(batches)]
        mod = tr.model.module if hasattr(tr.model, "module") else tr.model
        flat = torch.cat([p.detach().flatten() for p in mod.parameters()])
        return losses, flat

    base_losses, base_params = run(trainer(False, None))
    res["losses_bare"] = base_losses
    res["hooks"] = {}
    for hook in ("default", "timed", "rs_ag"):
        tr = trainer(True, hook)
        losses, params = run(tr)
        entry = {"losses": losses, "losses_equal": losses == base_losses,
                 "params_equal": bool(torch.equal(params, base_params))}
        if hook == "timed":
            s = tr.comm_hook.summary()
            entry["buckets"] = s.get("buckets", 0)
            entry["total_ms"] = s.get("total_ms", 0.0)
        res["hooks"][hook] = entry
        del tr

    # the DDP step as two HIP graphs around ONE all-reduce of the flat gradient buffer (Trainer(ddp=True, graph=True)): the
    # collective runs on RCCL between the replays; a third batch of another shape takes the eager form of the same exchange
    tr = trainer(True, None, graph=True)
    losses, params = run(tr)
    odd = inputs(7, B=1)
    bare = trainer(False, None)
    run(bare)
    res["graph2"] = {"losses": losses, "losses_equal": losses == base_losses, "params_equal": bool(torch.equal(params, base_params)),
                     "two_graphs": tr._graph is not None and len(tr._graph) == 5, "wrapped": hasattr(tr.model, "module"),
                     "odd_batch_loss_equal": tr.train_step(*odd, 2)[0] == bare.train_step(*odd, 2)[0]}
    del tr, bare

    # ---- row-stripe encode: the window-mask reduce on RCCL (MAX over the mask bits)
    H, W, K, win, n_win = 240, 304, 8, 10_000, 8
    ev = synth.synth_events(77, 300_000, W, H, n_win * win)
    w_idx = np.minimum(ev["t"] // win, n_win - 1)
    keep = w_idx != 6   # window 6 empty everywhere
    rec = synth.to_dat8({k: v[keep] for k, v in ev.items()})
    dat = torch.from_numpy(np.ascontiguousarray(rec).view(np.uint8).reshape(-1, 8).copy()).cuda()
    st = torch.full((1, H, W, 2, K), -6000.0, device="cuda")
    u8s, _ = er.encode_taf_stripe(dat, [0, len(rec)], (H, W), (0, H), st, 0, win, n_win, K)   # group = WORLD: RCCL
    full = torch.full((1, H, W, 2, K), -6000.0, device="cuda")
    u8f, _ = er.encode_taf_batch(dat, [0, len(rec)], (H, W), full, 0, win, n_win, K)
    res["stripe_equals_whole"] = bool(torch.equal(st, full) and torch.equal(u8s, u8f))
    masks = torch.tensor([0x5A5A00000000F00F - (1 << 64) if 0x5A5A00000000F00F >= (1 << 63) else 0x5A5A00000000F00F,
                          -1, 0, 1 << 62], dtype=torch.int64, device="cuda")
    want = masks.clone()
    er._or_reduce_window_masks(masks, None)
    res["mask_reduce_identity"] = bool(torch.equal(masks, want))

    fd.barrier_sync()
    with open("/proc/self/maps") as f:
        libs = sorted({line.split()[-1] for line in f if "rccl" in line.lower() or "libfrlw_evd" in line})
    res["libs"] = libs
    dist.destroy_process_group()
    res["destroyed"] = not dist.is_initialized()
    with open(out_path, "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    main(sys.argv[1])
