"""See tools/train_gaps.sh: `run` = a few Trainer.train_step calls at batch B; `report <kernel_trace.csv>` = idle analysis."""
import csv
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import numpy as np
    import torch
    from frlw_evd_amd import e2e
    from frlw_evd_amd.trainer import Trainer
    B = int(os.environ.get("B", "64"))
    sync = os.environ.get("SYNC", "1") == "1"
    graph = os.environ.get("GRAPH", "0") == "1"
    m = e2e.build_model(in_channels=16, num_classes=2)
    tr = Trainer(m, global_batch=B, nodes=1, iters_per_epoch=100, graph=graph)
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.integers(0, 256, size=(B, 16, 256, 320, 1, 1), dtype=np.uint8)).float().div(255).cuda()
    lab = torch.zeros(B, 80, 5, dtype=torch.float64)
    lab[:, 0] = torch.tensor([0, 100, 90, 60, 40.])
    lab[:, 1] = torch.tensor([1, 220, 150, 50, 80.])
    lab = lab.cuda()
    import time
    for it in range(8):
        if it == 3:
            torch.cuda.synchronize()
            t0 = time.time()
        if sync:
            tr.train_step(x, lab, it)
        else:
            tr.train_step(x, lab, it, sync=False)
    torch.cuda.synchronize()
    print(f"B={B} sync={sync} graph={graph}: {(time.time() - t0) / 5 * 1e3:.2f} ms per step (host clock, 5 steps)")


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n)
    return re.sub(r"[<(].*", "", n)[:40]


def report(path):
    rows = list(csv.DictReader(open(path)))
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda k: k[0])
    # steps are delimited by k_focus (first kernel of every forward)
    starts = [i for i, k in enumerate(ks) if "k_focus" in k[2]]
    print(f"{len(ks)} dispatches, {len(starts)} steps")
    for si in range(max(0, len(starts) - 4), len(starts) - 1):
        seg = ks[starts[si]:starts[si + 1]]
        t0, t1 = seg[0][0], ks[starts[si + 1]][0]
        busy, cur_end, gaps = 0, seg[0][0], []
        for j, (s, e, n) in enumerate(seg):
            if s > cur_end:
                gaps.append((s - cur_end, short(seg[j - 1][2]) if j else "-", short(n), (cur_end - t0) / 1e6))
                busy += e - s
                cur_end = e
            else:
                busy += max(0, e - max(s, cur_end))
                cur_end = max(cur_end, e)
        gaps.append((t1 - cur_end, short(seg[-1][2]), "next step", (cur_end - t0) / 1e6))
        idle = (t1 - t0) - busy
        print(f"step {si}: {(t1 - t0) / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {idle / 1e6:.2f} ms, {len(seg)} dispatches, "
              f"gaps > 20 us: {sum(g[0] for g in gaps if g[0] > 20000) / 1e6:.2f} ms in {sum(1 for g in gaps if g[0] > 20000)}, "
              f"gaps <= 20 us: {sum(g[0] for g in gaps if g[0] <= 20000) / 1e6:.2f} ms in {sum(1 for g in gaps if g[0] <= 20000)}")
        for g in sorted(gaps, key=lambda g: -g[0])[:12]:
            print(f"    {g[0] / 1e3:8.1f} us at +{g[3]:6.2f} ms  after {g[1]:40s} before {g[2]}")


def sequence(path, out):
    """The dispatch sequence of the last complete step: start (us from the step's first kernel), duration, grid, name."""
    rows = list(csv.DictReader(open(path)))
    ks = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(ks) if "k_focus" in r["Kernel_Name"]]
    seg = ks[starts[-2]:starts[-1]]
    t0 = int(seg[0]["Start_Timestamp"])
    with open(out, "w") as f:
        for r in seg:
            n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Kernel_Name"])[:110]
            f.write(f"{(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} "
                    f"g={r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} {n}\n")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    elif sys.argv[1] == "sequence":
        sequence(sys.argv[2], sys.argv[3])
    else:
        report(sys.argv[2])
