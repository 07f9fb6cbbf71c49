"""decode + NMS launches alone (k_decode_sort + k_nms_matrix + k_nms_sweep): time vs number of candidates (objectness threshold
sweep), GEN1 shape at batch 32 and the 1 Mpx shape at batch 8."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from frlw_evd_amd.yolox import build_yolox
from frlw_evd_amd.yolox.model import recipe_state_dict
from frlw_evd_amd.detector import DetectorEngine
for tag, B, H, W, nc, radius in (("gen1", 32, 256, 320, 2, 5.0), ("1mpx", 8, 512, 640, 7, 2.5)):
    m = build_yolox(10, nc, radius=radius); m.load_state_dict(recipe_state_dict(m, seed=1004)); m.eval()
    rng = np.random.default_rng(4)
    x = torch.from_numpy(rng.integers(0, 256, size=(B, 10, H, W)).astype(np.float32) / np.float32(255)).cuda()
    for thr in (0.3, 0.5, 0.7, 0.9, 0.97):
        m.head.obj_threshold = thr
        eng = DetectorEngine(m)  # fresh plan with this threshold
        raw = eng.raw_outputs(x)
        cand = (raw[:, :, 4] > thr).sum(1).float()
        def run(): eng._run(x, eng.n_forward_ops, -1)
        for _ in range(3): run()
        torch.cuda.synchronize()
        regions = []  # three regions of 20 batches: the MEDIAN is reported, all three are listed (a fresh box now and then spends
        for _r in range(3):  # tens of ms of one region on a clock transition: DESIGN.md 4)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            regions.append(e0.elapsed_time(e1) / 20 * 1e3)
        kept = eng.detect(x)
        print(f"{tag} B={B} obj > {thr}: candidates mean {cand.mean():.0f} max {cand.max():.0f}, kept mean {np.mean([len(k) for k in kept]):.0f}: "
              f"{sorted(regions)[1]:.0f} us per batch (regions {[round(r) for r in regions]})", flush=True)
