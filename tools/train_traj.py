import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from frlw_evd_amd.trainer import Trainer
from frlw_evd_amd.yolox import build_yolox
from frlw_evd_amd.yolox.model import recipe_state_dict
def inputs(B, seed):
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.integers(0, 256, size=(B, 16, 128, 160, 1, 1)).astype(np.float32) / np.float32(255))
    lab = torch.zeros(B, 80, 5, dtype=torch.float64)
    lab[:, 0] = torch.tensor([0, 60.0 + seed, 50.0, 40.0, 30.0]); lab[:, 1] = torch.tensor([1, 100.0, 90.0 - seed, 30.0, 50.0])
    return x.cuda(), lab.cuda()
m = build_yolox(16, 2); m.load_state_dict(recipe_state_dict(m, seed=31))
tr = Trainer(m.cuda(), global_batch=8, nodes=1, iters_per_epoch=4, max_epoch=10, warmup_epochs=1)
ls = [tr.train_step(*inputs(8, s), s)[0] for s in range(12)]
print(os.environ.get("FRLW_TRAIN_STACK"), os.environ.get("FRLW_TRAIN_FUSE"), " ".join(f"{l:.6f}" for l in ls))
