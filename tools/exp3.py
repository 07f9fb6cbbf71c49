import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from frlw_evd_amd import synth, event_representation as er
H, W, K = 720, 1280, 8
ev = synth.synth_events(1003, 10_000_000, W, H, 80_000)
dat = torch.from_numpy(synth.to_dat8(ev).view(np.uint8).reshape(-1, 8).copy()).cuda()
st = torch.full((H, W, 2, K), -6000.0, device="cuda")
for _ in range(3):
    er.encode_taf_dat(dat, (H, W), st, 0, 10_000, 8, K, check=False)
    er.encode_eci_dat(dat, (H, W), check=False)
    er.encode_sae_dat(dat, (H, W), [1e-5], None, 80_000, 0, check=False)
    er.encode_ev_dat(dat, (H, W), 80_000, 80_000, check=False)
torch.cuda.synchronize()
