import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from frlw_evd_amd import synth, event_representation as er
H, W, K = 720, 1280, 8
for hot in (False, True):
    ev = synth.synth_events(1003, 10_000_000, W, H, 80_000, hotspot=hot)
    dat = torch.from_numpy(synth.to_dat8(ev).view(np.uint8).reshape(-1, 8).copy()).cuda()
    st = torch.full((H, W, 2, K), -6000.0, device="cuda")
    er.encode_taf_dat(dat, (H, W), st, 0, 10_000, 8, K, check=True)
    ws = list(er._WORKSPACES.values())[0]
    hdr = ws[:16].cpu().numpy().view(np.uint32)
    print("hot" if hot else "uniform", "status", hdr[0], "fallback tiles", hdr[1], "wmask", hex(int(ws[8:16].cpu().numpy().view(np.uint64)[0])))
    t = ev["t"]; print("  sorted t:", bool(np.all(np.diff(t) >= 0)))
