"""EV encode of 32 stacked GEN1 streams (4 across x 8 down): the workload of the ev_gen1 x64 bench row, for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from frlw_evd_amd import event_representation as er, synth
H2, W2, G, across = 240, 304, 32, 4
parts = []
for j in range(G):
    e = dict(synth.synth_events(1052 + j, 1_000_000, W2, H2, 250_000))
    e["x"] = e["x"] + (j % across) * W2
    e["y"] = e["y"] + (j // across) * H2
    parts.append(synth.to_dat8(e))
dat = torch.from_numpy(np.concatenate(parts).view(np.uint8).reshape(-1, 8)).cuda()
for _ in range(8):
    er.encode_ev_dat(dat, ((G // across) * H2, across * W2), 250_000, 250_000, volume_bins=5, check=False)
torch.cuda.synchronize()
