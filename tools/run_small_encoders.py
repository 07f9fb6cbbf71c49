#!/usr/bin/env python3
"""The SAE and Event Count Image rows of bench.py as a stand-alone loop (what tools/refresh_r04.sh profiles for their
kernel stats and PMC traffic):   python tools/run_small_encoders.py sae|eci [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from frlw_evd_amd import event_representation as er, synth

which, reps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5
H, W = 240, 304
if which == "sae":
    rec = synth.to_dat8(synth.synth_events(1006, 1_000_000, W, H, 5_000_000, t_offset=30_000_000))
    dat = torch.from_numpy(rec.view(np.uint8).reshape(-1, 8).copy()).cuda()
    lam = [0.00001, 0.0000025, 0.000001]
    mem = er.encode_sae_dat(dat, (H, W), lam, None, 35_000_000, 5_541_263)[2]
    for _ in range(reps):
        mem = er.encode_sae_dat(dat, (H, W), lam, mem, 35_000_000, 5_541_263, check=False)[2]
else:
    rec = synth.to_dat8(synth.synth_events(1001, 100_000, W, H, 50_000))
    dat = torch.from_numpy(rec.view(np.uint8).reshape(-1, 8).copy()).cuda()
    er.encode_eci_dat(dat, (H, W))
    for _ in range(reps):
        er.encode_eci_dat(dat, (H, W), check=False)
torch.cuda.synchronize()
