import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frlw_evd_amd import e2e
src = e2e.SyntheticTafSource(32)
for _ in range(3):
    x = src.encode_batch(list(range(32)))
torch.cuda.synchronize()
