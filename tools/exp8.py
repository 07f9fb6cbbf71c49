import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from frlw_evd_amd import synth, event_representation as er
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
H, W, K = 240, 304, 8
ev = synth.synth_events(1005, 1_000_000, W, H, 80_000)
dat = torch.from_numpy(synth.to_dat8(ev).view(np.uint8).reshape(-1, 8).copy()).cuda()
st = torch.full((H, W, 2, K), -6000.0, device="cuda")
print(os.environ.get("FRLW_DBG", "0"), f"GEN1 TAF: {timeit(lambda: er.encode_taf_dat(dat, (H, W), st, 0, 10_000, 8, K, check=False, want_u8=(os.environ.get('U8','1')=='1'))):.1f} us")
