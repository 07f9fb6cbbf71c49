import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from frlw_evd_amd import synth, event_representation as er
H, W, K = 720, 1280, 8
NS = int(os.environ.get("NS", "2"))
dats, sts, streams = [], [], []
for i in range(NS):
    ev = synth.synth_events(1003 + i, 10_000_000, W, H, 80_000)
    dats.append(torch.from_numpy(synth.to_dat8(ev).view(np.uint8).reshape(-1, 8).copy()).cuda())
    sts.append(torch.full((H, W, 2, K), -6000.0, device="cuda"))
    streams.append(torch.cuda.Stream())
def run(n):
    for it in range(n):
        i = it % NS
        with torch.cuda.stream(streams[i]):
            er.encode_taf_dat(dats[i], (H, W), sts[i], 0, 10_000, 8, K, check=False)
run(2 * NS); torch.cuda.synchronize()
t0 = time.perf_counter(); N = 40; run(N); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print(f"streams={NS}: {dt*1e6:.1f} us per encode -> {10e6/dt/1e9:.2f} Gev/s")
