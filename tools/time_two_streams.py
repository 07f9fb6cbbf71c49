"""Do two independent 10 M-event TAF encodes overlap when they run on two HIP streams?  (aggregate throughput of two streams
against one stream doing the same number of encodes)   gpurun -- 'python tools/time_two_streams.py'"""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frlw_evd_amd import synth, event_representation as er

H, W, K = 720, 1280, 8
n = 10_000_000
dats = [torch.from_numpy(synth.to_dat8(synth.synth_events(1003 + j, n, W, H, 80_000)).view(np.uint8).reshape(-1, 8).copy()).cuda() for j in range(2)]
states = [torch.full((1, H, W, 2, K), -6000.0, device="cuda") for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]


def run(k_streams, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for j in range(k_streams):
            with torch.cuda.stream(streams[j]):
                er.encode_taf_batch(dats[j], [0, n], (H, W), states[j], 0, 10_000, 8, K, check=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * k_streams) * 1e6


for j in range(2):
    with torch.cuda.stream(streams[j]):
        er.encode_taf_batch(dats[j], [0, n], (H, W), states[j], 0, 10_000, 8, K, check=True)
run(1, 5); run(2, 5)
print(f"one stream : {run(1, 40):7.1f} us per encode")
print(f"two streams: {run(2, 20):7.1f} us per encode (aggregate)")
