"""Device-side time of the small single-stream encodes: one call captured into a HIP graph and replayed (no host launch path
in the timed region) next to the eager per-call wall time.   gpurun -- 'python tools/time_small_graph.py'"""
import sys
import numpy as np
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frlw_evd_amd import synth, event_representation as er, _lib

H, W = 240, 304
LAM = [0.00001, 0.0000025, 0.000001]


def dev(ev):
    return torch.from_numpy(synth.to_dat8(ev).view(np.uint8).reshape(-1, 8).copy()).cuda()


def timed(fn, reps=50):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
        side.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        side.synchronize()
        eager = e0.elapsed_time(e1) / reps * 1e3
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            fn()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return eager, e0.elapsed_time(e1) / reps * 1e3


d_eci = dev(synth.synth_events(1001, 100_000, W, H, 50_000))
d_ev = dev(synth.synth_events(1002, 1_000_000, W, H, 250_000))
d_taf = dev(synth.synth_events(1005, 1_000_000, W, H, 80_000))
d_sae = dev(synth.synth_events(1006, 1_000_000, W, H, 5_000_000, t_offset=30_000_000))
st = torch.full((H, W, 2, 8), -6000.0, device="cuda")
rows = [("eci 100k", lambda: er.encode_eci_dat(d_eci, (H, W), check=False)),
        ("ev 1M fast", lambda: er.encode_ev_dat(d_ev, (H, W), 250_000, 250_000, 5, check=False, fast=True)),
        ("ev 1M general", lambda: er.encode_ev_dat(d_ev, (H, W), 250_000, 250_000, 5, check=False, fast=False)),
        ("taf 1M fast", lambda: er.encode_taf_dat(d_taf, (H, W), st, 0, 10_000, 8, 8, check=False, fast=True)),
        ("sae 1M", lambda: er.encode_sae_dat(d_sae, (H, W), LAM, None, 35_000_000, 5_541_263, check=False))]
for name, fn in rows:
    eager, graph = timed(fn)
    print(f"{name:16s} eager {eager:7.1f} us / call   graph replay {graph:7.1f} us / call")
