"""TAF encode (5 launches) replayed from a HIP graph: does it shorten the launch-bound GEN1-shaped encode?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from frlw_evd_amd import synth, event_representation as er

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (H, W, n) in ((240, 304, 1_000_000), (720, 1280, 10_000_000)):
    ev = synth.synth_events(1005, n, W, H, 80_000)
    dat = torch.from_numpy(synth.to_dat8(ev).view(np.uint8).reshape(-1, 8).copy()).cuda()
    st = torch.full((H, W, 2, 8), -6000.0, device="cuda")
    run = lambda: er.encode_taf_dat(dat, (H, W), st, 0, 10_000, 8, 8, check=False)
    eager = timeit(run)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = run()
    print(f"{W}x{H} n={n}: eager {eager:.1f} us, graph replay {timeit(g.replay):.1f} us")
