import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from frlw_evd_amd import event_representation as er, synth
H2, W2 = 240, 304
def build(G, across):
    parts = []
    for j in range(G):
        e = dict(synth.synth_events(1052 + j, 1_000_000, W2, H2, 250_000))
        e["x"] = e["x"] + (j % across) * W2
        e["y"] = e["y"] + (j // across) * H2
        parts.append(synth.to_dat8(e))
    return torch.from_numpy(np.concatenate(parts).view(np.uint8).reshape(-1, 8)).cuda(), ((G // across) * H2, across * W2)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ref = None
for G, across in [(32, 1), (32, 4), (16, 4), (16, 2), (8, 4), (8, 2), (4, 4), (4, 2)]:
    try:
        dat, shape = build(G, across)
        t = timeit(lambda: er.encode_ev_dat(dat, shape, 250_000, 250_000, volume_bins=5, check=False))
        out, _ = er.encode_ev_dat(dat, shape, 250_000, 250_000, volume_bins=5, check=True)
        # sample 5 must be identical in every layout
        j = 5; r0, c0 = (j // across) * H2, (j % across) * W2
        tile = out[:, r0:r0 + H2, c0:c0 + W2].contiguous()
        if ref is None: ref = tile
        print(f"G={G} across={across} shape={shape}: {t*1e3:8.1f} us  {G*1e6/t/1e6:7.2f} Gev/s  same={bool(torch.equal(tile, ref))}")
        del dat, out
    except Exception as ex:
        print(G, across, "failed:", type(ex).__name__, str(ex)[:100])
