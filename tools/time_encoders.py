"""Ad-hoc timing experiments for the TAF encode (run on the GPU box)."""
import os, sys, time
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
    return e0.elapsed_time(e1) / n * 1e3  # us

H, W, K = 720, 1280, 8
ev = synth.synth_events(1003, 10_000_000, W, H, 80_000)
dat = torch.from_numpy(synth.to_dat8(ev).view(np.uint8).reshape(-1, 8).copy()).cuda()
st = torch.full((H, W, 2, K), -6000.0, device="cuda")
variants = {
    "full (u8)": dict(want_u8=True),
    "no outputs": dict(want_u8=False),
    "view f32 only": dict(want_u8=False, want_view=True),
}
for name, kw in variants.items():
    print(f"{name:28s} {timeit(lambda: er.encode_taf_dat(dat, (H, W), st, 0, 10_000, 8, K, check=False, **kw)):8.1f} us")
print(f"{'1 window of 80 ms':28s} {timeit(lambda: er.encode_taf_dat(dat, (H, W), st, 0, 80_000, 1, K, check=False, want_u8=False)):8.1f} us")
print(f"{'64 windows of 1.25 ms':28s} {timeit(lambda: er.encode_taf_dat(dat, (H, W), st, 0, 1_250, 64, K, check=False, want_u8=False)):8.1f} us")
# other encoders on the same stream
print(f"{'ECI 10M':28s} {timeit(lambda: er.encode_eci_dat(dat, (H, W), check=False)):8.1f} us")
print(f"{'SAE 10M':28s} {timeit(lambda: er.encode_sae_dat(dat, (H, W), [1e-5,2.5e-6,1e-6], None, 80_000, 0, check=False)):8.1f} us")
print(f"{'EV 10M':28s} {timeit(lambda: er.encode_ev_dat(dat, (H, W), 80_000, 80_000, check=False)):8.1f} us")
# GEN1-shaped
H, W = 240, 304
ev = synth.synth_events(1005, 1_000_000, W, H, 80_000)
dat1 = torch.from_numpy(synth.to_dat8(ev).view(np.uint8).reshape(-1, 8).copy()).cuda()
st1 = torch.full((H, W, 2, K), -6000.0, device="cuda")
print(f"{'GEN1 TAF 1M':28s} {timeit(lambda: er.encode_taf_dat(dat1, (H, W), st1, 0, 10_000, 8, K, check=False)):8.1f} us")
print(f"{'GEN1 EV 1M':28s} {timeit(lambda: er.encode_ev_dat(dat1, (H, W), 80_000, 80_000, check=False)):8.1f} us")
# plain copy bandwidth reference
a = torch.empty(256 * 1024 * 1024 // 4, device="cuda"); b = torch.empty_like(a)
t = timeit(lambda: b.copy_(a)); print(f"copy 256MiB: {t:.1f} us -> {2*a.numel()*4/t/1e6:.2f} TB/s (r+w)")
