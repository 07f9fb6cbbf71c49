"""Randomised differential test of the HIP encoders against the CPU oracle (test infrastructure; run on the GPU box).

    python tools/fuzz_encoders.py [n_cases] [seed]

Random frame shapes, event counts, skew (uniform / blob / single hot pixel / hot row), window counts and widths, FIFO
depths, sorted and shuffled streams, events outside the window span, forced hot-tile thresholds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from frlw_evd_amd import _lib, synth, event_representation as er
from oracle import oracle as orc

def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
def dat_dev(ev): return torch.from_numpy(synth.to_dat8(ev).view(np.uint8).reshape(-1, 8).copy()).cuda()

def make_events(rng, n, W, H, span, mode):
    ev = synth.synth_events(int(rng.integers(1 << 30)), n, W, H, span, hotspot=(mode == "blob"))
    if mode == "pixel":  # 40 % of the events on one pixel
        sel = rng.random(n) < 0.4
        ev["x"][sel] = int(rng.integers(W)); ev["y"][sel] = int(rng.integers(H))
    elif mode == "row":  # half of the events on one row
        sel = rng.random(n) < 0.5
        ev["y"][sel] = int(rng.integers(H))
    elif mode == "ties":  # many equal timestamps
        ev["t"] = np.sort((ev["t"] // 500) * 500)
    return ev

def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    large = len(sys.argv) > 3 and sys.argv[3] == "large"  # multi-slice tiles, listed hot tiles, up to 4 M events
    bad = 0
    LAMDAS = [1e-5, 2.5e-6, 1e-6]
    for case in range(n_cases):
        H = int(rng.integers(1, 300)); W = int(rng.choice([int(rng.integers(1, 700)), int(rng.integers(513, 1400))]))
        n = int(rng.choice([0, int(rng.integers(1, 2000)), int(rng.integers(2000, 400_000))]))
        if large:
            H = int(rng.integers(100, 730)); W = int(rng.integers(200, 1300)); n = int(rng.integers(500_000, 4_000_000))
        K = int(rng.integers(1, 9)); nw = int(rng.choice([1, 2, 8, 13, 64])); wus = int(rng.choice([1, 777, 1250, 10_000, 50_000]))
        mode = str(rng.choice(["uniform", "blob", "pixel", "row", "ties"]))
        shuffle = rng.random() < 0.25
        thr = rng.choice([None, 0, 50, 1000])
        span = nw * wus + (wus // 2 if rng.random() < 0.3 else 0)
        t_off = int(rng.integers(0, 3)) * 5000
        desc = f"case {case}: {W}x{H} n={n} K={K} nw={nw} w={wus} {mode} shuffle={shuffle} thr={thr} toff={t_off}"
        er.TUNING = None if thr is None else _lib.FrlwTuning(hot_tile_records=int(thr))
        ev = make_events(rng, n, W, H, max(span, 1), mode)
        ev["t"] = ev["t"] + t_off
        if shuffle and n:
            perm = rng.permutation(n); ev = {k: v[perm] for k, v in ev.items()}
        dat = synth.to_dat8(ev)
        st0 = rng.uniform(-300, 0, size=(H, W, 2, K)).astype(np.float32)
        try:
            oview, ost = orc.taf_stream_dat8(dat, (H, W), (H, W), K, t_off, wus, nw, st0)
            st = dev(st0)
            u8, view = er.encode_taf_dat(dat_dev(ev), (H, W), st, t_off, wus, nw, K, want_view=True)
            ok = np.array_equal(st.cpu().numpy(), ost) and np.array_equal(view.cpu().numpy(), oview)
            bins = int(rng.integers(1, 9))
            f32 = er.encode_ev_dat(dat_dev(ev), (H, W), t_off + span, max(span, 1), volume_bins=bins)[0]
            ok &= np.array_equal(f32.cpu().numpy(), orc.ev_stream_dat8(dat, (H, W), (H, W), bins, t_off + span, max(span, 1)))
            eci = er.encode_eci_dat(dat_dev(ev), (H, W))[0]
            ok &= np.array_equal(eci.cpu().numpy(), orc.eci_stream_dat8(dat, (H, W), (H, W)))
            # SAE: memory bit-exact (last writer), outputs within 2 ulp (expf)
            mem0 = rng.uniform(-1e6, 1e5, size=(2, H, W)).astype(np.float32) if rng.random() < 0.5 else None
            now = t_off + span
            o, _, mem = er.encode_sae_dat(dat_dev(ev), (H, W), LAMDAS, dev(mem0) if mem0 is not None else None, now, max(span, 1))
            oo, omem = orc.sae_stream_dat8(dat, (H, W), (H, W), LAMDAS, mem0, now, max(span, 1))
            ok &= np.array_equal(mem.cpu().numpy(), omem)
            ok &= bool(np.all(np.abs(o.cpu().numpy().view(np.int32).astype(np.int64) - oo.view(np.int32).astype(np.int64)) <= 2))
            if not large and n and rng.random() < 0.5:  # the reference-named float64 (N, 4) entry points
                tn = (ev["t"] - ev["t"].min()) / max(float(ev["t"].max() - ev["t"].min()), 1.0)
                e = synth.to_xytp_f64(ev, tn)
                vv, ss, _ = er.generate_taf_cuda(dev(e), (H, W), dev(st0), K)
                ov, os_ = orc.taf_window(e, (H, W), st0, K)
                ok &= np.array_equal(ss.cpu().numpy(), os_) and np.array_equal(vv.cpu().numpy(), ov)
                ok &= np.array_equal(er.generate_agile_event_volume_cuda(dev(e), (H, W), 0, bins)[0].cpu().numpy(),
                                     orc.event_volume(e, (H, W), bins))
                ok &= np.array_equal(er.generate_eventframe(dev(e), (H, W))[0].cpu().numpy(), orc.eventframe(e, (H, W)))
        except Exception as e:  # noqa: BLE001
            ok = False
            desc += f" EXC {type(e).__name__}: {e}"
        if not ok:
            bad += 1
            print("MISMATCH", desc)
    er.TUNING = None
    print(f"{n_cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
