#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own encoder
functions (imported from /root/reference, torch-CPU, one thread) on seeded synthetic streams.

Runs only in the build container (the reference never travels to the GPU box).  Inputs are not
stored: every consumer regenerates them from the seed with frlw_evd_amd.synth.  The harness
steps between the functions (window selection, f64 normalisation, coordinate down-scale,
nearest resize, uint8 truncation) are restated here with the same torch/numpy calls as the
reference's ``__main__`` blocks; each block cites the lines it follows.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz
"""
import hashlib
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("FRLW_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

# --- stubs for modules the reference imports but this image lacks ---
for name, attrs in {"tkinter": {"S": None}, "h5py": {}}.items():
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:
            m = types.ModuleType(name)
            for k, v in attrs.items():
                setattr(m, k, v)
            sys.modules[name] = m
torch.cuda.synchronize = lambda *a, **k: None  # called unconditionally, generate_eventvolume.py:39
torch.set_num_threads(1)  # sequential stream order = the oracle semantics

import generate_eventcountimage as ref_eci  # noqa: E402
import generate_eventvolume as ref_ev  # noqa: E402
import generate_surfaceofactiveevents as ref_sae  # noqa: E402
import generate_taf as ref_taf  # noqa: E402

from frlw_evd_amd import synth  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def sample_idx(n, k=4096, seed=7):
    return np.sort(np.random.default_rng(seed).choice(n, size=min(k, n), replace=False))


def pack_big(prefix, arr, out):
    """Big f32 buffer -> sha256 + a seeded sample of flat positions."""
    flat = np.ascontiguousarray(arr).reshape(-1)
    idx = sample_idx(flat.size)
    out[prefix + "_sha"] = np.array(sha(flat))
    out[prefix + "_shape"] = np.array(arr.shape)
    out[prefix + "_idx"] = idx
    out[prefix + "_val"] = flat[idx]


def save(name, d):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


# ---------------------------------------------------------------------------------------------
# harness restatements
# ---------------------------------------------------------------------------------------------
def harness_resize(vol, target_shape):
    # generate_eventvolume.py:149
    return torch.nn.functional.interpolate(vol[None, :, :, :], size=target_shape, mode="nearest")[0]


def harness_eci(events, shape, target_shape):
    # generate_eventcountimage.py:155-180
    rh, rw = target_shape[0] / shape[0], target_shape[1] / shape[1]
    events_ = events.clone()
    if target_shape[0] < shape[0]:
        events_[:, 0] = events_[:, 0] * rw
        events_[:, 1] = events_[:, 1] * rh
        volume, _ = ref_eci.generate_eventframe(events_, target_shape)
        native = volume
    else:
        native, _ = ref_eci.generate_eventframe(events_, shape)
        volume = harness_resize(native, target_shape)
    return native.numpy(), volume.numpy().copy().astype(np.uint8)


def harness_ev(events_, shape, target_shape, end_time, time_window, bins=5):
    # generate_eventvolume.py:139-157
    rh, rw = target_shape[0] / shape[0], target_shape[1] / shape[1]
    events = events_[events_[:, 2] > end_time - time_window]
    events[:, 2] = (events[:, 2] - (end_time - time_window)) / time_window
    if target_shape[0] < shape[0]:
        events[:, 0] = events[:, 0] * rw
        events[:, 1] = events[:, 1] * rh
        volume, _ = ref_ev.generate_agile_event_volume_cuda(events, target_shape, time_window, bins)
        native = volume
    else:
        native, _ = ref_ev.generate_agile_event_volume_cuda(events, shape, time_window, bins)
        volume = harness_resize(native, target_shape)
    v = volume.numpy()
    v = np.where(v > 255, 255, v)
    return native.numpy(), v.astype(np.uint8)


def harness_sae(events, shape, target_shape, lamdas, memory, unique_time, tw):
    # generate_surfaceofactiveevents.py:183-194
    rh, rw = target_shape[0] / shape[0], target_shape[1] / shape[1]
    end_time = int(unique_time)
    events_ = events[events[:, 2] > end_time - tw].clone()
    if target_shape[0] < shape[0]:
        events_[:, 0] = events_[:, 0] * rw
        events_[:, 1] = events_[:, 1] * rh
        volume, memory, _ = ref_sae.generate_leaky_cuda(events_, target_shape, lamdas, memory, unique_time)
        native = volume
    else:
        native, memory, _ = ref_sae.generate_leaky_cuda(events_, shape, lamdas, memory, unique_time)
        volume = harness_resize(native, target_shape)
    return native.numpy(), memory, volume.numpy().copy().astype(np.uint8)


def harness_taf(events, shape, target_shape, start_time, end_time, memory, K=8, win=10000):
    # generate_taf.py:195-235
    rh, rw = target_shape[0] / shape[0], target_shape[1] / shape[1]
    z = torch.zeros_like(events[:, 0])
    bins = math.ceil((end_time - start_time) / win)
    for i in range(bins):
        z = torch.where((events[:, 2] >= start_time + i * win) & (events[:, 2] <= start_time + (i + 1) * win),
                        torch.zeros_like(events[:, 2]) + i, z)
    events = torch.cat([events, z[:, None]], dim=1)
    if memory is None:
        if target_shape[0] < shape[0]:
            memory = torch.zeros((target_shape[0], target_shape[1], 2, K)) - 6000
        else:
            memory = torch.zeros((shape[0], shape[1], 2, K)) - 6000
    enc_shape = target_shape if target_shape[0] < shape[0] else shape
    states = []
    native = None
    for it in range(bins):
        events_ = events[events[..., 4] == it]
        t_max = start_time + (it + 1) * win
        t_min = start_time + it * win
        events_[:, 2] = (events_[:, 2] - t_min) / (t_max - t_min + 1e-8)
        if target_shape[0] < shape[0]:
            events_[:, 0] = events_[:, 0] * rw
            events_[:, 1] = events_[:, 1] * rh
            volume, memory, _ = ref_taf.generate_taf_cuda(events_, target_shape, memory, K)
            native = volume
        else:
            volume, memory, _ = ref_taf.generate_taf_cuda(events_, shape, memory, K)
            native = volume
            volume = harness_resize(volume, target_shape)
        states.append(memory.clone())
    volume = volume.view(K, 2, target_shape[0], target_shape[1])
    volume = ref_taf.leaky_transform(volume)
    ecd = volume.numpy().copy()
    ecd = np.flip(ecd, axis=0)
    return dict(native=native.numpy().copy(), state=memory.numpy().copy(), states=states,
                leaky=volume.numpy().copy(), u8=np.ascontiguousarray(ecd).astype(np.uint8),
                enc_shape=enc_shape)


# ---------------------------------------------------------------------------------------------
# tiny hand-checkable streams on (H, W) = (8, 12)
# ---------------------------------------------------------------------------------------------
def tiny_events():
    H, W = 8, 12
    rows = []
    # hot pixel: 25 events on (x=3, y=2, p=1)  -> ECI saturation (>= 20)
    rows += [(3, 2, 0.02 * i, 1) for i in range(25)]
    # exactly 20 and 19 events on two other cells
    rows += [(0, 0, 0.5, 0)] * 20 + [(11, 7, 0.25, 1)] * 19
    # first/last row/col, both polarities
    rows += [(0, 7, 0.1, 1), (11, 0, 0.9, 0), (11, 7, 1.0, 0), (0, 0, 0.0, 1)]
    # t exactly on EV bin centres / edges (k / 5) and just around them
    for k in range(0, 6):
        rows += [(5, 4, k / 5.0, k & 1), (6, 4, np.nextafter(k / 5.0, 1.0), 1), (6, 4, np.nextafter(k / 5.0, 0.0) if k else 0.0, 0)]
    # x >= W is NOT an error in the reference: the flat index x + W*y aliases into the next row
    rows += [(W + 1, 1, 0.3, 1), (W, 0, 0.7, 0)]
    rng = np.random.default_rng(11)
    for _ in range(300):
        rows.append((int(rng.integers(0, W)), int(rng.integers(0, H)), float(rng.random()), int(rng.integers(0, 2))))
    ev = np.array(rows, dtype=np.float64)
    return H, W, ev


def gen_tiny():
    H, W, ev = tiny_events()
    d = {"events": ev, "shape": np.array([H, W])}
    out, _ = ref_eci.generate_eventframe(T(ev), (H, W))
    d["eci"] = out.numpy()
    out, _ = ref_ev.generate_agile_event_volume_cuda(T(ev), (H, W), 50000, 5)
    d["ev"] = out.numpy()
    out3, _ = ref_ev.generate_agile_event_volume_cuda(T(ev), (H, W), 50000, 3)
    d["ev_bins3"] = out3.numpy()
    # SAE: absolute microsecond stamps, 2 successive calls, some out-of-range coordinates
    ev_abs = ev.copy()
    ev_abs[:, 2] = 30_000_000 + np.floor(ev[:, 2] * 4_000_000)
    order = np.argsort(ev_abs[:, 2], kind="stable")
    ev_abs = ev_abs[order]
    ev_oob = np.concatenate([ev_abs, np.array([[W, 1, 33_000_000, 1], [2, H, 33_500_000, 0]], dtype=np.float64)])
    lam = [0.00001, 0.0000025, 0.000001]
    half = len(ev_oob) // 2
    now1 = np.int64(32_000_000)
    o1, m1, _ = ref_sae.generate_leaky_cuda(T(ev_oob[:half]), (H, W), lam, None, now1)
    now2 = np.int64(34_000_000)
    o2, m2, _ = ref_sae.generate_leaky_cuda(T(ev_oob[half:]), (H, W), lam, m1, now2)
    d.update(sae_events=ev_oob, sae_half=np.array(half), sae_now=np.array([now1, now2]),
             sae_out1=o1.numpy(), sae_mem1=m1.numpy(), sae_out2=o2.numpy(), sae_mem2=m2.numpy())
    # TAF: 4 windows on K=8 state: normal, EMPTY (state must be unchanged), single event, normal
    K = 8
    state = torch.zeros((H, W, 2, K)) - 6000
    wins = [ev[:150], ev[:0], ev[150:151], ev[151:]]
    d["taf_splits"] = np.array([0, 150, 150, 151, len(ev)])
    for i, w in enumerate(wins):
        w5 = np.concatenate([w, np.zeros((len(w), 1))], axis=1)
        view, state, _ = ref_taf.generate_taf_cuda(T(w5), (H, W), state, K)
        d[f"taf_view{i}"] = view.numpy().copy()
        d[f"taf_state{i}"] = state.numpy().copy()
    d["taf_leaky"] = ref_taf.leaky_transform(T(d["taf_view3"])).numpy()
    # K = 4 variant
    state = torch.zeros((H, W, 2, 4)) - 6000
    for i, w in enumerate(wins):
        w5 = np.concatenate([w, np.zeros((len(w), 1))], axis=1)
        view, state, _ = ref_taf.generate_taf_cuda(T(w5), (H, W), state, 4)
    d["taf_k4_state"] = state.numpy().copy()
    d["taf_k4_view"] = view.numpy().copy()
    save("tiny", d)


# ---------------------------------------------------------------------------------------------
# GEN1-shaped (240 x 304 sensor, 256 x 320 detector) -- SURVEY.md section 8d cfg 1, 2, 5
# ---------------------------------------------------------------------------------------------
GEN1 = ((240, 304), (256, 320))
MPX = ((720, 1280), (512, 640))


def gen_gen1():
    shape, tshape = GEN1
    d = {}
    # cfg 1: ECI, seed 1001, N = 100000, T = 50000
    ev = synth.synth_events(1001, 100_000, shape[1], shape[0], 50_000)
    native, u8 = harness_eci(T(synth.to_xytp_f64(ev)), shape, tshape)
    d["eci_native"] = native
    d["eci_u8"] = u8
    # hotspot variant
    ev = synth.synth_events(1001, 100_000, shape[1], shape[0], 50_000, hotspot=True)
    native, u8 = harness_eci(T(synth.to_xytp_f64(ev)), shape, tshape)
    d["eci_hot_native_sha"] = np.array(sha(native))
    d["eci_hot_u8"] = u8
    save("gen1_eci", d)

    # cfg 2: EV, seed 1002, N = 1e6, window 250000 us, 5 bins
    d = {}
    for tag, hot in (("", False), ("hot_", True)):
        ev = synth.synth_events(1002, 1_000_000, shape[1], shape[0], 250_000, hotspot=hot)
        native, u8 = harness_ev(T(synth.to_xytp_f64(ev)), shape, tshape, 250_000, 250_000)
        pack_big(tag + "native", native, d)
        d[tag + "u8"] = u8
    save("gen1_ev", d)

    # SAE: seed 1006, N = 1e6 over 5 s starting at 30 s, two label times with memory carry
    d = {}
    lam = [0.00001, 0.0000025, 0.000001]
    ev = synth.synth_events(1006, 1_000_000, shape[1], shape[0], 5_000_000, t_offset=30_000_000)
    e = T(synth.to_xytp_f64(ev))
    now1, now2 = np.int64(33_000_000), np.int64(35_000_000)
    cut = int(np.searchsorted(ev["t"], now1, side="right"))
    n1, m1, u1 = harness_sae(e[:cut], shape, tshape, lam, None, now1, 5541263)
    n2, m2, u2 = harness_sae(e[cut:], shape, tshape, lam, m1, now2, 5541263)
    d.update(now=np.array([now1, now2]), cut=np.array(cut), u8_1=u1, u8_2=u2)
    pack_big("mem1", m1.numpy(), d)
    pack_big("mem2", m2.numpy(), d)
    pack_big("native1", n1, d)
    pack_big("native2", n2, d)
    save("gen1_sae", d)

    # cfg 5 encoder part: TAF K=8, 8 windows x 125000 events, then a second label (state carry)
    d = {}
    for tag, hot in (("", False), ("hot_", True)):
        ev = synth.synth_events(1005, 1_000_000, shape[1], shape[0], 80_000, hotspot=hot)
        r = harness_taf(T(synth.to_xytp_f64(ev)), shape, tshape, 0, 80_000, None)
        pack_big(tag + "state", r["state"], d)
        pack_big(tag + "native", r["native"], d)
        d[tag + "u8"] = r["u8"]
        if not hot:
            ev2 = synth.synth_events(2005, 300_000, shape[1], shape[0], 30_000, t_offset=80_000)
            r2 = harness_taf(T(synth.to_xytp_f64(ev2)), shape, tshape, 80_000, 110_000, T(r["state"]))
            pack_big("carry_state", r2["state"], d)
            d["carry_u8"] = r2["u8"]
    save("gen1_taf", d)


def gen_mpx():
    shape, tshape = MPX
    # cfg 3: TAF K=8, seed 1003, N = 1e7, 1280x720 NATIVE (no down-scale), 8 windows x 10000 us
    d = {}
    ev = synth.synth_events(1003, 10_000_000, shape[1], shape[0], 80_000)
    r = harness_taf(T(synth.to_xytp_f64(ev)), shape, shape, 0, 80_000, None)
    pack_big("state", r["state"], d)
    pack_big("leaky", r["leaky"], d)
    d["u8_sha"] = np.array(sha(r["u8"]))
    idx = sample_idx(r["u8"].size, 65536)
    d["u8_idx"] = idx
    d["u8_val"] = r["u8"].reshape(-1)[idx]
    save("mpx_taf_native", d)
    # gen4 recipe: coordinates down-scaled to 512x640 before encoding (generate_taf.py:216-219)
    d = {}
    ev = synth.synth_events(1013, 2_000_000, shape[1], shape[0], 80_000, hotspot=True)
    r = harness_taf(T(synth.to_xytp_f64(ev)), shape, tshape, 0, 80_000, None)
    pack_big("state", r["state"], d)
    d["u8_sha"] = np.array(sha(r["u8"]))
    idx = sample_idx(r["u8"].size, 65536)
    d["u8_idx"] = idx
    d["u8_val"] = r["u8"].reshape(-1)[idx]
    # EV and ECI in the same down-scale mode
    native, u8 = harness_ev(T(synth.to_xytp_f64(ev)), shape, tshape, 80_000, 80_000)
    pack_big("ev_native", native, d)
    d["ev_u8_sha"] = np.array(sha(u8))
    native, u8 = harness_eci(T(synth.to_xytp_f64(ev))[-200_000:], shape, tshape)
    pack_big("eci_native", native, d)
    save("mpx_downscale", d)


def gen_taf_grow():
    """The K-growing branch of taf_cuda (generate_taf.py:50-53): a ``past_volume`` with volume_bins - 1 slots.  The
    concatenated FIFO is not cut; slot 0 of the cells without events becomes -6000 (:53).  Dead in the reference's harness
    (it always passes volume_bins slots) but part of the function's contract.  Also what the reference does with other
    slot counts: an entirely empty window returns the short volume and the ``.view`` of :55 raises, as it does for
    fewer than volume_bins - 1 slots."""
    H, W, ev = tiny_events()
    d = {}
    for K in (8, 4):
        rng = np.random.default_rng(50 + K)
        past = (-rng.integers(0, 40, size=(H, W, 2, K - 1))).astype(np.float32) - rng.random((H, W, 2, K - 1)).astype(np.float32)
        w = ev[:150]
        w5 = np.concatenate([w, np.zeros((len(w), 1))], axis=1)
        view, state, _ = ref_taf.generate_taf_cuda(T(w5), (H, W), T(past.copy()), K)
        d[f"k{K}_past"], d[f"k{K}_view"], d[f"k{K}_state"] = past, view.numpy().copy(), state.numpy().copy()
        assert state.shape == (H, W, 2, K)
        for bad, what in ((T(past.copy()), w5[:0]), (T(past[..., :K - 2].copy()), w5)):
            try:
                ref_taf.generate_taf_cuda(T(what), (H, W), bad, K)
                raise SystemExit("the reference was expected to raise here")
            except RuntimeError:
                pass
    save("tiny_taf_grow", d)


if __name__ == "__main__":
    which = sys.argv[1:] or ["tiny", "gen1", "mpx", "grow"]
    if "grow" in which:
        gen_taf_grow()
    if "tiny" in which:
        gen_tiny()
    if "gen1" in which:
        gen_gen1()
    if "mpx" in which:
        gen_mpx()
