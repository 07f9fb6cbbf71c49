"""Event-stream -> tensor encoders: the reference's function signatures over the gfx950 kernels.

Mirror of the module-level encoder functions of the reference's ``generate_*.py`` scripts (same
names, argument meaning, return tuples and error behaviour), so their ``__main__`` harnesses run
unchanged on top of this module:

    generate_eventframe(events, shape)                         generate_eventcountimage.py:19-41
    generate_agile_event_volume_cuda(events, shape, ...)       generate_eventvolume.py:15-42
    generate_leaky_cuda(events, shape, lamdas, memory, now)    generate_surfaceofactiveevents.py:71-80
    generate_taf_cuda(events, shape, past_volume, volume_bins) generate_taf.py:60-67
    leaky_transform(ecd)                                       generate_taf.py:69-76

``events`` is the reference's device tensor, ``(N, 4 or 5)`` float64 ``[x, y, t, p, (z)]``.
The ``encode_*_dat`` functions are the fused fast path over raw 8-byte DAT records: window
selection, f64 time normalisation, coordinate down-scale, encode, leaky transform and uint8
truncation (the harness lines ``generate_taf.py:197-235``) happen on device in one call.

All compute goes through ``libfrlw_evd.so``; a missing library raises (no fallback).
"""
from __future__ import annotations

import ctypes as C
import time

import torch

from . import _lib

_WORKSPACES = {}
TUNING = None  # a _lib.FrlwTuning to attach to every encoder call (experiments / tests forcing a path); None = defaults
FAST_MIN_EVENTS = 1_000_000  # single streams shorter than this take the general TAF path (encode_taf_dat, fast="auto")
# False: every encoder takes the general path (the batched entry points raise NotImplementedError like on a device that failed the
# lane-order self-test).  Set by dist.agree_fast_path() when ANY rank of the job failed it: results do not depend on the path,
# step times do, and ranks that wait for each other in a collective should not run different kernels.
FAST_PATH_ENABLED = True


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _workspace(n, H, W, device):
    lib = _lib.load()
    need = lib.frlw_encoder_workspace_bytes(int(n), int(H), int(W))
    if need == 0:
        raise ValueError(f"unsupported encode shape {H}x{W}")
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() < need:
        ws = _new_workspace(ws, need, device)
        _WORKSPACES[key] = ws
    return ws


def _new_workspace(old, need, device):
    """A fresh workspace with a zeroed header (``frlw_workspace_init``): the deferred status word starts clean.  An
    outgrown workspace hands over what it has accumulated first."""
    if old is not None:
        _raise_deferred_of(old, "encoder call (workspace outgrown)")
    ws = torch.empty(int(need * 1.25) + 4096, dtype=torch.uint8, device=device)
    _lib.check(_lib.load().frlw_workspace_init(C.c_void_p(ws.data_ptr()), ws.numel(), _stream()), "frlw_workspace_init")
    return ws


def _raise_deferred_of(ws, what):
    st = C.c_int(0)
    _lib.check(_lib.load().frlw_encoder_deferred_status(C.c_void_p(ws.data_ptr()), _stream(), C.byref(st)), what)
    _lib.check(st.value, what)


def raise_deferred(what="encoder calls since the last check"):
    """For callers that pass ``check=False``: ONE host synchronisation that surfaces the data-dependent status of every
    encoder call made on the current stream since the last check (IndexError / ValueError like the checked calls).  An
    unchecked fast-path call whose events leave the sequence span or the frame writes nothing -- call this before
    trusting its state / outputs (``e2e.SyntheticTafSource.encode_u8`` does, once per batch)."""
    stream = torch.cuda.current_stream().cuda_stream
    for key, ws in list(_WORKSPACES.items()):
        if key[-1] == stream and ws.device.index == torch.cuda.current_device():
            _raise_deferred_of(ws, what)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def fast_path_ok(device=None):
    """The library's verdict on the LDS lane-order property for ``device`` (default: the current one): True = the fast paths run
    there.  Runs the one-time self-test if no fast-path call has yet (0.7 ms, one host synchronisation per process and device)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(device):
        ws = _workspace(1, 8, 8, device)
        ok = C.c_int(0)
        _lib.check(_lib.load().frlw_fast_path_verdict(_ptr(ws), ws.numel(), _stream(), C.byref(ok)), "frlw_fast_path_verdict")
    return bool(ok.value)


def _events_f64(events):
    if not events.is_cuda:
        raise RuntimeError("events must be a CUDA (ROCm) tensor: this path has no CPU implementation")
    if events.dtype != torch.float64 or events.dim() != 2 or events.shape[1] < 4:
        raise ValueError("events must be (N, >=4) float64 [x, y, t, p]")
    ev = events.contiguous()
    return ev, _lib.FrlwEvents(ev.data_ptr(), ev.shape[0], _lib.LAYOUT_XYTP_F64, ev.shape[1], None, None, 0, 0,
                               C.pointer(TUNING) if TUNING is not None else None)


def _events_dat(dat, xmap=None, ymap=None):
    """dat: uint8 (N, 8) / int64 (N,) / any 8-byte-per-event CUDA tensor of raw DAT records."""
    if not dat.is_cuda:
        raise RuntimeError("dat must be a CUDA (ROCm) tensor")
    d = dat.contiguous()
    nbytes = d.numel() * d.element_size()
    if nbytes % 8:
        raise ValueError("DAT records are 8 bytes each")
    mw = mh = 0
    if (xmap is None) != (ymap is None):
        raise ValueError("xmap and ymap come together")
    if xmap is not None:
        for m in (xmap, ymap):  # the kernels read both tables as uint16_t
            if m.dtype not in (torch.int16, torch.uint16) or not m.is_cuda or not m.is_contiguous() or m.dim() != 1:
                raise ValueError("coordinate maps must be contiguous 1-D int16 / uint16 CUDA tensors")
        mw, mh = xmap.numel(), ymap.numel()
    return d, _lib.FrlwEvents(d.data_ptr(), nbytes // 8, _lib.LAYOUT_DAT8, 0, _ptr(xmap), _ptr(ymap), mw, mh,
                              C.pointer(TUNING) if TUNING is not None else None)


def _finish(ws, what):
    """Synchronise and surface data-dependent errors the way torch would (IndexError): the status of this call and of any
    unchecked call on the same workspace before it, read and cleared in one go (``frlw_encoder_deferred_status``)."""
    _raise_deferred_of(ws, what)


def coordinate_maps(sensor_shape, shape, device):
    """uint16 x/y tables equal to the harness' ``x * rw``, ``y * rh`` + ``.long()`` (generate_taf.py:216-219)."""
    Hs, Ws = sensor_shape
    H, W = shape
    rw, rh = W / Ws, H / Hs
    xmap = (torch.arange(Ws, dtype=torch.float64) * rw).long().to(torch.int16)
    ymap = (torch.arange(Hs, dtype=torch.float64) * rh).long().to(torch.int16)
    return xmap.to(device), ymap.to(device)


# ------------------------------------------------------------------------------------------------
# reference signatures
# ------------------------------------------------------------------------------------------------
def generate_eventframe(events, shape):
    """generate_eventcountimage.py:19-41 -> (f32 (2, H, W) * 255, seconds)."""
    tick = time.time()
    H, W = int(shape[0]), int(shape[1])
    ev, desc = _events_f64(events)
    out = torch.empty((2, H, W), dtype=torch.float32, device=ev.device)
    ws = _workspace(ev.shape[0], H, W, ev.device)
    _lib.check(_lib.load().frlw_eci_encode(C.byref(desc), H, W, _ptr(out), None, _ptr(ws), ws.numel(), _stream()),
               "generate_eventframe")
    _finish(ws, "generate_eventframe")
    return out, time.time() - tick


def generate_agile_event_volume_cuda(events, shape, events_window=50000, volume_bins=5):
    """generate_eventvolume.py:15-42 -> (f32 (2*bins, H, W), seconds).  ``events_window`` is unused there too."""
    tick = time.time()
    H, W = int(shape[0]), int(shape[1])
    ev, desc = _events_f64(events)
    out = torch.empty((2 * volume_bins, H, W), dtype=torch.float32, device=ev.device)
    ws = _workspace(ev.shape[0], H, W, ev.device)
    _lib.check(_lib.load().frlw_ev_encode(C.byref(desc), H, W, int(volume_bins), 0, 1, _ptr(out), None, _ptr(ws),
                                          ws.numel(), _stream()), "generate_agile_event_volume_cuda")
    _finish(ws, "generate_agile_event_volume_cuda")
    return out, time.time() - tick


def generate_leaky_cuda(events, shape, lamdas, memory, now):
    """generate_surfaceofactiveevents.py:71-80 -> (f32 (2*len(lamdas), H, W), memory (2, H, W), seconds)."""
    tick = time.time()
    H, W = int(shape[0]), int(shape[1])
    ev, desc = _events_f64(events)
    lam = (C.c_double * len(lamdas))(*[float(l) for l in lamdas])
    out = torch.empty((2 * len(lamdas), H, W), dtype=torch.float32, device=ev.device)
    mem_out = torch.empty((2, H, W), dtype=torch.float32, device=ev.device)
    mem_in = None if memory is None else memory.to(torch.float32).contiguous()
    ws = _workspace(ev.shape[0], H, W, ev.device)
    _lib.check(_lib.load().frlw_sae_encode(C.byref(desc), H, W, lam, len(lamdas), _ptr(mem_in), _ptr(mem_out),
                                           int(now), 0, _ptr(out), None, _ptr(ws), ws.numel(), _stream()),
               "generate_leaky_cuda")
    _finish(ws, "generate_leaky_cuda")
    return out, mem_out, time.time() - tick


def generate_taf_cuda(events, shape, past_volume=None, volume_bins=5):
    """generate_taf.py:60-67 -> (view (2K, H, W), state (H, W, 2, K), seconds); inputs are not mutated."""
    tick = time.time()
    H, W = int(shape[0]), int(shape[1])
    K = int(volume_bins)
    if past_volume is None:
        raise TypeError("past_volume is required (the reference concatenates it, generate_taf.py:44)")
    ev, desc = _events_f64(events)
    if tuple(past_volume.shape) == (H, W, 2, K - 1) and K > 1:
        # the K-growing branch, generate_taf.py:50-53: the concatenated FIFO [old..., mean] is not cut and slot 0 of the
        # cells without events becomes -6000.  Same as a K-slot step on [-5999, old...]: cells with events drop slot 0,
        # the others age it to -5999 - 1 = -6000 (exact in f32).  An entirely empty window returns the short volume and
        # the reference's .view (:55) raises.
        if ev.shape[0] == 0:
            raise RuntimeError(f"shape '[{2 * K}, {H}, {W}]' is invalid for input of size {H * W * 2 * (K - 1)} "
                               "(generate_taf.py:40-41,55: a window without events keeps the short past_volume)")
        state = torch.cat([torch.full((H, W, 2, 1), -5999.0, dtype=torch.float32, device=ev.device),
                           past_volume.to(torch.float32)], dim=3).contiguous()
    elif tuple(past_volume.shape) != (H, W, 2, K):
        raise RuntimeError("past_volume must be (H, W, 2, volume_bins) or (H, W, 2, volume_bins - 1): any other slot count "
                           "fails the reference's .view (generate_taf.py:48-55)")
    else:
        state = past_volume.to(torch.float32).contiguous().clone()
    view = torch.empty((2 * K, H, W), dtype=torch.float32, device=ev.device)
    ws = _workspace(ev.shape[0], H, W, ev.device)
    _lib.check(_lib.load().frlw_taf_encode(C.byref(desc), H, W, K, 0, 1, 1, _ptr(state), _ptr(view), None, 0,
                                           _ptr(ws), ws.numel(), _stream()), "generate_taf_cuda")
    _finish(ws, "generate_taf_cuda")
    return view, state, time.time() - tick


def leaky_transform(ecd):
    """generate_taf.py:69-76."""
    src = ecd.to(torch.float32).contiguous()
    out = torch.empty_like(src)
    _lib.check(_lib.load().frlw_leaky_transform(_ptr(src), src.numel(), _ptr(out), None, _stream()), "leaky_transform")
    return out


def resize_nearest(volume, target_shape):
    """``F.interpolate(volume[None], size=target_shape, mode='nearest')[0]`` (generate_eventvolume.py:149)."""
    src = volume.contiguous()
    Cn, H, W = src.shape
    Ho, Wo = int(target_shape[0]), int(target_shape[1])
    out = torch.empty((Cn, Ho, Wo), dtype=src.dtype, device=src.device)
    fn = _lib.load().frlw_resize_nearest_u8 if src.dtype == torch.uint8 else _lib.load().frlw_resize_nearest_f32
    if src.dtype not in (torch.uint8, torch.float32):
        raise ValueError("resize_nearest: float32 or uint8")
    _lib.check(fn(_ptr(src), Cn, H, W, Ho, Wo, _ptr(out), _stream()), "resize_nearest")
    return out


def quantize_u8(volume, clip255=False):
    """``np.where(v > 255, 255, v).astype(np.uint8)`` / plain ``.astype(np.uint8)`` on device."""
    src = volume.to(torch.float32).contiguous()
    out = torch.empty(src.shape, dtype=torch.uint8, device=src.device)
    _lib.check(_lib.load().frlw_quantize_u8(_ptr(src), src.numel(), int(bool(clip255)), _ptr(out), _stream()),
               "quantize_u8")
    return out


# ------------------------------------------------------------------------------------------------
# fused DAT-record fast path (harness glue on device)
# ------------------------------------------------------------------------------------------------
def _batch_workspace(n, n_seq, H, W, window_us, device, slot="batch"):
    need = _lib.load().frlw_taf_batch_workspace_bytes(int(n), int(n_seq), int(H), int(W), int(window_us))
    if need == 0:
        return None
    key = (slot, device.index, torch.cuda.current_stream().cuda_stream)
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() < need:
        ws = _new_workspace(ws, need, device)
        _WORKSPACES[key] = ws
    return ws


def encode_taf_batch(dat, seq_offsets, shape, state, t_start, window_us=10000, n_windows=8, volume_bins=8,
                     want_view=False, want_u8=True, flip_k=True, xmap=None, ymap=None, check=True):
    """The harness loop generate_taf.py:193-235 for a batch of independent sequences in one launch sequence
    (``frlw_taf_encode_batch``, csrc/taf_fast.hip).

    ``dat``: the raw DAT records of all sequences back to back; sequence ``s`` owns records
    ``[seq_offsets[s], seq_offsets[s + 1])`` and starts at ``t_start[s]`` (an int applies to all).  ``state``
    ``(B, H, W, 2, K)`` is updated IN PLACE.  Returns ``(u8 (B, K, 2, H, W) or None, view (B, 2K, H, W) or None)``.

    Every event must lie inside its sequence's ``[t_start, t_start + n_windows * window_us]`` and inside the frame;
    with ``check`` a violation raises ``ValueError`` / ``IndexError`` and ``state`` is untouched.  With ``check=False``
    nothing synchronises and a violation leaves ``state`` AND the returned tensors unwritten: the caller owes one
    ``raise_deferred()`` before it uses them (the status of every unchecked call accumulates in the workspace).  Raises
    ``NotImplementedError`` if the shape / window does not fit the fast path's 4-byte records, or if the device failed
    the LDS lane-order self-test the library runs on its first fast-path call (then use ``encode_taf_dat``).
    """
    H, W = int(shape[0]), int(shape[1])
    K = int(volume_bins)
    offs = [int(o) for o in seq_offsets]
    B = len(offs) - 1
    if B < 1 or B > _lib.MAX_SEQUENCES:
        raise ValueError(f"1..{_lib.MAX_SEQUENCES} sequences per call")
    if not FAST_PATH_ENABLED:
        raise NotImplementedError("the fast path is switched off for this job (dist.agree_fast_path: a rank failed the self-test)")
    t0 = [int(t_start)] * B if not hasattr(t_start, "__len__") else [int(t) for t in t_start]
    assert state.dtype == torch.float32 and state.is_contiguous() and tuple(state.shape) == (B, H, W, 2, K)
    d, desc = _events_dat(dat, xmap, ymap)
    ws = _batch_workspace(offs[-1] - offs[0], B, H, W, window_us, d.device)
    if ws is None:
        raise NotImplementedError("shape outside the fast TAF path")
    u8 = torch.empty((B, K, 2, H, W), dtype=torch.uint8, device=d.device) if want_u8 else None
    view = torch.empty((B, 2 * K, H, W), dtype=torch.float32, device=d.device) if want_view else None
    flags = _lib.TAF_U8_FLIP_K if flip_k else 0
    rc = _lib.load().frlw_taf_encode_batch(C.byref(desc), (C.c_int64 * (B + 1))(*offs), (C.c_int64 * B)(*t0), B, H, W, K,
                                           int(window_us), int(n_windows), _ptr(state), _ptr(view), _ptr(u8), flags,
                                           _ptr(ws), ws.numel(), _stream())
    if rc == _lib.FRLW_ERR_UNSUPPORTED:
        raise NotImplementedError("window / shape outside the fast TAF path")
    _lib.check(rc, "encode_taf_batch")
    if check:
        _finish(ws, "encode_taf_batch")
    return u8, view


def _or_reduce_window_masks(masks, group):
    """OR of the 64-bit window masks over the ranks of ``group``, in place.  RCCL / NCCL have no bitwise reductions: the bits
    travel as a (B, 64) int32 tensor under MAX (256 bytes per sequence instead of 8 -- still nothing)."""
    import torch.distributed as dist
    shifts = torch.arange(64, device=masks.device, dtype=torch.int64)
    bits = ((masks[:, None] >> shifts) & 1).to(torch.int32)
    dist.all_reduce(bits, op=dist.ReduceOp.MAX, group=group)
    masks.copy_((bits.to(torch.int64) << shifts).sum(dim=1))


def encode_taf_stripe(dat, seq_offsets, shape, stripe, state, t_start, window_us=10000, n_windows=8, volume_bins=8,
                      want_view=False, want_u8=True, flip_k=True, xmap=None, ymap=None, check=True, group=None,
                      exchange=None):
    """Row-stripe sharding of ONE frame over several GPUs: this rank encodes rows ``stripe = (y_lo, y_hi)`` of
    the ``shape = (H, W)`` frame from the WHOLE stream ``dat`` (events of other rows are skipped) into its stripe of the FIFO
    state ``(B, y_hi - y_lo, W, 2, K)`` (in place) and of the outputs.  No halo, no event exchange; the one global quantity --
    "a window without any event in the whole frame leaves the state untouched", generate_taf.py:40-41 -- is OR-reduced
    between the two halves of the encode: 8 bytes per sequence over ``torch.distributed`` (``group``; RCCL on the GPU box).
    The stripes of all ranks put together equal ``encode_taf_batch`` on the whole frame, bit for bit.

    ``exchange(masks)``: replaces the collective (tests emulate the other stripes in one process: it receives the
    ``(B,)`` int64 device tensor of this stripe's masks and ORs the others' into it in place)."""
    import torch.distributed as dist
    H, W = int(shape[0]), int(shape[1])
    y_lo, y_hi = int(stripe[0]), int(stripe[1])
    rows = y_hi - y_lo
    if not (0 <= y_lo < y_hi <= H):
        raise ValueError("stripe must be (y_lo, y_hi) with 0 <= y_lo < y_hi <= H")
    K = int(volume_bins)
    offs = [int(o) for o in seq_offsets]
    B = len(offs) - 1
    if B < 1 or B > _lib.MAX_SEQUENCES:
        raise ValueError(f"1..{_lib.MAX_SEQUENCES} sequences per call")
    t0 = [int(t_start)] * B if not hasattr(t_start, "__len__") else [int(t) for t in t_start]
    assert state.dtype == torch.float32 and state.is_contiguous() and tuple(state.shape) == (B, rows, W, 2, K)
    d, desc = _events_dat(dat, xmap, ymap)
    ws = _batch_workspace(offs[-1] - offs[0], B, rows, W, window_us, d.device, slot=("stripe", y_lo, y_hi))
    if ws is None:
        raise NotImplementedError("shape outside the fast TAF path")
    lib = _lib.load()
    c_offs, c_t0 = (C.c_int64 * (B + 1))(*offs), (C.c_int64 * B)(*t0)
    rc = lib.frlw_taf_stripe_partition(C.byref(desc), c_offs, c_t0, B, H, W, y_lo, rows, K, int(window_us), int(n_windows),
                                       _ptr(ws), ws.numel(), _stream())
    if rc == _lib.FRLW_ERR_UNSUPPORTED:
        raise NotImplementedError("window / shape outside the fast TAF path")
    _lib.check(rc, "encode_taf_stripe (partition)")
    off = int(lib.frlw_taf_stripe_window_masks(_ptr(ws))) - ws.data_ptr()
    masks = ws[off:off + 8 * B].view(torch.int64)  # the kernels' own words: reduced in place, read by the second half
    if exchange is not None:
        exchange(masks)
    elif dist.is_available() and dist.is_initialized():  # (also a one-rank group: the collective is the same code path)
        _or_reduce_window_masks(masks, group)
    u8 = torch.empty((B, K, 2, rows, W), dtype=torch.uint8, device=d.device) if want_u8 else None
    view = torch.empty((B, 2 * K, rows, W), dtype=torch.float32, device=d.device) if want_view else None
    flags = _lib.TAF_U8_FLIP_K if flip_k else 0
    _lib.check(lib.frlw_taf_stripe_finish(C.byref(desc), c_offs, c_t0, B, H, W, y_lo, rows, K, int(window_us), int(n_windows),
                                          _ptr(state), _ptr(view), _ptr(u8), flags, _ptr(ws), ws.numel(), _stream()),
               "encode_taf_stripe (finish)")
    if check:
        _finish(ws, "encode_taf_stripe")
    return u8, view


def encode_taf_dat(dat, shape, state, t_start, window_us=10000, n_windows=8, volume_bins=8, want_view=False,
                   want_u8=True, flip_k=True, xmap=None, ymap=None, check=True, fast="auto"):
    """generate_taf.py:193-235 in one device pass; ``state`` (H, W, 2, K) is updated IN PLACE.

    Returns ``(u8 (K, 2, H, W) or None, view (2K, H, W) or None)``.  With ``flip_k`` the uint8 volume is
    newest-slot-first like ``np.flip(ecd, axis=0)`` (:229): ``u8[:4]`` is the bins4 file, ``u8[4:]`` bins8.

    ``fast``: run the batched fast path with one sequence (csrc/taf_fast.hip); ``"auto"`` = for CHECKED calls on streams
    of at least ``FAST_MIN_EVENTS`` events (below that its extra launch costs more than it saves: 61 vs 64 us at 1 M
    events, the break-even).  It needs every event inside ``[t_start, t_start + n_windows * window_us]``; if the device
    check says otherwise, or the window does not fit its 4-byte records, the general path (csrc/encoders.hip) runs -- same
    bits either way.  An unchecked call cannot see that verdict, so ``"auto"`` keeps it on the general path, which places
    such events like the reference does; ``fast=True`` with ``check=False`` is an explicit opt-in whose caller owes a
    ``raise_deferred()`` (a violation leaves ``state`` and the outputs unwritten).
    """
    H, W = int(shape[0]), int(shape[1])
    K = int(volume_bins)
    assert state.dtype == torch.float32 and state.is_contiguous() and tuple(state.shape) == (H, W, 2, K)
    n = dat.numel() * dat.element_size() // 8
    if fast == "auto":
        fast = bool(check) and n >= FAST_MIN_EVENTS
    fast = fast and FAST_PATH_ENABLED
    if fast:
        if check:
            # The fall-back below takes a ValueError / IndexError of the fast call to mean "THIS call wrote nothing".  The status
            # word it reads is sticky: an earlier unchecked call on the same workspace may have left its error there.  Drain
            # that first and let it propagate -- it belongs to the earlier call, and swallowing it here would run the general
            # path on a state the fast call has already stepped.
            pending = _WORKSPACES.get(("batch", dat.device.index, torch.cuda.current_stream().cuda_stream))
            if pending is not None:
                _raise_deferred_of(pending, "an earlier unchecked encoder call on this stream")
        try:
            u8, view = encode_taf_batch(dat, [0, n], (H, W), state.view(1, H, W, 2, K), t_start, window_us, n_windows, K,
                                        want_view, want_u8, flip_k, xmap, ymap, check)
            return (None if u8 is None else u8[0]), (None if view is None else view[0])
        except NotImplementedError:
            pass
        except (ValueError, IndexError):
            pass  # out-of-span / out-of-frame events: nothing was written, the general path places or reports them
    d, desc = _events_dat(dat, xmap, ymap)
    u8 = torch.empty((K, 2, H, W), dtype=torch.uint8, device=d.device) if want_u8 else None
    view = torch.empty((2 * K, H, W), dtype=torch.float32, device=d.device) if want_view else None
    ws = _workspace(desc.n, H, W, d.device)
    flags = _lib.TAF_U8_FLIP_K if flip_k else 0
    _lib.check(_lib.load().frlw_taf_encode(C.byref(desc), H, W, K, int(t_start), int(window_us), int(n_windows),
                                           _ptr(state), _ptr(view), _ptr(u8), flags, _ptr(ws), ws.numel(), _stream()),
               "encode_taf_dat")
    if check:
        _finish(ws, "encode_taf_dat")
    return u8, view


def encode_taf_label(dat, shape, state, start_time, window_us, bins, volume_bins=8, flip_k=True, xmap=None, ymap=None,
                     check=True, fast="auto", stale_transforms=0):
    """One annotation timestamp of the TAF harness (generate_taf.py:197-235): the time-sorted records ``dat`` of
    ``[start_time, start_time + bins * window_us]`` (``dat_io.taf_label_slices`` says which) through ``bins`` windows;
    ``state`` is updated in place, the newest-first uint8 volume (K, 2, H, W) is returned.

    ``bins`` is unbounded in the reference (the first label of a file spans everything before it); one launch takes at
    most 64 windows, so longer labels run in groups of 64 with the FIFO state carried -- an event on a group boundary
    belongs to the later window, as in the reference's ``z`` column (:197-203).

    ``bins == 0`` (a label that rounds onto the previous one, :181): the reference's window loop does not run and its
    ``volume`` variable still holds the PREVIOUS label's volume -- after ``leaky_transform``.  Lines :226-227 transform it
    again, so the file holds ``uint8(leaky_transform(leaky_transform(view)))``: 255 where the first transform gave 0, NaN -> 0
    where it gave more than 1, values in (0, 1) land above 255 and wrap in the uint8 cast.  Reproduced here as it is
    (the state has not moved since the previous label, so ``view`` is the state's); ``stale_transforms`` = how many such
    labels directly precede this one (each adds one more application).  Pinned by the ``seqA_603000`` files of
    tests/golden/harness.npz, which the reference script wrote.
    """
    H, W = int(shape[0]), int(shape[1])
    K = int(volume_bins)
    lib = _lib.load()
    max_w = 64
    n = dat.numel() * dat.element_size() // 8
    if bins <= 0:
        vol = state.permute(3, 2, 0, 1).contiguous()  # (K, 2, H, W), the view of :55
        for _ in range(2 + int(stale_transforms)):    # :226-227 on the stale, already transformed volume
            vol = leaky_transform(vol)
        u8 = quantize_u8(vol)
        return torch.flip(u8, dims=[0]).contiguous() if flip_k else u8
    cuts = [0]
    if bins > max_w and n:
        t = dat.contiguous().view(torch.int32).reshape(-1, 2)[:, 0].to(torch.int64) & 0xFFFFFFFF
        bounds = torch.tensor([start_time + g * max_w * window_us for g in range(1, (bins + max_w - 1) // max_w)],
                              dtype=torch.int64, device=t.device)
        cuts += [int(c) for c in torch.searchsorted(t, bounds, right=False).tolist()]
    elif bins > max_w:
        cuts += [0] * ((bins + max_w - 1) // max_w - 1)
    cuts.append(n)
    rows = dat.contiguous().view(torch.uint8).reshape(-1, 8)
    u8 = None
    for g in range(len(cuts) - 1):
        nw = min(max_w, bins - g * max_w)
        last = g == len(cuts) - 2
        u8, _ = encode_taf_dat(rows[cuts[g]:cuts[g + 1]], (H, W), state, start_time + g * max_w * window_us, window_us, nw, K,
                               want_view=False, want_u8=last, flip_k=flip_k, xmap=xmap, ymap=ymap, check=check, fast=fast)
    return u8


def encode_ev_dat(dat, shape, t_end, window_us, volume_bins=5, want_f32=True, want_u8=False, xmap=None, ymap=None,
                  check=True, fast="auto"):
    """generate_eventvolume.py:139-157 on device -> (f32 (2*bins, H, W) or None, u8 or None).

    ``fast``: run the batched path with one label window (``frlw_ev_encode_batch``: two launches -- the chunk-major scatter and
    ``kf_ev_fadd`` -- instead of the general path's five; 40 against 51 us for 1 M events at 304x240); ``"auto"`` = for CHECKED
    calls of at least ``FAST_MIN_EVENTS`` events.  It needs every event at or in front of ``t_end``; if the device check says
    otherwise, or the window does not fit its 4-byte records, the general path runs -- same bits either way.  ``fast=True`` with
    ``check=False`` is an explicit opt-in whose caller owes a ``raise_deferred()`` (a violation leaves the outputs unwritten)."""
    H, W = int(shape[0]), int(shape[1])
    n = dat.numel() * dat.element_size() // 8
    if fast == "auto":
        fast = bool(check) and n >= FAST_MIN_EVENTS
    fast = fast and FAST_PATH_ENABLED
    if fast:
        if check:  # (an earlier unchecked call's error must not be mistaken for this call's: encode_taf_dat explains)
            pending = _WORKSPACES.get(("batch", dat.device.index, torch.cuda.current_stream().cuda_stream))
            if pending is not None:
                _raise_deferred_of(pending, "an earlier unchecked encoder call on this stream")
        try:
            out, u8 = encode_ev_batch(dat, [0, n], (H, W), t_end, window_us, volume_bins, want_f32, want_u8, xmap, ymap, check)
            return (None if out is None else out[0]), (None if u8 is None else u8[0])
        except NotImplementedError:
            pass
        except (ValueError, IndexError):
            pass  # an event behind t_end / outside the frame: nothing was written, the general path places or reports it
    d, desc = _events_dat(dat, xmap, ymap)
    out = torch.empty((2 * volume_bins, H, W), dtype=torch.float32, device=d.device) if want_f32 else None
    u8 = torch.empty((2 * volume_bins, H, W), dtype=torch.uint8, device=d.device) if want_u8 else None
    ws = _workspace(desc.n, H, W, d.device)
    _lib.check(_lib.load().frlw_ev_encode(C.byref(desc), H, W, int(volume_bins), int(t_end), int(window_us), _ptr(out),
                                          _ptr(u8), _ptr(ws), ws.numel(), _stream()), "encode_ev_dat")
    if check:
        _finish(ws, "encode_ev_dat")
    return out, u8


def encode_ev_batch(dat, seq_offsets, shape, t_end, window_us, volume_bins=5, want_f32=True, want_u8=False, xmap=None, ymap=None,
                    check=True):
    """generate_eventvolume.py:139-157 for a batch of independent label windows in one launch sequence
    (``frlw_ev_encode_batch``, csrc/taf_fast.hip): sequence ``s`` owns the records ``[seq_offsets[s], seq_offsets[s + 1])`` of
    ``dat`` and ends at ``t_end[s]`` (an int applies to all); events with ``t <= t_end - window_us`` are dropped like the
    harness does.  Returns ``(f32 (B, 2 * bins, H, W) or None, u8 or None)`` -- bit for bit what ``encode_ev_dat`` gives per
    sequence.  An event behind its ``t_end`` is outside the contract: with ``check`` it raises ``ValueError`` (nothing is
    written); unchecked callers owe a ``raise_deferred()``.  ``NotImplementedError``: shape / window outside the path."""
    H, W = int(shape[0]), int(shape[1])
    offs = [int(o) for o in seq_offsets]
    B = len(offs) - 1
    if B < 1 or B > _lib.MAX_SEQUENCES:
        raise ValueError(f"1..{_lib.MAX_SEQUENCES} sequences per call")
    if not FAST_PATH_ENABLED:
        raise NotImplementedError("the fast path is switched off for this job (dist.agree_fast_path: a rank failed the self-test)")
    te = [int(t_end)] * B if not hasattr(t_end, "__len__") else [int(t) for t in t_end]
    d, desc = _events_dat(dat, xmap, ymap)
    ws = _batch_workspace(offs[-1] - offs[0], B, H, W, window_us, d.device)
    if ws is None:
        raise NotImplementedError("shape outside the batched Event Volume path")
    out = torch.empty((B, 2 * volume_bins, H, W), dtype=torch.float32, device=d.device) if want_f32 else None
    u8 = torch.empty((B, 2 * volume_bins, H, W), dtype=torch.uint8, device=d.device) if want_u8 else None
    rc = _lib.load().frlw_ev_encode_batch(C.byref(desc), (C.c_int64 * (B + 1))(*offs), (C.c_int64 * B)(*te), B, H, W,
                                          int(volume_bins), int(window_us), _ptr(out), _ptr(u8), _ptr(ws), ws.numel(), _stream())
    if rc == _lib.FRLW_ERR_UNSUPPORTED:
        raise NotImplementedError("window / shape outside the batched Event Volume path")
    _lib.check(rc, "encode_ev_batch")
    if check:
        _finish(ws, "encode_ev_batch")
    return out, u8


def encode_eci_dat(dat, shape, want_f32=True, want_u8=False, xmap=None, ymap=None, check=True):
    """generate_eventcountimage.py:155-180 on device (the caller has cut the last ``events_window`` records)."""
    H, W = int(shape[0]), int(shape[1])
    d, desc = _events_dat(dat, xmap, ymap)
    out = torch.empty((2, H, W), dtype=torch.float32, device=d.device) if want_f32 else None
    u8 = torch.empty((2, H, W), dtype=torch.uint8, device=d.device) if want_u8 else None
    ws = _workspace(desc.n, H, W, d.device)
    _lib.check(_lib.load().frlw_eci_encode(C.byref(desc), H, W, _ptr(out), _ptr(u8), _ptr(ws), ws.numel(), _stream()),
               "encode_eci_dat")
    if check:
        _finish(ws, "encode_eci_dat")
    return out, u8


def encode_sae_dat(dat, shape, lamdas, memory, now, window_us, want_f32=True, want_u8=False, xmap=None, ymap=None,
                   check=True):
    """generate_surfaceofactiveevents.py:183-194 on device -> (f32 or None, u8 or None, new memory)."""
    H, W = int(shape[0]), int(shape[1])
    d, desc = _events_dat(dat, xmap, ymap)
    lam = (C.c_double * len(lamdas))(*[float(l) for l in lamdas])
    out = torch.empty((2 * len(lamdas), H, W), dtype=torch.float32, device=d.device) if want_f32 else None
    u8 = torch.empty((2 * len(lamdas), H, W), dtype=torch.uint8, device=d.device) if want_u8 else None
    mem_out = torch.empty((2, H, W), dtype=torch.float32, device=d.device)
    mem_in = None if memory is None else memory.to(torch.float32).contiguous()
    ws = _workspace(desc.n, H, W, d.device)
    _lib.check(_lib.load().frlw_sae_encode(C.byref(desc), H, W, lam, len(lamdas), _ptr(mem_in), _ptr(mem_out), int(now),
                                           int(window_us), _ptr(out), _ptr(u8), _ptr(ws), ws.numel(), _stream()),
               "encode_sae_dat")
    if check:
        _finish(ws, "encode_sae_dat")
    return out, u8, mem_out
