"""Synthetic Prophesee-shaped event streams.

All draws come from ``numpy.random.default_rng(seed)`` (PCG64), so the same seed
regenerates the same stream in the golden-vector script, in the tests and in
``bench.py`` on any machine.  Two packings are offered:

* ``to_xytp_f64``  - the reference's device tensor, ``(N, 4)`` float64 ``[x, y, t, p]``
  (``generate_eventvolume.py:135``),
* ``to_dat8``      - raw 8-byte Prophesee DAT ``Event2D`` records, ``t:u32`` then
  ``x | y << 14 | p << 28`` (``src/io/dat_events_tools.py:16,96-98``).
"""
from __future__ import annotations

import numpy as np

DAT_DTYPE = np.dtype([("t", "<u4"), ("_", "<u4")])


def synth_events(seed: int, n: int, width: int, height: int, t_span: int,
                 hotspot: bool = False, t_offset: int = 0):
    """Return ``dict(x, y, p, t)``: uniform pixels/polarity, time-sorted integer microseconds.

    ``hotspot`` puts 25 % of the events in a Gaussian blob (sigma 8 px) at the centre, the
    contention/saturation variant of SURVEY.md section 8d.
    """
    rng = np.random.default_rng(seed)
    x = rng.integers(0, width, size=n, dtype=np.int64)
    y = rng.integers(0, height, size=n, dtype=np.int64)
    p = rng.integers(0, 2, size=n, dtype=np.int64)
    t = np.sort(rng.integers(0, t_span, size=n, dtype=np.int64)) + int(t_offset)
    if hotspot and n:
        m = n // 4
        sel = rng.permutation(n)[:m]
        hx = np.rint(rng.normal(width / 2.0, 8.0, size=m)).astype(np.int64)
        hy = np.rint(rng.normal(height / 2.0, 8.0, size=m)).astype(np.int64)
        x[sel] = np.clip(hx, 0, width - 1)
        y[sel] = np.clip(hy, 0, height - 1)
    return {"x": x, "y": y, "p": p, "t": t}


def to_xytp_f64(ev, t=None) -> np.ndarray:
    """``(N, 4)`` float64 ``[x, y, t, p]``; ``t`` may be overridden (e.g. pre-normalised)."""
    tt = ev["t"] if t is None else t
    out = np.empty((len(ev["x"]), 4), dtype=np.float64)
    out[:, 0] = ev["x"]
    out[:, 1] = ev["y"]
    out[:, 2] = tt
    out[:, 3] = ev["p"]
    return out


def to_dat8(ev) -> np.ndarray:
    """Structured array of raw DAT Event2D records (8 bytes each)."""
    out = np.empty(len(ev["x"]), dtype=DAT_DTYPE)
    out["t"] = ev["t"].astype(np.uint32)
    out["_"] = (ev["x"].astype(np.uint32) & 16383) | ((ev["y"].astype(np.uint32) & 16383) << 14) \
        | ((ev["p"].astype(np.uint32) & 1) << 28)
    return out


def from_dat8(rec) -> dict:
    """Bit-unpack DAT records (``src/io/dat_events_tools.py:96-98``)."""
    w = rec["_"].astype(np.int64)
    return {"x": w & 16383, "y": (w & 268419072) >> 14, "p": (w & 268435456) >> 28,
            "t": rec["t"].astype(np.int64)}
