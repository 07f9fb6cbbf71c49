"""The four offline pre-processing harnesses of the reference -- the ``__main__`` blocks of ``generate_eventcountimage.py``,
``generate_eventvolume.py``, ``generate_surfaceofactiveevents.py`` and ``generate_taf.py`` -- over the gfx950 encoders:
same command line (``-raw_dir -label_dir -target_dir -dataset``), same walk over ``train / val / test``, same label slicing
(``dat_io.*_label_slices``, pinned to the reference's own ``PSEELoader`` calls), same output tree and file names, and the
same bytes in every file (Event Count Image / Event Volume bit for bit; SAE / TAF within 1 LSB in <= 1e-4 of the bytes --
``exp`` / ``log1p`` come from different math libraries).  Pinned end to end by tests/golden/harness.npz, which the reference's
scripts wrote on a fabricated dataset (tests/golden/make_golden_harness.py).

Each ``*_td.dat`` file goes to HBM ONCE as raw 8-byte records (``DatFile.to_device``); every label is a record range of that
tensor: bit-unpacking, window selection, f64 time normalisation, the coordinate down-scale, encode, nearest resize and uint8
quantisation all run on the device, only the finished uint8 file comes back.
"""
from __future__ import annotations

import argparse
import os

import numpy as np
import torch

from . import dat_io
from . import event_representation as er

SHAPES = {"gen4": ((720, 1280), (512, 640)), "gen1": ((240, 304), (256, 320))}  # sensor, detector input (all four scripts)


def _parser(default_dataset):
    p = argparse.ArgumentParser(description="visualize one or several event files along with their boxes")
    p.add_argument("-raw_dir", type=str)      # "train, val, test" level directory of the datasets, data source
    p.add_argument("-label_dir", type=str)    # "train, val, test" level directory of the datasets, annotations
    p.add_argument("-target_dir", type=str)   # output directory
    p.add_argument("-dataset", type=str, default=default_dataset)
    return p


def read_label_times(bbox_file):
    """``np.unique(dat_bbox['t'])`` of a ``*_bbox.npy`` file (generate_taf.py:146-151): an .npy of a structured array whose
    timestamp field is ``t`` (``ts`` in older files, src/io/npy_events_tools.py:57)."""
    boxes = np.load(bbox_file)
    name = "t" if "t" in boxes.dtype.names else "ts"
    return np.unique(boxes[name])


def _sequences(raw_dir, label_dir):
    """(mode, sequence name, event file, bbox file) in the order the reference walks them (generate_taf.py:112-146)."""
    for mode in ("train", "val", "test"):
        file_dir = os.path.join(raw_dir, mode)
        try:
            files = os.listdir(file_dir)
        except Exception:
            continue
        for name in [f[:-7] for f in files if f[-3:] == "dat"]:
            yield mode, name, os.path.join(file_dir, name + "_td.dat"), os.path.join(label_dir, mode, name + "_bbox.npy")


def _geometry(dataset, device):
    shape, target = SHAPES["gen4" if dataset == "gen4" else "gen1"]
    if target[0] < shape[0]:   # down-scale: the events are moved, the encode runs at the target shape (generate_taf.py:216-219)
        xmap, ymap = er.coordinate_maps(shape, target, device)
        return shape, target, target, xmap, ymap
    return shape, target, shape, None, None   # encode natively, then nearest-resize the volume up (:221-222)


def _write(u8, enc_shape, target, path):
    """uint8 volume (C, h, w) at the encode shape -> nearest resize to the target shape (a no-op when equal) -> file."""
    if tuple(enc_shape) != tuple(target):
        u8 = er.resize_nearest(u8, target)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    u8.cpu().numpy().tofile(path)


# ---- generate_eventcountimage.py:66-185 ---------------------------------------------------------------------------------
def generate_eventcountimage(raw_dir, label_dir, target_dir, dataset="gen4", device="cuda"):
    events_windows = [400000, 800000, 1200000] if dataset == "gen4" else [50000, 100000, 200000]
    _, target, enc, xmap, ymap = _geometry(dataset, device)
    os.makedirs(target_dir, exist_ok=True)
    n_files = 0
    for mode, name, event_file, bbox_file in _sequences(raw_dir, label_dir):
        f_event = dat_io.DatFile(event_file)
        dat = f_event.to_device(device=device)
        for sl in dat_io.eci_label_slices(f_event, read_label_times(bbox_file), events_windows):
            for n in events_windows:
                lo = max(sl["tail_start"], sl["end_count"] - n)
                _, u8 = er.encode_eci_dat(dat[lo:sl["end_count"]], enc, want_f32=False, want_u8=True, xmap=xmap, ymap=ymap)
                _write(u8, enc, target, os.path.join(target_dir, f"EventCountImage{n}", mode, f"{name}_{sl['label_time']}.npy"))
                n_files += 1
    return n_files


# ---- generate_eventvolume.py:63-172 -------------------------------------------------------------------------------------
def generate_eventvolume(raw_dir, label_dir, target_dir, dataset="gen1", device="cuda"):
    time_windows = [250000, 500000, 1000000]
    bins = 5
    _, target, enc, xmap, ymap = _geometry(dataset, device)
    os.makedirs(target_dir, exist_ok=True)
    n_files = 0
    for mode, name, event_file, bbox_file in _sequences(raw_dir, label_dir):
        f_event = dat_io.DatFile(event_file)
        dat = f_event.to_device(device=device)
        for sl in dat_io.ev_label_slices(f_event, read_label_times(bbox_file), time_windows):
            rec = dat[sl["start_count"]:sl["end_count"]]
            for tw in time_windows:
                # events with t > end_time - tw, t* = (t - (end_time - tw)) / tw (:139-141); > 255 -> 255, uint8 (:155-157)
                _, u8 = er.encode_ev_dat(rec, enc, sl["end_time"], tw, bins, want_f32=False, want_u8=True, xmap=xmap, ymap=ymap)
                _write(u8, enc, target, os.path.join(target_dir, f"EventVolume{tw}", mode, f"{name}_{sl['label_time']}.npy"))
                n_files += 1
    return n_files


# ---- generate_surfaceofactiveevents.py:82-215 ---------------------------------------------------------------------------
def generate_surfaceofactiveevents(raw_dir, label_dir, target_dir, dataset="gen1", device="cuda"):
    lamdas = [0.00001, 0.0000025, 0.000001]
    time_window = [554126, 2216505, 5541263]
    _, target, enc, xmap, ymap = _geometry(dataset, device)
    os.makedirs(target_dir, exist_ok=True)
    n_files = 0
    for mode, name, event_file, bbox_file in _sequences(raw_dir, label_dir):
        f_event = dat_io.DatFile(event_file)
        dat = f_event.to_device(device=device)
        memory = None
        for sl in dat_io.sae_label_slices(f_event, read_label_times(bbox_file)):
            rec = dat[sl["start_count"]:sl["end_count"]]
            # the test split runs all three windows, each on the memory the previous one left (:178-194); the file holds the
            # volume of the largest window (:199-200)
            u8 = None
            for tw in (time_window if mode == "test" else [max(time_window)]):
                _, u8, memory = er.encode_sae_dat(rec, enc, lamdas, memory, sl["label_time"], tw, want_f32=False, want_u8=True,
                                                  xmap=xmap, ymap=ymap)
            vol = u8.view(len(lamdas), 2, enc[0], enc[1])
            for j, lam in enumerate(lamdas):
                _write(vol[j], enc, target, os.path.join(target_dir, f"SurfaceOfActiveEvents{lam}", mode,
                                                         f"{name}_{sl['label_time']}.npy"))
                n_files += 1
    return n_files


# ---- generate_taf.py:78-243 ---------------------------------------------------------------------------------------------
def generate_taf(raw_dir, label_dir, target_dir, dataset="gen4", device="cuda"):
    events_window_abin, K, min_event_count = 10000, 8, 50000000
    _, target, enc, xmap, ymap = _geometry(dataset, device)
    target_dir = os.path.join(target_dir, "taf")
    os.makedirs(target_dir, exist_ok=True)
    n_files = 0
    for mode, name, event_file, bbox_file in _sequences(raw_dir, label_dir):
        os.makedirs(os.path.join(target_dir, mode), exist_ok=True)
        f_event = dat_io.DatFile(event_file)
        dat = f_event.to_device(device=device)
        state, stale = None, 0
        for sl in dat_io.taf_label_slices(f_event, read_label_times(bbox_file), events_window_abin, K, min_event_count):
            if sl["fresh"]:
                state = torch.full((enc[0], enc[1], 2, K), -6000.0, device=device)   # :205-209
            # a label that rounds onto the previous one has no window (:181): the reference's `volume` is then STALE -- the
            # previous label's already transformed volume goes through leaky_transform again (:226-227); `stale` counts how
            # many such labels precede this one in a row
            stale = stale + 1 if sl["bins"] <= 0 else 0
            u8 = er.encode_taf_label(dat[sl["start_count"]:sl["end_count"]], enc, state, sl["start_time"], events_window_abin,
                                     sl["bins"], K, flip_k=True, xmap=xmap, ymap=ymap, stale_transforms=max(stale - 1, 0))
            if tuple(enc) != tuple(target):
                u8 = er.resize_nearest(u8.reshape(2 * K, enc[0], enc[1]), target).reshape(K, 2, target[0], target[1])
            host = u8.cpu().numpy()
            for part, sub in ((host[:K // 2], f"bins{K // 2}"), (host[K // 2:], f"bins{K}")):   # :229-235
                path = os.path.join(target_dir, mode, sub, f"{name}_{sl['label_time']}.npy")
                os.makedirs(os.path.dirname(path), exist_ok=True)
                part.tofile(path)
                n_files += 1
    return n_files


HARNESSES = {
    "eventcountimage": (generate_eventcountimage, "gen4"),   # the scripts' own -dataset defaults
    "eventvolume": (generate_eventvolume, "gen1"),
    "surfaceofactiveevents": (generate_surfaceofactiveevents, "gen1"),
    "taf": (generate_taf, "gen4"),
}


def main(which, argv=None):
    fn, default_dataset = HARNESSES[which]
    args = _parser(default_dataset).parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("the encoders run on the GPU: no ROCm device is visible (there is no CPU fallback)")
    n = fn(args.raw_dir, args.label_dir, args.target_dir, args.dataset)
    print(f"{which}: {n} files written under {args.target_dir}")


if __name__ == "__main__":
    import sys
    main(sys.argv[1], sys.argv[2:])
