"""Prophesee ``*_td.dat`` ingest for the fused encoders.

The on-disk Event2D record (t:u32, x | y << 14 | p << 28; src/io/dat_events_tools.py:16,96-98) is exactly
the ``FRLW_LAYOUT_DAT8`` the kernels consume, so ingest = parse the text header, memory-map the records and
copy a record range to the GPU as raw bytes; bit-unpacking, window selection and time normalisation happen
on device.  ``seek_time`` mirrors ``PSEELoader.seek_time`` (src/io/psee_loader.py:185-228): the index of the
first event with t >= final_time, by binary search over the time-sorted file.
"""
from __future__ import annotations

import numpy as np
import torch

from .synth import DAT_DTYPE


def parse_header(path):
    """-> (offset of the first record, event type, event size, (height, width)); header lines start with
    '% ', then one byte event type and one byte event size (src/io/dat_events_tools.py:118-173)."""
    size = [None, None]
    with open(path, "rb") as f:
        n_comment = 0
        while True:
            bod = f.tell()
            line = f.readline()
            if line[:2] != b"% ":
                break
            words = line.split()
            if len(words) > 2 and words[1] == b"Height":
                size[0] = int(words[2])
            if len(words) > 2 and words[1] == b"Width":
                size[1] = int(words[2])
            n_comment += 1
        f.seek(bod)
        if n_comment > 0:
            ev_type, ev_size = np.frombuffer(f.read(2), dtype=np.uint8)
            bod = f.tell()
        else:  # header-less legacy files
            ev_type, ev_size = 0, 8
    return bod, int(ev_type), int(ev_size), tuple(size)


def write_dat(path, records, height, width):
    """Write a minimal Event2D .dat file (header as src/io/dat_events_tools.py:176-199 writes it)."""
    rec = np.ascontiguousarray(records)
    assert rec.dtype.itemsize == 8
    with open(path, "wb") as f:
        f.write(b"% Data file containing Event2D events.\n% Version 2\n")
        f.write(b"% Date 2020-01-01 00:00:00\n")
        f.write(f"% Height {height}\n% Width {width}\n".encode())
        f.write(np.array([0, 8], dtype=np.uint8).tobytes())
        f.write(rec.tobytes())


class DatFile:
    """Memory-mapped view of the records of one .dat file."""

    def __init__(self, path):
        self.path = path
        self.start, self.ev_type, self.ev_size, self.size = parse_header(path)
        if self.ev_type != 0 or self.ev_size != 8:
            raise ValueError("only Event2D records (type 0, 8 bytes) are supported")
        self.records = np.memmap(path, dtype=DAT_DTYPE, mode="r", offset=self.start)

    def __len__(self):
        return len(self.records)

    def total_time(self):
        return int(self.records["t"][-1]) if len(self.records) else 0

    def seek_time(self, final_time):
        """Index of the first event with t >= final_time (len(self) past the end), like PSEELoader.seek_time."""
        if final_time <= 0:
            return 0
        if len(self.records) == 0 or final_time > self.total_time():
            return len(self.records)
        return int(np.searchsorted(self.records["t"], final_time, side="left"))

    def to_device(self, start=0, count=None, device="cuda"):
        """Raw records [start, start + count) as an (n, 8) uint8 tensor on the GPU (8 bytes per event)."""
        stop = len(self.records) if count is None else min(len(self.records), start + count)
        host = np.ascontiguousarray(self.records[start:stop]).view(np.uint8).reshape(-1, 8)
        return torch.from_numpy(host).to(device, non_blocking=True)
