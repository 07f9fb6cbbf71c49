#!/usr/bin/env python3
"""TFLOP/s of frlw_conv2d_fwd on the detector's main layer shapes (B = 32).   python tools/time_conv.py [B]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frlw_evd_amd import _lib  # noqa: E402

lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
PREC = int(sys.argv[2]) if len(sys.argv) > 2 else 1  # 0: float32 MFMA, 1: three bf16 MFMAs per product
SHAPES = [(32, 40, 256, 256, 3, 1), (16, 20, 256, 256, 3, 1), (8, 10, 256, 256, 3, 1), (16, 20, 128, 128, 3, 1),
          (32, 40, 64, 64, 3, 1), (128, 160, 40, 32, 3, 1), (16, 20, 256, 256, 1, 1), (32, 40, 128, 128, 1, 1),
          (64, 80, 64, 64, 1, 1), (32, 40, 256, 128, 1, 1)]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
scratch = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
tot = 0.0
for H, W, Cin, Cout, k, s in SHAPES:
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    npad = (Cout + 31) // 32 * 32
    wf = torch.empty(lib.frlw_conv_operand_floats(k * k * Cin, Cout, PREC), device="cuda")
    _lib.check(lib.frlw_conv_weight_layouts(w.data_ptr(), Cout, Cin, k, 0, wf.data_ptr(), None, PREC, st))
    Ho, Wo = H // s, W // s
    z = torch.empty(B, Ho, Wo, Cout, device="cuda")

    def run():
        _lib.check(lib.frlw_conv2d_fwd(x.data_ptr(), B, H, W, Cin, wf.data_ptr(), Cout, k, s, z.data_ptr(), scratch.data_ptr(),
                                       scratch.numel(), PREC, st))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    fl = 2.0 * B * Ho * Wo * Cout * Cin * k * k
    tot += ms
    print(f"{H:4d}x{W:<4d} {Cin:4d}->{Cout:<4d} k{k} s{s}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s")
print(f"sum {tot * 1e3:.0f} us")
