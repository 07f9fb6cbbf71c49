#!/usr/bin/env python3
"""Sustained fp32 MFMA rate of the chip (bare v_mfma_f32_32x32x2_f32 loop): random vs all-zero operands."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frlw_evd_amd import _lib
lib = _lib.load()
def rate(seed, blocks=256 * 5, iters=4000):
    sink = torch.zeros(4, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        lib.frlw_selftest_mfma_f32_rate(blocks, iters, seed.data_ptr(), sink.data_ptr(), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        lib.frlw_selftest_mfma_f32_rate(blocks, iters, seed.data_ptr(), sink.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    return blocks * 4 * iters * 32 * 4096 / ms / 1e9, ms
for nacc in (1, 2):
    for wg in (1, 2, 5):
        seed = torch.randn(256, device="cuda")
        sink = torch.zeros(4, device="cuda"); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        b = 256 * wg
        for _ in range(2): lib.frlw_selftest_mfma_f32_rate(-(b * 10 + nacc), 4000, seed.data_ptr(), sink.data_ptr(), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): lib.frlw_selftest_mfma_f32_rate(-(b * 10 + nacc), 4000, seed.data_ptr(), sink.data_ptr(), st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"{nacc} accumulator(s), {wg} workgroups per CU: {b * 4 * 4000 * 32 * 4096 / ms / 1e9:7.1f} TFLOP/s")
for name, seed in (("random N(0,1)", torch.randn(256, device="cuda")), ("zeros", torch.zeros(256, device="cuda")),
                   ("small +-1e-3", torch.randn(256, device="cuda") * 1e-3)):
    tf, ms = rate(seed)
    print(f"{name:14s}: {tf:7.1f} TFLOP/s  ({ms:.2f} ms per launch)")
