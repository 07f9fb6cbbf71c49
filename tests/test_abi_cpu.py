"""The C-ABI library loads on a box without a GPU and exports every symbol include/frlw_evd.h declares
(no compute calls here)."""
import os
import re

from frlw_evd_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "frlw_evd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#ifdef FRLW_DEV_BUILD.*?#endif", "", text, flags=re.S)  # developer-build hooks: not product symbols
    return sorted(set(re.findall(r"\b(frlw_[a-z0-9_]+)\s*\(", text)))


def test_product_library_has_no_developer_hooks():
    """frlw_debug_force_lds_order (a process-wide switch) exists only in the -DFRLW_DEV_BUILD library."""
    lib = _lib.load()
    for name in _lib.DEV_SYMBOLS:
        assert not hasattr(lib, name), f"{name} is exported by the product library"


def test_tuning_struct_carries_its_size():
    t = _lib.FrlwTuning(direct_bins=1)
    assert t.struct_size == 12 * 4 and t.direct_bins == 1 and t.tile_width_log2 == -1 and t.ev_lds_float_atomics == -1 and t.walk_window_table == -1


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/frlw_evd.h but not exported"
        assert n in _lib.SYMBOLS, f"{n} has no ctypes prototype in _lib.SYMBOLS"
    assert b"gfx950" in lib.frlw_version()


def test_workspace_query_is_host_only():
    lib = _lib.load()
    assert lib.frlw_encoder_workspace_bytes(10_000_000, 720, 1280) > 80_000_000
    assert lib.frlw_encoder_workspace_bytes(0, 8, 12) > 0
    assert lib.frlw_encoder_workspace_bytes(10, 0, 12) == 0


def test_batch_workspace_queries_are_host_only():
    """frlw_taf_batch_workspace_bytes / frlw_ev_batch_workspace_bytes: host arithmetic only; the answer covers BOTH partition
    modes (tile bins, and sub-tile bins where the frame allows them: 16 x the counters per chunk), grows with the events and
    is 0 for shapes the fast path refuses."""
    lib = _lib.load()
    mpx = lib.frlw_taf_batch_workspace_bytes(10_000_000, 1, 720, 1280, 10_000)
    assert mpx > 2 * 4 * 10_000_000                      # two 4-byte record arrays at least
    gen1 = lib.frlw_taf_batch_workspace_bytes(1_000_000, 1, 240, 304, 10_000)
    gen1x8 = lib.frlw_taf_batch_workspace_bytes(8_000_000, 8, 240, 304, 10_000)
    assert 8_000_000 < gen1 < gen1x8
    # 304x240 = 36 tiles = 576 sub-tile bins per sequence: the counters of the direct mode (576 per chunk) are budgeted
    chunks = 1_000_000 // 8192
    assert gen1 > 2 * 4 * 1_000_000 + chunks * 576 * 4
    assert lib.frlw_ev_batch_workspace_bytes(1_000_000, 1, 240, 304, 250_000) == lib.frlw_taf_batch_workspace_bytes(1_000_000, 1, 240, 304, 250_000)
    assert lib.frlw_taf_batch_workspace_bytes(1_000, 65, 240, 304, 10_000) == 0        # more than 64 sequences
    assert lib.frlw_taf_batch_workspace_bytes(1_000, 1, 240, 304, 1 << 21) == 0        # window beyond 2^20 us
    assert lib.frlw_taf_batch_workspace_bytes(1_000, 1, 4000, 4000, 10_000) == 0       # more tiles than the scatter's LDS holds


def test_operand_size_queries_and_precision_arguments_are_host_only():
    """Sizes of the GEMM operands in both arithmetics (float32: K rows; split bf16 image: ceil16(K) rows -- the same bytes per
    row) and argument checks of the precision switches, without a GPU."""
    lib = _lib.load()
    assert lib.frlw_conv_operand_floats(360, 32, 0) == 360 * 32
    assert lib.frlw_conv_operand_floats(360, 32, 1) == 368 * 32
    assert lib.frlw_conv_operand_floats(2304, 200, 1) == 2304 * 224
    assert lib.frlw_conv_split_operand_bytes(360, 32) == 368 * 32 * 4
    assert lib.frlw_baseconv_weight_cache_floats(64, 128, 3, 0) == 9 * 64 * 128 + 9 * 128 * 64
    assert lib.frlw_baseconv_weight_cache_floats(40, 32, 3, 1) == 368 * 32 + 288 * 64
    assert lib.frlw_det_set_precision(None, 1) == _lib.FRLW_ERR_ARG
    d = lib.frlw_det_create()
    try:
        assert lib.frlw_det_set_precision(d, 1) == 0 and lib.frlw_det_set_precision(d, 0) == 0
        assert lib.frlw_det_set_precision(d, 2) == _lib.FRLW_ERR_ARG
    finally:
        lib.frlw_det_destroy(d)


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "frlw-evd_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("test infrastructure", ""), f"{f} mentions the oracle"


def test_round6_queries_are_host_only():
    """The size query of the NMS workspace and the argument checks of the round-6 entry points run without a GPU."""
    import ctypes as C
    lib = _lib.load()
    # per image: 4 + 256 header floats, 4 floats per (padded) anchor, 2 floats per (row, 64-column word) of the bit matrix
    assert lib.frlw_det_nms_workspace_floats(1680) == 260 + 4 * 1728 + 2 * 27 * 1728
    assert lib.frlw_det_nms_workspace_floats(6720) == 260 + 4 * 6720 + 2 * 105 * 6720
    assert lib.frlw_det_nms_workspace_floats(100_000) == 260 + 4 * 8192 + 2 * 128 * 8192   # capped at the 8192 candidates the device NMS holds
    assert lib.frlw_det_nms_workspace_floats(0) == 0
    counts = (C.c_uint64 * 4)()
    assert lib.frlw_encoder_path_counts(counts) == 0 and list(counts) == [0, 0, 0, 0]
    assert lib.frlw_encoder_path_counts(None) == _lib.FRLW_ERR_ARG
    ok = C.c_int(7)
    assert lib.frlw_fast_path_verdict(None, 0, None, C.byref(ok)) == _lib.FRLW_ERR_ARG and ok.value == 7
    # the two-launch forms of the single-stream encoders are covered by the documented size query
    small = lib.frlw_encoder_workspace_bytes(1_000_000, 240, 304)
    assert small >= 8 * 1_000_000 + 372_000   # records + the chunk-major tables (576 bins x 145 chunks and the rest)


def test_baseconv_fuse_struct_is_checked_on_the_host():
    """frlw_baseconv_fuse_t carries its size; a wrong one is refused before anything is launched (the next check, the scratch
    size, answers for a well-formed struct: no GPU needed for either)."""
    import ctypes as C
    lib = _lib.load()
    assert C.sizeof(_lib.FrlwBaseconvFuse) == 136
    p = C.c_void_p(0x1000)  # never dereferenced: both answers come from host-side argument checks
    good = _lib.FrlwBaseconvFuse(residual=0x2000, y_row_stride=64)
    bad = _lib.FrlwBaseconvFuse()
    bad.struct_size = 40

    def fwd(fuse):
        return lib.frlw_baseconv_train_fwd(p, p, p, p, C.c_float(1e-5), 1, 8, 8, 16, 16, 1, 1, p, p, p, p, p, None, None, C.c_float(0.1),
                                           None, None, p, 0, None, C.byref(fuse), 0, None)

    def bwd(fuse):
        return lib.frlw_baseconv_train_bwd(p, 0, p, p, p, p, p, p, p, 1, 8, 8, 16, 16, 1, 1, p, p, p, p, p, None, p, 0, None,
                                           C.byref(fuse), 0, None)
    assert fwd(bad) == _lib.FRLW_ERR_ARG and bwd(bad) == _lib.FRLW_ERR_ARG
    assert fwd(good) == _lib.FRLW_ERR_WORKSPACE and bwd(good) == _lib.FRLW_ERR_WORKSPACE
