"""The C-ABI library loads on a box without a GPU and exports every symbol include/frlw_evd.h declares
(no compute calls here)."""
import os
import re

from frlw_evd_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "frlw_evd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(frlw_[a-z0-9_]+)\s*\(", text)))


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


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "frlw-evd_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("test infrastructure", ""), f"{f} mentions the oracle"
