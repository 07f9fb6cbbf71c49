"""Builds libfrlw_evd.so (gfx950 HIP kernels + C-ABI) in-tree with hipcc."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(CSRC, "libfrlw_evd.so")

# -ffp-contract=off: the encoders reproduce the reference's f32 operation order bit for bit, so no
# multiply-add may be fused.  f32 divide / sqrt stay correctly rounded (hipcc's default, made explicit).
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off",
               "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)] + \
        [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        extra = os.environ.get("FRLW_EXTRA_HIPCC_FLAGS", "").split()  # experiments only (e.g. -DCONV_BK_BIG=32)
        out = os.environ.get("FRLW_LIB_OUT") or LIB  # experiments only: build a variant beside the product
        cmd = [hipcc()] + HIPCC_FLAGS + extra + ["-I", INCLUDE, "-I", CSRC] + sources() + ["-o", out]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
