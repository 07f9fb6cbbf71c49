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


def _headers():
    return [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)] + \
        [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in sources() + _headers())


OBJ_DIR = os.path.join(os.path.dirname(HERE), "build", "obj")


def _compile_and_link(out, extra, tag, force, verbose):
    """One object per .hip file, compiled in parallel and only when the source or any header is newer; then one link."""
    from concurrent.futures import ThreadPoolExecutor
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + list(extra) + ["-I", INCLUDE, "-I", CSRC]
    sig = "_".join([tag] + [e.replace("-D", "").replace("=", "-").replace("/", "-") for e in extra]) or "product"
    odir = os.path.join(OBJ_DIR, sig)
    os.makedirs(odir, exist_ok=True)
    newest_h = max(os.path.getmtime(h) for h in _headers())
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(odir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_h):
            jobs.append([hipcc()] + flags + ["-c", src, "-o", obj])
    if verbose:
        for j in jobs:
            print(" ".join(j))
    with ThreadPoolExecutor(max(1, min(len(jobs), os.cpu_count() or 1))) as pool:
        list(pool.map(subprocess.check_call, jobs))
    link = [hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", out]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)


DEV_LIB = os.path.join(CSRC, "libfrlw_evd_dev.so")  # -DFRLW_DEV_BUILD: test hooks + developer logging, never the product


def build_dev(force: bool = False) -> str:
    """The developer build next to the product library (tests that need a hook the product ABI does not carry, e.g.
    frlw_debug_force_lds_order, run a child process with FRLW_LIB_PATH pointing at it)."""
    stale = not os.path.exists(DEV_LIB) or any(os.path.getmtime(d) > os.path.getmtime(DEV_LIB) for d in sources() + _headers())
    if force or stale:
        _compile_and_link(DEV_LIB, ["-DFRLW_DEV_BUILD"], "dev", force, False)
    return DEV_LIB


def build(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build() or os.environ.get("FRLW_LIB_OUT"):
        extra = os.environ.get("FRLW_EXTRA_HIPCC_FLAGS", "").split()  # experiments only (e.g. -DCONV_BK_BIG=32)
        out = os.environ.get("FRLW_LIB_OUT") or LIB  # experiments only: build a variant beside the product
        _compile_and_link(out, extra, "product", force, verbose)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="--incremental" not in sys.argv, verbose=True))
    if "--dev" in sys.argv:
        print(build_dev())
