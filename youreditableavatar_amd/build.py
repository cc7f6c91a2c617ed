"""Builds libtgs_raster.so (HIP kernels + C ABI) for gfx950 with plain hipcc -- no torch headers,
no hipify, no CUDAExtension (the reference's setup.py:17-34 is what this replaces).

The library is built IN-TREE (youreditableavatar_amd/lib/) so that it travels with a repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, os.environ.get("TGS_LIB_NAME", "libtgs_raster.so"))       # TGS_LIB_NAME: build a variant next to the default one
SOURCES = ["tgs_forward.hip", "tgs_backward.hip", "tgs_api.hip", "tgs_shcolor.hip", "tgs_knn.hip", "tgs_loss.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
# experiment knobs, e.g. TGS_DEFINES="-DTGS_FAST_MATH=0" python -m youreditableavatar_amd.build --force
FLAGS += os.environ.get("TGS_DEFINES", "").split()


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm; set HIPCC=/path/to/hipcc)")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "tgs_raster.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_native(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cc = hipcc()
    objs = []

    def compile_one(src):
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        cmd = [cc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
