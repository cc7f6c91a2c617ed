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
SOURCES = ["tgs_forward.hip", "tgs_backward.hip", "tgs_api.hip", "tgs_shcolor.hip", "tgs_knn.hip", "tgs_loss.hip", "tgs_bind.hip"]
ARCH = "gfx950"
# -fno-slp-vectorize: the SLP vectoriser pairs scalar f32 operations into v_pk_fma_f32 / v_pk_mul_f32, which on gfx950 issue no faster than
# the two scalar instructions (MI355X_MICROARCH.md: packed f32 VALU is an anti-lever) while the pairing costs registers and moves:
# k_preprocess_bwd_batch_split 166 -> 127 VGPRs (3 -> 4 waves per SIMD), k_preprocess_bwd<false,true> 108 -> 72, the streaming SSIM kernels
# 138 -> 86 / 91 -> 68.  Measured on the MI355X (round 3, A/B by library): every kernel of the frame 3-7 % faster, the 8-view step
# 2.04 -> 1.97 ms, the drop-in frame 0.386 -> 0.376 ms.
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]
# Per-source flags (none at the moment; the hook stays for A/B builds of one file)
EXTRA_FLAGS = {}
# experiment knobs, e.g. TGS_LIB_NAME=libtgs_raster_x.so TGS_DEFINES="-DTGS_STAMPS=1" python -m youreditableavatar_amd.build --force.  They apply
# to a NAMED variant only: TGS_DEFINES left in the environment of an ordinary import must not change what the default library is built from
# (or make a fresh one look stale).
if os.environ.get("TGS_LIB_NAME"):
    FLAGS += os.environ.get("TGS_DEFINES", "").split()


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm; set HIPCC=/path/to/hipcc)")


HASH_FILE = os.path.join(LIBDIR, ".source_hash")            # the hash of the sources the library next to it was built from


def _sources_present() -> bool:
    return os.path.isdir(CSRC) and os.path.exists(os.path.join(HERE, "..", "include", "tgs_raster.h"))


def source_hash() -> str:
    """sha256 (first 16 hex digits) over the kernel / ABI sources and the compile flags: profiles/*_counters.json carry it, and bench.py only
    quotes counter figures (instructions, HBM traffic per launch) whose hash equals the sources it is running; the build stores it next to
    the library, and a library whose stored hash differs from the sources is STALE whatever its mtime says (a copy to another machine keeps
    no meaningful mtimes)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".cpp"))) 
    for f in files:
        h.update(f.encode()); h.update(open(os.path.join(CSRC, f), "rb").read())
    for hname in ("tgs_raster.h", "tgs_raster_testing.h"):
        hp = os.path.join(HERE, "..", "include", hname)
        if os.path.exists(hp):
            h.update(open(hp, "rb").read())
    return h.hexdigest()[:16]


def _stored_hash(path: str):
    try:
        return open(path).read().split()
    except OSError:
        return []


def _hash_line() -> list:
    return [source_hash(), " ".join(FLAGS).replace(" ", "|")]


EXT_CXX_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden"]


def _ext_hash_line() -> list:
    """what the compiled `_C` glue depends on: the sources, ITS compiler flags (g++, not hipcc's), and the torch / Python ABI it was built
    against -- after a torch upgrade in the same tree the old module must not be loaded"""
    import sysconfig
    try:
        import torch
        tv = torch.__version__
    except Exception:                                    # noqa: BLE001
        tv = "no-torch"
    return [source_hash(), "|".join(EXT_CXX_FLAGS), f"torch={tv}", f"abi={sysconfig.get_config_var('SOABI')}"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    if not _sources_present():                   # a binary-only install (wheel): nothing to compare with, nothing to rebuild from
        return False
    if os.path.basename(LIB) != "libtgs_raster.so":
        return False                             # a named variant (TGS_LIB_NAME) is rebuilt on request only
    return _stored_hash(HASH_FILE) != _hash_line()


class _BuildLock:
    """Serialises builds between processes (the ranks of one launch import the package at the same moment): flock on a file next to the
    outputs; whoever gets it second finds everything fresh."""

    def __enter__(self):
        import fcntl
        os.makedirs(LIBDIR, exist_ok=True)
        self.f = open(os.path.join(LIBDIR, ".build.lock"), "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()


def build_native(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    with _BuildLock():
        if not force and not _stale():
            return LIB
        return _build_native_locked(verbose)


def _build_native_locked(verbose: bool) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    cc = hipcc()
    objs = []

    def compile_one(src):
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        cmd = [cc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=7) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    tmp = LIB + f".tmp{os.getpid()}"
    cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", tmp, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    os.replace(tmp, LIB)                                    # atomic: a process that is loading the old file keeps its mapping
    if os.path.basename(LIB) == "libtgs_raster.so":
        with open(HASH_FILE, "w") as f:
            f.write(" ".join(_hash_line()) + "\n")
    return LIB


EXT_SRC = os.path.join(CSRC, "tgs_torch_ext.cpp")
EXT_NAME = "_Cext"
EXT_DIR = os.path.join(HERE, "diff_gaussian_rasterization")
EXT_HASH_FILE = os.path.join(EXT_DIR, ".source_hash")


def ext_path() -> str:
    import sysconfig
    return os.path.join(EXT_DIR, EXT_NAME + sysconfig.get_config_var("EXT_SUFFIX"))


def build_torch_ext(force: bool = False, verbose: bool = False) -> str:
    """The compiled `_C` glue (csrc/tgs_torch_ext.cpp: torch::Tensor <-> C ABI, pybind11) -- plain g++ against the torch and pybind11
    headers, linked to libtgs_raster.so (rpath $ORIGIN/../lib) and the torch libraries; host code only.  Replaces the reference's
    CUDAExtension recipe (setup.py:17-34) without hipify."""
    lib = build_native(force=force, verbose=verbose)
    out = ext_path()
    fresh = lambda: os.path.exists(out) and (not _sources_present() or _stored_hash(EXT_HASH_FILE) == _ext_hash_line())
    if not force and fresh():
        return out
    with _BuildLock():
        if not force and fresh():
            return out
        return _build_torch_ext_locked(lib, out, verbose)


def _build_torch_ext_locked(lib: str, out: str, verbose: bool) -> str:
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    cxx = os.environ.get("CXX") or shutil.which("g++") or "g++"
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    inc = [f"-I{p}" for p in ce.include_paths()] + [f"-I{sysconfig.get_paths()['include']}", f"-I{rocm}/include"]
    try:
        import pybind11
        inc.append(f"-I{pybind11.get_include()}")
    except ImportError:
        pass                                             # torch ships pybind11 headers as well
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [cxx, *EXT_CXX_FLAGS, "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", f"-DTORCH_EXTENSION_NAME={EXT_NAME}", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-Wno-deprecated-declarations", *inc, EXT_SRC, "-o", out + f".tmp{os.getpid()}", f"-L{LIBDIR}", f"-l:{os.path.basename(lib)}", f"-L{tlib}", "-ltorch_python", "-ltorch",
           "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip", "-Wl,-rpath,$ORIGIN/../lib", f"-Wl,-rpath,{tlib}"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"building {EXT_NAME} failed:\n{r.stdout}\n{r.stderr}")
    os.replace(out + f".tmp{os.getpid()}", out)
    with open(EXT_HASH_FILE, "w") as f:
        f.write(" ".join(_ext_hash_line()) + "\n")
    return out


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
    if "TGS_LIB_NAME" not in os.environ:                 # (a variant library is loaded through TGS_LIBRARY; the glue stays linked to the default one)
        print(build_torch_ext(force="--force" in sys.argv, verbose=True))
