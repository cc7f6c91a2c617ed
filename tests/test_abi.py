"""CPU: the C-ABI library loads and exports every symbol include/*.h declares (tgs_raster.h: the drop-in boundary; tgs_raster_testing.h:
the test-only shims); the product never links, imports or falls back to anything under oracle/."""
import ctypes
import os
import re
import subprocess

import pytest

from tests.util import ROOT

HEADER = os.path.join(ROOT, "include", "tgs_raster.h")
TESTING_HEADER = os.path.join(ROOT, "include", "tgs_raster_testing.h")


def declared_functions(header=HEADER):
    src = open(header).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tgs_[a-z0-9_]+)\s*\(", src)) - {"tgs_alloc_fn"})


def lib_path():
    from youreditableavatar_amd import build
    return build.build_native()


def test_header_declares_the_boundary():
    fns = declared_functions()
    for f in ("tgs_forward", "tgs_backward", "tgs_mark_visible", "tgs_last_error", "tgs_abi_version", "tgs_state_field"):
        assert f in fns
    text = open(HEADER).read()
    # every entry point cites the reference interface it replaces
    for cite in ("rasterizer.h:33-58", "rasterizer.h:60-85", "rasterizer.h:24-31", "rasterize_points.cu"):
        assert cite in text
    assert "torch" not in text.lower().replace("torch::zeros", "")   # no torch types in the signatures
    # the process- / thread-wide setters of rounds 1-2 are not part of the boundary: test-only header
    shims = declared_functions(TESTING_HEADER)
    assert set(shims) == {"tgs_set_instance_pruning", "tgs_set_render_streams", "tgs_set_tile_bound", "tgs_last_nonempty_tiles", "tgs_set_forward_group",
                          "tgs_set_sort_lds_cap", "tgs_set_deterministic"}
    assert not (set(shims) & set(fns))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(lib_path())
    for f in declared_functions() + declared_functions(TESTING_HEADER):
        assert hasattr(lib, f), f"libtgs_raster.so does not export {f}"
    lib.tgs_abi_version.restype = ctypes.c_int
    header_version = int(re.search(r"#define TGS_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert lib.tgs_abi_version() == header_version == 3


def test_bindings_check_abi_version_and_struct_sizes():
    """A stale libtgs_raster.so must not be walked with a newer tgs_view_t: the ctypes binding and the compiled module
    both compare the ABI version and sizeof(tgs_view_t) / sizeof(tgs_options_t) with the library's at import."""
    import ctypes as C
    from diff_gaussian_rasterization import _C
    assert _C.ABI_VERSION == 3 and _C._ext.abi_version() == 3 and _C._ext.compiled_abi_version() == 3
    assert _C._lib.tgs_sizeof_view() == C.sizeof(_C._ViewT) == _C._ext.sizeof_view()
    assert _C._lib.tgs_sizeof_options() == C.sizeof(_C._OptionsT)
    o = _C.options(tile_bound=640, pruning=False, deterministic=True, sort_lds_cap=512)
    assert (o.struct_size, o.tile_bound, o.instance_pruning, o.deterministic, o.sort_lds_cap, o.light_tiles) == (C.sizeof(_C._OptionsT), 640, 0, 1, 512, -1)
    src = open(os.path.join(ROOT, "youreditableavatar_amd", "diff_gaussian_rasterization", "_C.py")).read()
    assert "could not be rebuilt" in src and "older than its sources" in src        # a failed rebuild of a stale library raises, it is not swallowed


def test_library_is_independent_of_torch_and_oracle():
    out = subprocess.run(["ldd", lib_path()], capture_output=True, text=True).stdout
    assert "libtorch" not in out and "libc10" not in out, "the C ABI must not depend on torch"
    assert "tgs_oracle" not in out
    syms = subprocess.run(["nm", "-D", lib_path()], capture_output=True, text=True).stdout
    assert "tgs_oracle" not in syms


def test_product_package_never_touches_the_oracle():
    """No import, link or execution of anything under oracle/ from the product package."""
    pkg = os.path.join(ROOT, "youreditableavatar_amd")
    for dp, _dn, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dp, fn), errors="ignore").read()
                for bad in ("import oracle", "from oracle", "tgs_oracle", "libtgs_oracle", "torch_splat"):
                    assert bad not in text, f"{os.path.join(dp, fn)} references the oracle ({bad})"


def test_invalid_arguments_are_rejected_without_a_gpu():
    """Argument validation happens before any HIP call, so it is checkable on a CPU-only box."""
    lib = ctypes.CDLL(lib_path())
    lib.tgs_forward.restype = ctypes.c_int64
    lib.tgs_last_error.restype = ctypes.c_char_p
    vp = ctypes.c_void_p
    args = [vp(0), vp(0), vp(0), ctypes.c_int(1), ctypes.c_int(0), ctypes.c_int(0), vp(0), ctypes.c_int(16), ctypes.c_int(16)] + \
           [vp(0)] * 5 + [ctypes.c_float(1.0)] + [vp(0)] * 5 + [ctypes.c_float(1.0), ctypes.c_float(1.0), ctypes.c_int(0), vp(0), vp(0), ctypes.c_int(0)]
    r = lib.tgs_forward(*args)
    assert r == -1 and b"alloc" in lib.tgs_last_error()


def test_argument_validation_needs_no_gpu():
    """Every entry point rejects bad arguments with TGS_ERR_INVALID and a message before it touches the device."""
    lib = ctypes.CDLL(lib_path())
    vp, it, i64, fl, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t
    lib.tgs_last_error.restype = ctypes.c_char_p
    INVALID = -1

    def msg():
        return (lib.tgs_last_error() or b"").decode()

    lib.tgs_forward.restype = i64
    lib.tgs_forward.argtypes = [vp, vp, vp, it, it, it, vp, it, it, vp, vp, vp, vp, vp, fl, vp, vp, vp, vp, vp, fl, fl, it, vp, vp, it]
    assert lib.tgs_forward(None, None, None, 10, 0, 0, None, 64, 64, None, None, None, None, None, 1.0, None, None, None, None, None, 1.0, 1.0, 0, None, None, 0) == INVALID
    assert "alloc" in msg()
    lib.tgs_forward_async.restype = i64
    lib.tgs_forward_async.argtypes = [i64] + lib.tgs_forward.argtypes
    assert lib.tgs_forward_async(-5, None, None, None, 10, 0, 0, None, 64, 64, None, None, None, None, None, 1.0, None, None, None, None, None, 1.0, 1.0, 0, None, None, 0) == INVALID
    lib.tgs_backward_render.restype = it
    lib.tgs_backward_render.argtypes = [vp, it, i64, vp, it, it, vp, vp, vp]
    assert lib.tgs_backward_render(None, 10, 5, None, 64, 64, None, None, None) == INVALID and "NULL" in msg()
    assert lib.tgs_backward_render(None, 0, 0, None, 64, 64, None, None, None) == 0          # P == 0: nothing to do
    lib.tgs_backward_batch.restype = it
    lib.tgs_backward_batch.argtypes = [vp, it, it, it, it, vp, vp, vp, vp, fl, vp, vp, vp, vp, vp, vp, vp, vp, it]
    assert lib.tgs_backward_batch(None, 10, 3, 16, 2, None, None, None, None, 1.0, None, None, None, None, None, None, None, None, 0) == INVALID
    # level-major dL_dsh (round 6): M = 16, a plane stride >= 3 P that is a multiple of 4, a 16-byte aligned pointer
    lib.tgs_backward_batch_range_planes.restype = it
    lib.tgs_backward_batch_range_planes.argtypes = lib.tgs_backward_batch.argtypes + [it, it, i64]
    some = ctypes.c_void_p(4096)
    planes = lambda M, stride, dsh: lib.tgs_backward_batch_range_planes(None, 1000, 2, M, 1, some, some, some, some, 1.0, some, None, some, some, None, dsh, some, some, 0, 0, 1000, stride)
    assert planes(9, 3008, some) == INVALID and "level-major" in msg()
    assert planes(16, 2999, some) == INVALID and planes(16, 3002, some) == INVALID and planes(16, 3008, ctypes.c_void_p(4100)) == INVALID
    lib.tgs_forward_views.restype = it
    lib.tgs_forward_views.argtypes = [vp, it, i64, it, it, it, vp, vp, vp, vp, vp, fl, vp, vp, it, it, vp]
    assert lib.tgs_forward_views(None, 0, 100, 10, 3, 16, None, None, None, None, None, 1.0, None, None, 0, 2, None) == INVALID
    # the explicit-options entry points validate the same way (NULL options = defaults)
    lib.tgs_forward_opt.restype = i64
    lib.tgs_forward_opt.argtypes = [vp, it, i64, vp] + lib.tgs_forward.argtypes
    fwd_args = (None, None, None, 10, 0, 0, None, 64, 64, None, None, None, None, None, 1.0, None, None, None, None, None, 1.0, 1.0, 0, None, None, 0)
    assert lib.tgs_forward_opt(None, 0, 0, None, *fwd_args) == INVALID and "alloc" in msg()
    assert lib.tgs_forward_opt(None, 7, 0, None, *fwd_args) == INVALID and "mode" in msg()
    assert lib.tgs_forward_opt(None, 1, -3, None, *fwd_args) == INVALID
    lib.tgs_backward_render_opt.restype = it
    lib.tgs_backward_render_opt.argtypes = [vp] + lib.tgs_backward_render.argtypes
    assert lib.tgs_backward_render_opt(None, None, 10, 5, None, 64, 64, None, None, None) == INVALID and "NULL" in msg()
    lib.tgs_forward_views_opt.restype = it
    lib.tgs_forward_views_opt.argtypes = [vp] + lib.tgs_forward_views.argtypes
    assert lib.tgs_forward_views_opt(None, None, 0, 100, 10, 3, 16, None, None, None, None, None, 1.0, None, None, 0, 2, None) == INVALID
    lib.tgs_sizeof_view.restype = sz
    lib.tgs_sizeof_options.restype = sz
    assert lib.tgs_sizeof_view() == 192 and lib.tgs_sizeof_options() == 56
    lib.tgs_dist2.restype = it
    lib.tgs_dist2.argtypes = [vp, it, vp, vp, vp, sz]
    assert lib.tgs_dist2(None, -1, None, None, None, 0) == INVALID and "tgs_dist2" in msg()
    assert lib.tgs_dist2(None, 0, None, None, None, 0) == 0
    lib.tgs_knn_self.restype = it
    lib.tgs_knn_self.argtypes = [vp, it, it, vp, vp, vp, vp, sz]
    assert lib.tgs_knn_self(None, 10, 33, None, None, None, None, 0) == INVALID and "tgs_knn_self" in msg()
    lib.tgs_l1_ssim.restype = it
    lib.tgs_l1_ssim.argtypes = [vp, it, it, it, vp, vp, fl, vp, vp, vp, sz]
    assert lib.tgs_l1_ssim(None, 0, 8, 8, None, None, 0.2, None, None, None, 0) == INVALID and "tgs_l1_ssim" in msg()
    lib.tgs_sh_rgb_forward.restype = it
    lib.tgs_sh_rgb_forward.argtypes = [vp, it, it, it, vp, vp, vp, vp, vp]
    assert lib.tgs_sh_rgb_forward(None, 10, 16, 5, None, None, None, None, None) == INVALID and "levels" in msg()
    lib.tgs_set_sort_lds_cap.restype = it
    lib.tgs_set_sort_lds_cap.argtypes = [ctypes.c_uint]
    assert lib.tgs_set_sort_lds_cap(3) < 0 and lib.tgs_set_sort_lds_cap(8192) == 0
    # host-only size queries
    lib.tgs_state_sizes.restype = None
    lib.tgs_state_sizes.argtypes = [it, it, it, it, it, i64, ctypes.POINTER(sz)]
    a, b = (sz * 3)(), (sz * 3)()
    lib.tgs_state_sizes(500000, 1920, 1080, 1, 1, 1_000_000, a)
    lib.tgs_state_sizes(500000, 1920, 1080, 1, 1, 2_000_000, b)
    assert a[0] == b[0] > 500000 * 64 and a[2] == b[2] > 1920 * 1080 * 8 and b[1] > a[1] >= 1_000_000 * 100
    lib.tgs_dist2_workspace_bytes.restype = sz
    lib.tgs_dist2_workspace_bytes.argtypes = [it]
    assert lib.tgs_dist2_workspace_bytes(1000) > 1000 * 24
    lib.tgs_l1_ssim_workspace_bytes.restype = sz
    lib.tgs_l1_ssim_workspace_bytes.argtypes = [it, it, it]
    assert lib.tgs_l1_ssim_workspace_bytes(3, 1080, 1920) >= 3 * 3 * 1080 * 1920 * 4 and lib.tgs_l1_ssim_workspace_bytes(0, 8, 8) == 0


def test_compiled_extension_module_exports_the_reference_names():
    """The reference's `_C` is a compiled pybind module with three functions (ext.cpp:15-19); so is ours: built by plain g++ from
    csrc/tgs_torch_ext.cpp, linked against libtgs_raster.so (the C ABI does the work) and torch, and what `_C` hands out."""
    import inspect
    from youreditableavatar_amd import build
    path = build.build_torch_ext()
    assert os.path.basename(path).startswith("_Cext") and path.endswith(".so")
    out = subprocess.run(["ldd", path], capture_output=True, text=True).stdout
    assert "libtgs_raster.so" in out and "libtorch" in out and "tgs_oracle" not in out
    from diff_gaussian_rasterization import _C
    for name in ("rasterize_gaussians", "rasterize_gaussians_backward", "mark_visible"):
        fn = getattr(_C, name)
        assert inspect.isbuiltin(fn) and fn is getattr(_C._ext, name), f"_C.{name} must be the compiled function"
    doc = _C.rasterize_gaussians.__doc__
    for arg in ("background", "means3D", "colors", "opacity", "scales", "rotations", "scale_modifier", "cov3D_precomp", "viewmatrix", "projmatrix", "tan_fovx",
                "tan_fovy", "image_height", "image_width", "sh", "degree", "campos", "prefiltered", "debug"):
        assert arg in doc                      # positional order of rasterize_points.h:18-38
    src = open(os.path.join(ROOT, "youreditableavatar_amd", "csrc", "tgs_torch_ext.cpp")).read()
    assert "__global__" not in src and "hipLaunchKernel" not in src      # glue only: no kernel outside the C-ABI library


def test_compiled_extension_rejects_cpu_tensors_loudly():
    import torch
    from diff_gaussian_rasterization import _C
    z = torch.zeros(3)
    with pytest.raises(RuntimeError, match="means3D must have dimensions"):
        _C.rasterize_gaussians(z, z, z, z, z, z, 1.0, z, z, z, 1.0, 1.0, 4, 4, z, 0, z, False, False)
    with pytest.raises(RuntimeError, match="no CPU path"):
        _C.rasterize_gaussians(z, torch.zeros(5, 3), z, z, z, z, 1.0, z, z, z, 1.0, 1.0, 4, 4, z, 0, z, False, False)
    with pytest.raises(RuntimeError, match="no CPU path"):
        _C.mark_visible(torch.zeros(5, 3), torch.eye(4), torch.eye(4))
