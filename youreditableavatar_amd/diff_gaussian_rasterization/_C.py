"""``_C`` -- the native-extension surface of the reference package.

The three pybind11 exports of the reference (diff-gaussian-rasterization/ext.cpp:15-19, signatures
rasterize_points.h:18-67) -- ``rasterize_gaussians``, ``rasterize_gaussians_backward``, ``mark_visible`` -- are served by the COMPILED
module ``_Cext`` (csrc/tgs_torch_ext.cpp: pybind11 + torch::Tensor glue over the C ABI of include/tgs_raster.h, built by plain g++),
with identical positional arguments and return tuples; this file loads it and fails loudly when it (or libtgs_raster.so) cannot be
built or loaded.  The batch / introspection / test entry points that have no counterpart in the reference are thin ctypes bindings of the
same C ABI.  Torch supplies device memory and the current HIP stream, nothing else.  There is NO CPU fallback: tensors that are not on a
HIP device are rejected.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Tuple

import torch

from .. import build as _build

_LIB_PATH = os.environ.get("TGS_LIBRARY") or _build.LIB       # TGS_LIBRARY: another build of the same ABI (A/B measurements on one box)


ABI_VERSION = 3                # TGS_ABI_VERSION of include/tgs_raster.h this binding is written against


def _load() -> C.CDLL:
    if "TGS_LIBRARY" not in os.environ:
        try:
            _build.build_native()       # no-op when the library is newer than csrc/*.hip and tgs_raster.h: never run stale kernels
        except Exception as e:  # loud: no silent fallback -- neither to a CPU path nor to a library older than its sources
            what = "is older than its sources" if os.path.exists(_LIB_PATH) else "is missing"
            raise ImportError(f"libtgs_raster.so {what} and could not be rebuilt ({e}); run "
                              f"`python -c 'import __graft_entry__ as g; g.build()'` on a ROCm machine") from e
    lib = C.CDLL(_LIB_PATH, mode=C.RTLD_GLOBAL)        # the compiled _Cext resolves its tgs_* symbols against this very library
    vp, fl, it = C.c_void_p, C.c_float, C.c_int
    lib.tgs_abi_version.restype = it
    if lib.tgs_abi_version() != ABI_VERSION:
        raise ImportError(f"libtgs_raster.so has ABI version {lib.tgs_abi_version()}, this binding needs {ABI_VERSION}: rebuild it "
                          "(`python -m youreditableavatar_amd.build --force`)")
    lib.tgs_sizeof_view.restype = C.c_size_t
    lib.tgs_sizeof_options.restype = C.c_size_t
    lib.tgs_last_error.restype = C.c_char_p
    lib.tgs_forward.restype = C.c_int64
    lib.tgs_forward.argtypes = [vp, vp, vp, it, it, it, vp, it, it, vp, vp, vp, vp, vp, fl, vp, vp, vp, vp, vp, fl, fl, it, vp, vp, it]
    lib.tgs_forward_async.restype = C.c_int64
    lib.tgs_forward_async.argtypes = [C.c_int64] + lib.tgs_forward.argtypes
    lib.tgs_forward_speculative.restype = C.c_int64
    lib.tgs_forward_speculative.argtypes = [C.c_int64, C.POINTER(C.c_int64)] + lib.tgs_forward.argtypes
    lib.tgs_frame_status.restype = it
    lib.tgs_frame_status.argtypes = [vp, vp, C.POINTER(C.c_int64), C.POINTER(it)]
    lib.tgs_backward.restype = it
    lib.tgs_backward.argtypes = [vp, it, it, it, C.c_int64, vp, it, it, vp, vp, vp, vp, fl, vp, vp, vp, vp, vp, fl, fl, vp,
                                 vp, vp, vp, vp] + [vp] * 9 + [it]
    lib.tgs_backward_accumulate.restype = it
    lib.tgs_backward_accumulate.argtypes = lib.tgs_backward.argtypes
    lib.tgs_state_sizes.restype = None
    lib.tgs_state_sizes.argtypes = [it, it, it, it, it, C.c_int64, C.POINTER(C.c_size_t)]
    lib.tgs_forward_opt.restype = C.c_int64
    lib.tgs_forward_opt.argtypes = [vp, it, C.c_int64, vp] + lib.tgs_forward.argtypes
    lib.tgs_backward_opt.restype = it
    lib.tgs_backward_opt.argtypes = [vp, it] + lib.tgs_backward.argtypes
    lib.tgs_forward_views.restype = it
    lib.tgs_set_render_streams.restype = it
    lib.tgs_set_render_streams.argtypes = [vp, it]
    lib.tgs_forward_views.argtypes = [vp, it, C.c_int64, it, it, it, vp, vp, vp, vp, vp, fl, vp, vp, it, it, vp]
    lib.tgs_forward_views_opt.restype = it
    lib.tgs_forward_views_opt.argtypes = [vp] + lib.tgs_forward_views.argtypes
    lib.tgs_backward_render_views.restype = it
    lib.tgs_backward_render_views.argtypes = [vp, it, it, it, vp]
    lib.tgs_backward_render_views_opt.restype = it
    lib.tgs_backward_render_views_opt.argtypes = [vp] + lib.tgs_backward_render_views.argtypes
    lib.tgs_backward_render_opt.restype = it
    lib.tgs_backward_render_opt.argtypes = [vp, vp, it, C.c_int64, vp, it, it, vp, vp, vp]
    lib.tgs_backward_render.restype = it
    lib.tgs_backward_render.argtypes = [vp, it, C.c_int64, vp, it, it, vp, vp, vp]
    lib.tgs_backward_batch.restype = it
    lib.tgs_backward_batch.argtypes = [vp, it, it, it, it, vp, vp, vp, vp, fl, vp, vp, vp, vp, vp, vp, vp, vp, it]
    lib.tgs_backward_batch_range.restype = it
    lib.tgs_backward_batch_range.argtypes = lib.tgs_backward_batch.argtypes + [it, it]
    try:                                                     # (additive since round 6: an older A/B build of the same ABI version lacks it)
        lib.tgs_backward_batch_range_planes.restype = it
        lib.tgs_backward_batch_range_planes.argtypes = lib.tgs_backward_batch_range.argtypes + [C.c_int64]
    except AttributeError:
        pass
    lib.tgs_mark_visible.restype = it
    lib.tgs_mark_visible.argtypes = [vp, it, vp, vp, vp, vp]
    lib.tgs_state_field.restype = C.c_int64
    lib.tgs_state_field.argtypes = [vp, C.c_char_p, it, it, it, C.c_int64, it, it, vp, vp, vp, vp, C.c_size_t]
    lib.tgs_set_sort_lds_cap.restype = it
    lib.tgs_set_sort_lds_cap.argtypes = [C.c_uint]
    lib.tgs_set_instance_pruning.restype = None
    lib.tgs_set_instance_pruning.argtypes = [it]
    lib.tgs_set_forward_group.restype = None
    lib.tgs_set_forward_group.argtypes = [it]
    lib.tgs_last_nonempty_tiles.restype = C.c_int64
    lib.tgs_last_nonempty_tiles.argtypes = []
    lib.tgs_set_deterministic.restype = None
    lib.tgs_set_deterministic.argtypes = [it]
    lib.tgs_selftest_reduce36.restype = it
    lib.tgs_selftest_reduce36.argtypes = [vp, vp, vp]
    lib.tgs_profile_begin.restype = it
    lib.tgs_profile_begin.argtypes = [it]
    lib.tgs_profile_stages.restype = None
    lib.tgs_profile_stages.argtypes = [C.c_uint]
    lib.tgs_profile_end.restype = it
    lib.tgs_profile_end.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    return lib


_lib = _load()


def _load_ext():
    """The compiled glue module.  Built on demand (g++, ~40 s) when missing or older than its source; an ImportError otherwise."""
    import importlib
    try:
        _build.build_torch_ext()
    except Exception as e:
        what = "is older than its sources" if os.path.exists(_build.ext_path()) else "is missing"
        raise ImportError(f"the compiled _C extension ({_build.EXT_NAME}) {what} and could not be rebuilt ({e}); run "
                          f"`python -c 'import __graft_entry__ as g; g.build()'`") from e
    ext = importlib.import_module(__package__ + "." + _build.EXT_NAME)
    if ext.abi_version() != ABI_VERSION or ext.compiled_abi_version() != ABI_VERSION:
        raise ImportError(f"compiled _C extension: ABI version {ext.compiled_abi_version()} (library {ext.abi_version()}), this binding needs {ABI_VERSION}")
    return ext


_ext = _load_ext()
STAGES = ("preprocess_fwd", "scan", "scatter", "tile_sort", "render_fwd", "render_bwd", "preprocess_bwd")


# ---- test-only shims (include/tgs_raster_testing.h): process-wide DEFAULTS for calls made without explicit options.  Every entry point of this
# module passes an explicit tgs_options_t (options(...)); a field left at None / 0 there falls back to what these setters stored.  There is no
# shim for the tile bound any more: with explicit options a tile_bound of 0 means "none", so a thread-wide default could never take effect
# through these bindings -- pass tile_bound= to the call.
def set_sort_lds_cap(cap: int) -> None:
    """Test knob: tile lists longer than ``cap`` (power of two <= 8192) take the global-memory sort path."""
    if _lib.tgs_set_sort_lds_cap(int(cap)) < 0:
        raise _err(-1)


def set_instance_pruning(on: bool) -> None:
    """False: keep the reference's tile instances (every tile of the 3-sigma rectangle); True (default): drop those that cannot
    reach alpha >= 1/255 anywhere in the tile.  Same images and gradients either way."""
    _lib.tgs_set_instance_pruning(1 if on else 0)


def set_forward_group(views_per_launch: int) -> None:
    """Views per launch of the per-Gaussian forward stage inside forward_views (1..8; default 1)."""
    _lib.tgs_set_forward_group(int(views_per_launch))


def last_nonempty_tiles() -> int:
    """Tiles with instances of the last frame this thread rendered with the synchronous or the speculative forward (-1: unknown)."""
    return int(_lib.tgs_last_nonempty_tiles())


def set_deterministic(on: bool) -> None:
    """True: bitwise-reproducible backward (fixed in-tile summation order, slower); False: default."""
    _lib.tgs_set_deterministic(1 if on else 0)


def selftest_reduce36(x: torch.Tensor) -> torch.Tensor:
    """x[64,36] on the GPU -> [4,9] sums over the wave (checks the permlane-swap reduction on hardware)."""
    x = x.contiguous().float()
    out = torch.empty((4, 9), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        if _lib.tgs_selftest_reduce36(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), out.data_ptr()) < 0:
            raise RuntimeError("selftest launch failed")
    return out


def profile_begin(max_records: int = 100000, stages=None) -> None:
    """Bench instrumentation: hipEvents around the pipeline stages (all, or the names in ``stages``) on the caller's
    stream (no syncs)."""
    mask = 0xffffffff if stages is None else sum(1 << STAGES.index(n) for n in stages)
    _lib.tgs_profile_stages(mask)
    if _lib.tgs_profile_begin(int(max_records)) < 0:
        raise RuntimeError("profiling already active")


def profile_end():
    """-> {stage: (total_ms, launches)}; waits for the recorded events."""
    ms = (C.c_double * len(STAGES))()
    cnt = (C.c_int64 * len(STAGES))()
    if _lib.tgs_profile_end(ms, cnt) < 0:
        raise RuntimeError("profiling not active")
    return {n: (float(ms[i]), int(cnt[i])) for i, n in enumerate(STAGES)}
_ALLOC_T = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int, C.c_size_t)


class _OptionsT(C.Structure):
    """tgs_options_t (include/tgs_raster.h): everything that tunes one call, passed explicitly -- nothing process-wide is touched"""
    _fields_ = [("struct_size", C.c_uint32), ("instance_pruning", C.c_int32), ("deterministic", C.c_int32), ("forward_group", C.c_int32),
                ("sort_lds_cap", C.c_uint32), ("tile_bound", C.c_int64), ("heavy_bound", C.c_int64), ("mid_bound", C.c_int64),
                ("light_tiles", C.c_int32)]


class _FrameInfoT(C.Structure):
    """tgs_frame_info_t"""
    _fields_ = [("num_rendered", C.c_int64), ("nonempty_tiles", C.c_int64), ("flags", C.c_int32), ("mid_tiles", C.c_int32)]


def options(tile_bound: int = 0, pruning: Optional[bool] = None, deterministic: Optional[bool] = None, forward_group: int = 0, sort_lds_cap: int = 0,
            light_tiles: Optional[bool] = None, mid_bound: int = 0, heavy_bound: int = 0) -> _OptionsT:
    """A filled tgs_options_t (None / 0: the library default of that field)."""
    o = _OptionsT()
    o.struct_size = C.sizeof(_OptionsT)
    o.instance_pruning = -1 if pruning is None else int(bool(pruning))
    o.deterministic = -1 if deterministic is None else int(bool(deterministic))
    o.forward_group, o.sort_lds_cap = int(forward_group), int(sort_lds_cap)
    o.tile_bound, o.mid_bound, o.heavy_bound = max(0, int(tile_bound)), max(0, int(mid_bound)), max(0, int(heavy_bound))
    o.light_tiles = -1 if light_tiles is None else int(bool(light_tiles))
    return o


def loaded_library() -> str:
    """Path of the native library this process is using (for diagnostics)."""
    return _LIB_PATH


def _err(code: int) -> RuntimeError:
    msg = _lib.tgs_last_error()
    return RuntimeError(f"{msg.decode() if msg else 'tgs_raster error'} (code {code})")


def _dev_f32(t: torch.Tensor, dev: torch.device, name: str) -> Optional[torch.Tensor]:
    """contiguous fp32 tensor on ``dev``; an empty tensor is the reference's 'absent' (NULL)."""
    if t is None or t.numel() == 0:
        return None
    if t.dtype != torch.float32:
        raise RuntimeError(f"expected scalar type Float but found {t.dtype} for {name}")
    if t.device != dev:
        t = t.to(dev)
    return t.contiguous()


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _require_gpu(means3D: torch.Tensor) -> torch.device:
    if not means3D.is_cuda:
        raise RuntimeError("diff_gaussian_rasterization (MI355X build) has no CPU path: means3D must be on a HIP device")
    return means3D.device


def _rasterize_gaussians_ctypes(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                                viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                                prefiltered, debug, r_capacity: Optional[int] = None, r_guess: Optional[int] = None, tile_bound: int = 0,
                                pruning: Optional[bool] = None, sort_lds_cap: int = 0, info: bool = False, mid_bound: int = 0,
                                light_tiles: Optional[bool] = None
                                ) -> Tuple[int, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """RasterizeGaussiansCUDA (rasterize_points.cu:35-115) over ctypes -- kept for A/B measurements of the host cost
    (``TGS_CTYPES_GLUE=1``); the module-level ``rasterize_gaussians`` is the compiled one (csrc/tgs_torch_ext.cpp).

    ``r_capacity`` (extension, tgs_forward_async): render without the host read-back of num_rendered; the binning
    buffer holds ``r_capacity`` instances and that number is returned in place of num_rendered.  Check the frame with
    ``frame_status`` / ``frame_meta`` afterwards: a rejected frame renders as background and back-propagates nothing.

    ``r_guess`` (extension, tgs_forward_speculative): the complete frame like the plain call, but the stages behind the scan are
    enqueued against the guessed instance count while the read-back is in flight (they run again if the guess was too small).
    Returns an 8-tuple then (also with ``info=True``): the value to pass as ``R`` to the backward / ``state_field`` first, the true
    num_rendered and the pair (tiles with instances, tiles with >= 128 instances) last -- the exact ``tile_bound`` / ``mid_bound`` of the
    frame's backward.  ``tile_bound`` / ``pruning`` / ``sort_lds_cap``: tgs_options_t fields."""
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    dev = _require_gpu(means3D)
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    M = int(sh.size(1)) if (sh.dim() > 1 and sh.size(0) != 0) else 0
    with torch.cuda.device(dev):
        out_color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        bufs: List[Optional[torch.Tensor]] = [torch.empty(0, dtype=torch.uint8, device=dev) for _ in range(3)]

        def alloc(_ctx, which, nbytes):     # resizeFunctional (rasterize_points.cu:27-33)
            bufs[which] = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
            return bufs[which].data_ptr()

        cb = _ALLOC_T(alloc)
        t = dict(bg=_dev_f32(background, dev, "background"), means=_dev_f32(means3D, dev, "means3D"),
                 colors=_dev_f32(colors, dev, "colors"), opac=_dev_f32(opacity, dev, "opacity"),
                 scales=_dev_f32(scales, dev, "scales"), rots=_dev_f32(rotations, dev, "rotations"),
                 cov=_dev_f32(cov3D_precomp, dev, "cov3D_precomp"), view=_dev_f32(viewmatrix, dev, "viewmatrix"),
                 proj=_dev_f32(projmatrix, dev, "projmatrix"), sh=_dev_f32(sh, dev, "sh"), campos=_dev_f32(campos, dev, "campos"))
        stream = torch.cuda.current_stream(dev).cuda_stream
        args = (C.cast(cb, C.c_void_p), None, stream, P, int(degree), M, _p(t["bg"]), W, H, _p(t["means"]),
                _p(t["sh"]), _p(t["colors"]), _p(t["opac"]), _p(t["scales"]), float(scale_modifier), _p(t["rots"]),
                _p(t["cov"]), _p(t["view"]), _p(t["proj"]), _p(t["campos"]), float(tan_fovx), float(tan_fovy),
                int(bool(prefiltered)), out_color.data_ptr(), _p(radii) if P else None, int(bool(debug)))
        opt, fi = options(tile_bound=tile_bound, pruning=pruning, sort_lds_cap=sort_lds_cap, mid_bound=mid_bound, light_tiles=light_tiles), _FrameInfoT()
        mode, rr = (2, int(r_guess)) if r_guess is not None else ((1, int(r_capacity)) if r_capacity is not None else (0, 0))
        r = _lib.tgs_forward_opt(C.byref(opt), mode, rr, C.byref(fi), *args)
        if r < 0:
            raise _err(int(r))
    if r_guess is not None or info:
        return int(r), out_color, radii, bufs[0], bufs[1], bufs[2], int(fi.num_rendered), (int(fi.nonempty_tiles), int(fi.mid_tiles))
    return int(r), out_color, radii, bufs[0], bufs[1], bufs[2]


META_BYTES = 64
FRAME_PREFILTERED, FRAME_REJECTED = 1, 2


def frame_meta(image_buffer: torch.Tensor) -> torch.Tensor:
    """The frame's 64-byte Meta record (a view of the image buffer, on the device): gather these for a batch of
    sync-free frames and decode them on the host with ``decode_meta`` after ONE copy."""
    return image_buffer[:META_BYTES]


def decode_meta_full(meta_bytes: torch.Tensor) -> Tuple[int, int, int, int]:
    """64 Meta bytes on the host -> (num_rendered, flags, longest tile list, tiles beyond the LDS sort).  In a record the scan kernel wrote
    to pinned memory itself (tgs_view_t.host_meta) the word at byte 32 says that a tile bound was exceeded: folded into FRAME_REJECTED."""
    import struct
    raw = bytes(meta_bytes.cpu().numpy().tobytes())
    R, max_count, n_overflow, flags = struct.unpack_from("<QIII", raw, 0)
    if struct.unpack_from("<I", raw, 32)[0]:
        flags |= FRAME_REJECTED
    return int(R), int(flags), int(max_count), int(n_overflow)


def decode_meta_tiles(meta_bytes: torch.Tensor) -> Tuple[int, int, int]:
    """64 Meta bytes on the host -> tiles that hold instances, tiles with >= 1024 of them, tiles with >= 128."""
    import struct
    n, heavy, mid = struct.unpack_from("<III", bytes(meta_bytes.cpu().numpy().tobytes()), 20)
    return int(n), int(heavy), int(mid)


def decode_meta(meta_bytes: torch.Tensor) -> Tuple[int, int]:
    """64 Meta bytes on the host -> (num_rendered, flags)."""
    R, flags, _m, _o = decode_meta_full(meta_bytes)
    return R, flags


def frame_status(image_buffer: torch.Tensor) -> Tuple[int, int]:
    """tgs_frame_status: synchronises the current stream; -> (true num_rendered, TGS_FRAME_* flags)."""
    dev = image_buffer.device
    R, fl = C.c_int64(0), C.c_int(0)
    with torch.cuda.device(dev):
        r = _lib.tgs_frame_status(torch.cuda.current_stream(dev).cuda_stream, image_buffer.data_ptr(), C.byref(R), C.byref(fl))
    if r < 0:
        raise _err(int(r))
    return int(R.value), int(fl.value)


def _rasterize_gaussians_backward_ctypes(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                         viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh, degree, campos,
                                         geomBuffer, R, binningBuffer, imageBuffer, debug, _with_conic=False, tile_bound: int = 0,
                                         deterministic: Optional[bool] = None, mid_bound: int = 0, light_tiles: Optional[bool] = None,
                                         need_colors: bool = True, need_cov3D: bool = True):
    """RasterizeGaussiansBackwardCUDA (rasterize_points.cu:117-196) over ctypes (see _rasterize_gaussians_ctypes); return order of :195.
    ``_with_conic`` (tests only) appends the scratch tensor dL_dconic[P,2,2].  ``need_colors`` / ``need_cov3D`` False: the caller discards
    dL_dcolors (SH path) / dL_dcov3D (scale + rotation path) -- not written, empty tensors returned (as the compiled module does)."""
    dev = _require_gpu(means3D)
    P = int(means3D.size(0))
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    M = int(sh.size(1)) if (sh.dim() > 1 and sh.size(0) != 0) else 0
    with torch.cuda.device(dev):
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        has_sr0 = scales is not None and scales.numel() != 0
        want_col, want_cov = bool(need_colors) or M == 0, bool(need_cov3D) or not has_sr0
        dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dconic = e(P, 3), e(P, 3), e(P if want_col else 0, 3), e(P if _with_conic else 0, 2, 2)
        dL_dopacity, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations = e(P, 1), e(P if want_cov else 0, 6), e(P, M, 3), e(P, 3), e(P, 4)
        t = dict(bg=_dev_f32(background, dev, "background"), means=_dev_f32(means3D, dev, "means3D"),
                 colors=_dev_f32(colors, dev, "colors"), scales=_dev_f32(scales, dev, "scales"),
                 rots=_dev_f32(rotations, dev, "rotations"), cov=_dev_f32(cov3D_precomp, dev, "cov3D_precomp"),
                 view=_dev_f32(viewmatrix, dev, "viewmatrix"), proj=_dev_f32(projmatrix, dev, "projmatrix"),
                 sh=_dev_f32(sh, dev, "sh"), campos=_dev_f32(campos, dev, "campos"), dL=_dev_f32(dL_dout_color, dev, "dL_dout_color"))
        if P != 0:
            has_sr = t["scales"] is not None
            if not has_sr:      # the reference leaves these at zero on the cov3D_precomp path
                dL_dscales.zero_(); dL_drotations.zero_()
            radii_c = radii.contiguous()
            stream = torch.cuda.current_stream(dev).cuda_stream
            opt = options(tile_bound=tile_bound, deterministic=deterministic, mid_bound=mid_bound, light_tiles=light_tiles)
            r = _lib.tgs_backward_opt(C.byref(opt), 0, stream, P, int(degree), M, int(R), _p(t["bg"]), W, H, _p(t["means"]), _p(t["sh"]), _p(t["colors"]),
                                  _p(t["scales"]), float(scale_modifier), _p(t["rots"]), _p(t["cov"]), _p(t["view"]), _p(t["proj"]),
                                  _p(t["campos"]), float(tan_fovx), float(tan_fovy), radii_c.data_ptr(), geomBuffer.data_ptr(),
                                  binningBuffer.data_ptr(), imageBuffer.data_ptr(), _p(t["dL"]), dL_dmeans2D.data_ptr(),
                                  dL_dconic.data_ptr() if _with_conic else None, dL_dopacity.data_ptr(), dL_dcolors.data_ptr() if want_col else None, dL_dmeans3D.data_ptr(),
                                  dL_dcov3D.data_ptr() if want_cov else None, dL_dsh.data_ptr() if M else None,
                                  dL_dscales.data_ptr() if has_sr else None, dL_drotations.data_ptr() if has_sr else None,
                                  int(bool(debug)))
            if r < 0:
                raise _err(int(r))
    out = (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)
    return out + (dL_dconic,) if _with_conic else out


def rasterize_gaussians_backward_accumulate(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                            viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh, degree, campos,
                                            geomBuffer, R, binningBuffer, imageBuffer, debug, into, tile_bound: int = 0,
                                            deterministic: Optional[bool] = None, mid_bound: int = 0):
    """Multi-view extension (tgs_backward_accumulate): parameter gradients are ADDED into the fp32 tensors of ``into``
    (keys: means3D, opacities, and sh|colors_precomp, scales+rotations|cov3D_precomp; contiguous, on the device).
    Returns dL_dmeans2D[P,3], the only per-view gradient."""
    dev = _require_gpu(means3D)
    P = int(means3D.size(0))
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    M = int(sh.size(1)) if (sh.dim() > 1 and sh.size(0) != 0) else 0
    with torch.cuda.device(dev):
        dL_dmeans2D = torch.empty((P, 3), dtype=torch.float32, device=dev)
        if P == 0:
            return dL_dmeans2D
        dL_dconic = torch.empty((P, 4), dtype=torch.float32, device=dev)
        t = dict(bg=_dev_f32(background, dev, "background"), means=_dev_f32(means3D, dev, "means3D"),
                 colors=_dev_f32(colors, dev, "colors"), scales=_dev_f32(scales, dev, "scales"),
                 rots=_dev_f32(rotations, dev, "rotations"), cov=_dev_f32(cov3D_precomp, dev, "cov3D_precomp"),
                 view=_dev_f32(viewmatrix, dev, "viewmatrix"), proj=_dev_f32(projmatrix, dev, "projmatrix"),
                 sh=_dev_f32(sh, dev, "sh"), campos=_dev_f32(campos, dev, "campos"), dL=_dev_f32(dL_dout_color, dev, "dL_dout_color"))

        def dst(name, shape):
            g = into.get(name)
            if g is None:
                return None
            if g.dtype != torch.float32 or g.device != dev or not g.is_contiguous() or tuple(g.shape) != shape:
                raise RuntimeError(f"accumulation buffer for {name} must be a contiguous fp32 tensor of shape {shape} on {dev}")
            return g.data_ptr()

        has_sh, has_sr = t["sh"] is not None, t["scales"] is not None
        opt = options(tile_bound=tile_bound, deterministic=deterministic, mid_bound=mid_bound)
        r = _lib.tgs_backward_opt(
            C.byref(opt), 1, torch.cuda.current_stream(dev).cuda_stream, P, int(degree), M, int(R), _p(t["bg"]), W, H, _p(t["means"]), _p(t["sh"]), _p(t["colors"]),
            _p(t["scales"]), float(scale_modifier), _p(t["rots"]), _p(t["cov"]), _p(t["view"]), _p(t["proj"]), _p(t["campos"]),
            float(tan_fovx), float(tan_fovy), radii.contiguous().data_ptr(), geomBuffer.data_ptr(), binningBuffer.data_ptr(),
            imageBuffer.data_ptr(), _p(t["dL"]), dL_dmeans2D.data_ptr(), dL_dconic.data_ptr(), dst("opacities", (P, 1)),
            None if has_sh else dst("colors_precomp", (P, 3)), dst("means3D", (P, 3)), None if has_sr else dst("cov3D_precomp", (P, 6)),
            dst("sh", (P, M, 3)) if has_sh else None, dst("scales", (P, 3)) if has_sr else None, dst("rotations", (P, 4)) if has_sr else None,
            int(bool(debug)))
        if r < 0:
            raise _err(int(r))
    return dL_dmeans2D


class _ViewT(C.Structure):
    """tgs_view_t (include/tgs_raster.h)"""
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("tan_fovx", C.c_float), ("tan_fovy", C.c_float),
                ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p), ("radii", C.c_void_p),
                ("geom_buffer", C.c_void_p), ("binning_buffer", C.c_void_p), ("img_buffer", C.c_void_p), ("R", C.c_int64),
                ("dL_dmean2D", C.c_void_p), ("dL_dcolor", C.c_void_p),
                ("background", C.c_void_p), ("out_color", C.c_void_p), ("radii_out", C.c_void_p), ("dL_dpix", C.c_void_p),
                ("geom_bytes", C.c_size_t), ("binning_bytes", C.c_size_t), ("img_bytes", C.c_size_t), ("colors_precomp", C.c_void_p),
                ("tile_bound", C.c_int64), ("heavy_bound", C.c_int64), ("mid_bound", C.c_int64), ("host_meta", C.c_void_p)]


if _lib.tgs_sizeof_view() != C.sizeof(_ViewT) or _ext.sizeof_view() != C.sizeof(_ViewT) or _lib.tgs_sizeof_options() != C.sizeof(_OptionsT):
    raise ImportError(f"tgs_view_t / tgs_options_t: the library says {_lib.tgs_sizeof_view()} / {_lib.tgs_sizeof_options()} bytes, this binding declares "
                      f"{C.sizeof(_ViewT)} / {C.sizeof(_OptionsT)} -- a stale library or binding")
ViewArray = lambda n: (_ViewT * n)()


def state_sizes(P: int, width: int, height: int, has_sh: bool, has_scale_rot: bool, r_capacity: int) -> Tuple[int, int, int]:
    """tgs_state_sizes -> bytes of the (geometry, binning, image) state buffers of one view."""
    out = (C.c_size_t * 3)()
    _lib.tgs_state_sizes(int(P), int(width), int(height), int(bool(has_sh)), int(bool(has_scale_rot)), int(r_capacity), out)
    return int(out[0]), int(out[1]), int(out[2])


def forward_views(stream_handles, r_capacity, P, D, M, means3D, shs, opacities, scales, scale_modifier, rotations, views, n_views, prefiltered=False,
                  opt: Optional[_OptionsT] = None) -> None:
    """tgs_forward_views_opt on prepared device pointers (ints) and a filled ``ViewArray``; ``opt``: ``options(...)`` or None (defaults)."""
    arr = (C.c_void_p * len(stream_handles))(*stream_handles)
    r = _lib.tgs_forward_views_opt(C.byref(opt) if opt is not None else None, arr, len(stream_handles), int(r_capacity), int(P), int(D), int(M), means3D, shs, None,
                                   opacities, scales, float(scale_modifier), rotations, None, int(bool(prefiltered)), int(n_views), C.cast(views, C.c_void_p))
    if r < 0:
        raise _err(int(r))


def set_render_streams(stream_handles) -> None:
    """tgs_set_render_streams: later forward_views calls of this thread put k_render_fwd of view k on stream_handles[k mod n] ([] resets)."""
    arr = (C.c_void_p * max(1, len(stream_handles)))(*stream_handles)
    _lib.tgs_set_render_streams(arr, len(stream_handles))


def backward_render_views(stream_handles, P, views, n_views, opt: Optional[_OptionsT] = None) -> None:
    arr = (C.c_void_p * len(stream_handles))(*stream_handles)
    r = _lib.tgs_backward_render_views_opt(C.byref(opt) if opt is not None else None, arr, len(stream_handles), int(P), int(n_views),
                                           views if isinstance(views, C.c_void_p) else C.cast(views, C.c_void_p))
    if r < 0:
        raise _err(int(r))


def backward_batch_raw(stream, P, D, M, views, n_views, means3D, shs, scales, scale_modifier, rotations, dL_dopacity, dL_dmean3D, dL_dsh, dL_dscale,
                       dL_drot, accumulate, first: int = 0, count: Optional[int] = None, dsh_plane_stride: int = 0) -> None:
    """tgs_backward_batch[_range] on prepared device pointers (scales/rotations path; ``shs`` None: per-view colours, their gradients go to
    the views' dL_dcolor).  ``first`` / ``count``: only Gaussians [first, first + count) (multiples of 256, or ending at P).
    ``dsh_plane_stride`` > 0: dL_dsh level-major (tgs_backward_batch_range_planes: coefficient k of Gaussian p at k * stride + 3 p)."""
    args = (stream, int(P), int(D), int(M), int(n_views), views if isinstance(views, C.c_void_p) else C.cast(views, C.c_void_p), means3D, shs,
            scales, float(scale_modifier),
            rotations, None, dL_dopacity, dL_dmean3D, None, dL_dsh if shs else None, dL_dscale, dL_drot, 1 if accumulate else 0,
            int(first), int(P - first if count is None else count))
    r = _lib.tgs_backward_batch_range_planes(*args, int(dsh_plane_stride)) if dsh_plane_stride else _lib.tgs_backward_batch_range(*args)
    if r < 0:
        raise _err(int(r))


def rasterize_gaussians_backward_render(background, dL_dout_color, R, binningBuffer, imageBuffer, P, tile_bound: int = 0,
                                        deterministic: Optional[bool] = None) -> None:
    """tgs_backward_render: the per-pixel half of one view's backward; the tile partials stay in ``binningBuffer`` for
    ``rasterize_gaussians_backward_batch``."""
    dev = _require_gpu(dL_dout_color)
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    with torch.cuda.device(dev):
        bg, dL = _dev_f32(background, dev, "background"), _dev_f32(dL_dout_color, dev, "dL_dout_color")
        opt = options(tile_bound=tile_bound, deterministic=deterministic)
        r = _lib.tgs_backward_render_opt(C.byref(opt), torch.cuda.current_stream(dev).cuda_stream, int(P), int(R), _p(bg), W, H, binningBuffer.data_ptr(),
                                         imageBuffer.data_ptr(), _p(dL))
    if r < 0:
        raise _err(int(r))


def rasterize_gaussians_backward_batch(views, means3D, sh, degree, scales, rotations, scale_modifier, cov3D_precomp, into, accumulate=True):
    """tgs_backward_batch: the per-Gaussian half for all ``views`` of a batch in one pass.

    ``views``: dicts with keys viewmatrix, projmatrix, campos, tanfovx, tanfovy, image_height, image_width, radii, geom, binning, img,
    R (and colors=True on the colors_precomp path).  Parameter gradients are stored (``accumulate=False``) or added into the tensors
    of ``into`` (keys: means3D, opacities, and sh, scales + rotations | cov3D_precomp).  Returns, per view, dL_dmeans2D[P,3]
    (and dL_dcolors[P,3] on the colors_precomp path)."""
    dev = _require_gpu(means3D)
    P = int(means3D.size(0))
    M = int(sh.size(1)) if (sh is not None and sh.dim() > 1 and sh.size(0) != 0) else 0
    has_sh = M > 0
    outs = []
    if P == 0 or not views:
        return outs
    with torch.cuda.device(dev):
        t = dict(means=_dev_f32(means3D, dev, "means3D"), sh=_dev_f32(sh, dev, "sh") if has_sh else None, scales=_dev_f32(scales, dev, "scales"),
                 rots=_dev_f32(rotations, dev, "rotations"), cov=_dev_f32(cov3D_precomp, dev, "cov3D_precomp"))
        has_sr = t["scales"] is not None
        arr = (_ViewT * len(views))()
        keep = []
        for i, v in enumerate(views):
            vm, pm, cp = _dev_f32(v["viewmatrix"], dev, "viewmatrix"), _dev_f32(v["projmatrix"], dev, "projmatrix"), _dev_f32(v["campos"], dev, "campos")
            radii = v["radii"].contiguous()
            g2d = torch.empty((P, 3), dtype=torch.float32, device=dev)
            gcol = None if has_sh else torch.empty((P, 3), dtype=torch.float32, device=dev)
            keep += [vm, pm, cp, radii]
            a = arr[i]
            a.width, a.height, a.tan_fovx, a.tan_fovy = int(v["image_width"]), int(v["image_height"]), float(v["tanfovx"]), float(v["tanfovy"])
            a.viewmatrix, a.projmatrix, a.campos, a.radii = vm.data_ptr(), pm.data_ptr(), cp.data_ptr(), radii.data_ptr()
            a.geom_buffer, a.binning_buffer, a.img_buffer, a.R = v["geom"].data_ptr(), v["binning"].data_ptr(), v["img"].data_ptr(), int(v["R"])
            a.dL_dmean2D, a.dL_dcolor = g2d.data_ptr(), (gcol.data_ptr() if gcol is not None else None)
            outs.append((g2d, gcol))

        def dst(name, shape):
            g = into.get(name)
            if g is None:
                raise RuntimeError(f"rasterize_gaussians_backward_batch: no gradient buffer for {name}")
            if g.dtype != torch.float32 or g.device != dev or not g.is_contiguous() or tuple(g.shape) != shape:
                raise RuntimeError(f"gradient buffer for {name} must be a contiguous fp32 tensor of shape {shape} on {dev}")
            return g.data_ptr()

        r = _lib.tgs_backward_batch(torch.cuda.current_stream(dev).cuda_stream, P, int(degree), M, len(views), C.cast(arr, C.c_void_p), _p(t["means"]),
                                    _p(t["sh"]), _p(t["scales"]), float(scale_modifier), _p(t["rots"]), _p(t["cov"]), dst("opacities", (P, 1)),
                                    dst("means3D", (P, 3)), None if has_sr else dst("cov3D_precomp", (P, 6)), dst("sh", (P, M, 3)) if has_sh else None,
                                    dst("scales", (P, 3)) if has_sr else None, dst("rotations", (P, 4)) if has_sr else None, 1 if accumulate else 0)
        del keep
    if r < 0:
        raise _err(int(r))
    return outs


def _mark_visible_ctypes(means3D, viewmatrix, projmatrix) -> torch.Tensor:
    """markVisible (rasterize_points.cu:198-217) over ctypes."""
    dev = _require_gpu(means3D)
    P = int(means3D.size(0))
    with torch.cuda.device(dev):
        present = torch.zeros((P,), dtype=torch.bool, device=dev)
        if P != 0:
            m, v, pj = _dev_f32(means3D, dev, "means3D"), _dev_f32(viewmatrix, dev, "viewmatrix"), _dev_f32(projmatrix, dev, "projmatrix")
            r = _lib.tgs_mark_visible(torch.cuda.current_stream(dev).cuda_stream, P, _p(m), _p(v), _p(pj), present.data_ptr())
            if r < 0:
                raise _err(int(r))
    return present


_FIELD_DTYPES = {"n_contrib": torch.int32, "final_T": torch.float32, "ranges": torch.int32, "point_list": torch.int32, "block_masks": torch.int32,
                 "means2D": torch.float32, "depths": torch.float32, "conic_opacity": torch.float32, "rgb": torch.float32,
                 "tiles_touched": torch.int32, "tile_order": torch.int32, "stamps": torch.int64, "quad_masks": torch.int64}


def state_field(name: str, P: int, width: int, height: int, R: int, has_sh: bool, has_scale_rot: bool,
                geomBuffer, binningBuffer, imageBuffer) -> torch.Tensor:
    """Test/bench introspection of the opaque state buffers (tgs_state_field)."""
    dev = geomBuffer.device
    T = ((width + 15) // 16) * ((height + 15) // 16)
    count = {"n_contrib": width * height, "final_T": width * height, "ranges": 2 * T, "point_list": R, "block_masks": R, "quad_masks": R, "means2D": 2 * P,
             "depths": P, "conic_opacity": 4 * P, "rgb": 3 * P, "tiles_touched": P, "tile_order": T, "stamps": 8 * T}[name]
    out = torch.empty((count,), dtype=_FIELD_DTYPES[name], device=dev)
    with torch.cuda.device(dev):
        r = _lib.tgs_state_field(torch.cuda.current_stream(dev).cuda_stream, name.encode(), P, width, height, int(R),
                                 int(has_sh), int(has_scale_rot), geomBuffer.data_ptr(), binningBuffer.data_ptr(),
                                 imageBuffer.data_ptr(), out.data_ptr(), out.numel() * out.element_size())
    if r < 0:
        raise _err(int(r))
    return out


# The reference's three exports (ext.cpp:15-19): the compiled module's functions, positional signatures of rasterize_points.h:18-67
# (+ the keyword-only extensions r_capacity / r_guess / _with_conic documented in csrc/tgs_torch_ext.cpp).
if os.environ.get("TGS_CTYPES_GLUE") == "1":        # A/B of the host cost only
    rasterize_gaussians, rasterize_gaussians_backward, mark_visible = _rasterize_gaussians_ctypes, _rasterize_gaussians_backward_ctypes, _mark_visible_ctypes
else:
    rasterize_gaussians = _ext.rasterize_gaussians
    rasterize_gaussians_backward = _ext.rasterize_gaussians_backward
    mark_visible = _ext.mark_visible
