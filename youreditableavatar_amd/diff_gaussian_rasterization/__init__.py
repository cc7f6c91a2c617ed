"""Drop-in ``diff_gaussian_rasterization`` for MI355X.

Public surface = the reference package's (diff_gaussian_rasterization/__init__.py:21-220):
``GaussianRasterizationSettings``, ``GaussianRasterizer`` and ``rasterize_gaussians`` with the same
field order, argument names, validation messages, debug snapshot files and gradient ordering, so
``tetgs_scene`` (tetgs_model.py:7, tetgs_edit_2d.py:6, tetgs_edit_3d.py:5) imports it unchanged.
"""
from __future__ import annotations

from typing import NamedTuple

import os

import torch
import torch.nn as nn

from . import _C

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]


class GaussianRasterizationSettings(NamedTuple):
    # field order of __init__.py:157-169 (callers construct it by keyword, tetgs_model.py:507-520)
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _cpu_snapshot(args):
    """Host copies taken BEFORE the native call (debug mode, __init__.py:18-20)."""
    return tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)


def _call_native(fn, args, debug: bool, dump_name: str, what: str):
    if not debug:
        return fn(*args)
    saved = _cpu_snapshot(args)
    try:
        return fn(*args)
    except Exception:
        torch.save(saved, dump_name)
        print(f"\nAn error occured in {what}. Please forward {dump_name} for debugging.")
        raise


_SPECULATE = os.environ.get("TGS_SPECULATIVE_FORWARD", "1") != "0"


class _Speculation:
    """Instance-count guesses of the speculative forward (tgs_forward_speculative), per (P, H, W, device).

    The guess is a slowly decaying maximum of the counts seen (trainers sample views in random order -- refine.py:257-266 -- so the last
    call's count alone misses on every small-to-large transition, and a miss enqueues scatter / sort / render twice) plus 15 % headroom.
    MAX_MISSES misses in a row switch speculation off for that key for the next COOLDOWN calls (e.g. frames that need the host-sized overflow
    sort, which always miss).

    Round 5: the maximum decays by 0.1 % per call, not 3 %.  At 3 % the 15 % of headroom are gone five calls after the largest view, so a
    camera set whose instance counts differ by more than that (the 16 views of bench.py's drop-in loop do: an ellipsoid seen end-on and
    side-on) missed whenever a large view came round again -- scatter / sort / render enqueued twice, ~130 us -- and spent most frames in
    the 64-call cool-down on the synchronous forward (rocprofv3 of tools/dropin_loop.py: the device-to-host copy of the synchronous path in
    65 of 100 frames).  What a generous guess costs is memory for the binning buffer (112 B per instance of the largest view) and
    workgroups that return at once.  A miss also lifts the bound 10 % above the count that caused it (a run of growing views -- the first
    calls of a camera set -- then misses once or twice, not on every view), and the cool-down is 16 calls after four misses in a row."""

    HEADROOM, DECAY, MAX_MISSES, COOLDOWN, GRANULE, FAR_CALLS = 1.15, 0.999, 4, 16, 65536, 64

    def __init__(self):
        self.state = {}

    def guess(self, key):
        st = self.state.get(key)
        if st is None:
            return None
        if st[2] > 0:                       # cooling down after repeated misses
            st[2] -= 1
            return None
        return (int(st[0] * self.HEADROOM) + self.GRANULE - 1) // self.GRANULE * self.GRANULE

    def tile_guess(self, key):
        """bound on the tiles with instances for the grids of the speculative forward (0: none yet): the same kind of decaying maximum"""
        st = self.state.get(key)
        return (int(st[3] * self.HEADROOM) + 64) // 64 * 64 if st is not None and st[3] > 0 else 0

    LIGHT_TILES_MIN = 4096      # light tiles (fewer than 128 instances) of the last frame from which the light groups pay for a frame that has the GPU to itself

    def light_tiles(self, key):
        """Whether the render kernels should composite light tiles several per workgroup (tgs_options_t::light_tiles) for the next frame of
        this key: yes when the last frame had many of them.  Measured on the MI355X, one view per step: at 1920x1080 (1800 light tiles) the
        groups cost 1.4 % (0.381 -> 0.387 ms per frame: the kernels end with the groups' ~10-us tail), at 2048x2048 they gain 1-5 % of the
        trainers' step (0.823 -> 0.778 ms at SH degree 3)."""
        st = self.state.get(key)
        return bool(st is not None and len(st) > 4 and st[4] >= self.LIGHT_TILES_MIN)

    def update(self, key, true_count, guess, tiles=-1, tile_guess=0, mid_tiles=-1):
        st = self.state.setdefault(key, [0, 0, 0, 0, 0])      # [bound, consecutive misses, calls left without speculation, bound on the tiles, light tiles of the last frame]
        missed = guess is not None and (true_count > guess or (tile_guess > 0 and tiles > tile_guess))     # (the ONE place that decides what a miss is)
        lift = 1.1 if missed else 1.0
        # A bound far above what the views need is memory (112 B per instance): when the last FAR_CALLS calls all stayed below half of it -- one
        # outlier view long ago, or a camera set that moved closer -- it falls to twice the largest of them at once instead of by 0.1 % per call.
        far = st[5] if len(st) > 5 else [0, 0]
        if not missed and 2 * true_count < st[0]:
            far = [far[0] + 1, max(far[1], int(true_count))]
            if far[0] >= self.FAR_CALLS:
                st[0], far = min(st[0], 2 * far[1]), [0, 0]
        else:
            far = [0, 0]
        if len(st) > 5:
            st[5] = far
        else:
            st.append(far)
        st[0] = max(int(true_count * lift), int(st[0] * self.DECAY))
        if tiles >= 0:
            st[3] = max(int(tiles * lift), int(st[3] * self.DECAY))
            if mid_tiles >= 0:
                st[4] = max(0, int(tiles) - int(mid_tiles))
        if guess is not None:
            st[1] = st[1] + 1 if missed else 0
            if st[1] >= self.MAX_MISSES:
                st[1], st[2] = 0, self.COOLDOWN


_speculation = _Speculation()
_KEEP_ALL_OUTPUTS = os.environ.get("TGS_KEEP_ALL_OUTPUTS") == "1"      # A/B: have the backward write dL_dcolors / dL_dcov3D even when autograd drops them


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
        rs = raster_settings
        # native argument order (__init__.py:60-80)
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
                rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh, rs.sh_degree, rs.campos,
                rs.prefiltered, rs.debug)
        # The forward reads num_rendered back like the reference (rasterizer_impl.cu:280-281), but does not wait for it before it
        # enqueues the stages behind the scan: they run against last call's count + headroom (tgs_forward_speculative; a frame that
        # needs more repeats them with the exact sizes).  num_rendered below is what the binning buffer is carved for.
        key = (int(means3D.shape[0]), int(rs.image_height), int(rs.image_width), means3D.device)
        guess = _speculation.guess(key) if _SPECULATE else None
        light = _speculation.light_tiles(key) if os.environ.get("TGS_LIGHT_TILES") is None else None      # (None: the library's default / the A-B variable)
        tile_guess = 0
        # (the extension keywords travel to the library as an explicit tgs_options_t: no process- or thread-wide knob is touched)
        if guess is not None:
            # the grids of the stages enqueued ahead of the read-back cover a guessed number of tiles with instances, not all tiles
            tile_guess = _speculation.tile_guess(key)
            num_rendered, color, radii, geom, binning, img, true_R, (tiles, mid_tiles) = _call_native(
                lambda *a: _C.rasterize_gaussians(*a, r_guess=guess, tile_bound=tile_guess, light_tiles=light), args, rs.debug, "snapshot_fw.dump", "forward")
        else:
            num_rendered, color, radii, geom, binning, img, true_R, (tiles, mid_tiles) = _call_native(
                lambda *a: _C.rasterize_gaussians(*a, info=True, light_tiles=light), args, rs.debug, "snapshot_fw.dump", "forward")
        # tiles / mid_tiles: tiles with instances / with >= 128 instances of THIS frame (the forward has read its Meta); -1: unknown
        if _SPECULATE:
            _speculation.update(key, true_R, guess, tiles, tile_guess, mid_tiles)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.nonempty_tiles, ctx.mid_tiles, ctx.light = tiles, mid_tiles, light
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img)
        ctx.mark_non_differentiable(radii)
        # autograd otherwise hands backward() a zero tensor for the int32 `radii` output on every step: a P-element fill kernel (5 us at
        # 500 k Gaussians) in front of every backward pass
        ctx.set_materialize_grads(False)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii):
        rs = ctx.raster_settings
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img = ctx.saved_tensors
        if grad_out_color is None:                           # the image did not take part in the loss (gradients not materialised, see forward)
            grad_out_color = torch.zeros((3, int(rs.image_height), int(rs.image_width)), dtype=torch.float32, device=means3D.device)
        # native argument order (__init__.py:109-129)
        args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
                rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, sh, rs.sh_degree, rs.campos, geom, ctx.num_rendered,
                binning, img, rs.debug)
        # the per-pixel backward visits the tiles with instances only: their number is known exactly from the forward's read-back
        # (and the 1024-thread kernel only the ones with >= 128 instances; the rest take the 256-thread one)
        bound = ctx.nonempty_tiles if ctx.nonempty_tiles > 0 else 0
        mid = max(ctx.mid_tiles, 1) if (bound > 0 and ctx.mid_tiles >= 0) else 0
        # dL_dcolors / dL_dcov3D go to inputs that were None in the forward (empty tensors here, __init__.py:137-152): autograd drops them, so the
        # native backward neither allocates nor writes them (36 of the ~300 B per Gaussian its per-Gaussian kernel stores; round 6)
        need_col, need_cov = colors_precomp.numel() != 0 or _KEEP_ALL_OUTPUTS, cov3Ds_precomp.numel() != 0 or _KEEP_ALL_OUTPUTS
        (g_means2D, g_colors, g_opac, g_means3D, g_cov3D, g_sh, g_scales, g_rots) = _call_native(
            lambda *a: _C.rasterize_gaussians_backward(*a, tile_bound=bound, mid_bound=mid, light_tiles=ctx.light, need_colors=need_col, need_cov3D=need_cov),
            args, rs.debug, "snapshot_bw.dump", "backward")
        # forward-argument order (__init__.py:143-153)
        return (g_means3D, g_means2D, g_sh if sh.numel() != 0 else None, g_colors if need_col else None, g_opac, g_scales if scales.numel() != 0 else None,
                g_rots if rotations.numel() != 0 else None, g_cov3D if need_cov else None, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """bool per Gaussian: view-space z > 0.2 for this camera (__init__.py:176-185)."""
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        has_sr = scales is not None and rotations is not None
        any_sr = scales is not None or rotations is not None
        if (not has_sr and cov3D_precomp is None) or (any_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        absent = torch.Tensor([])       # empty CPU tensor = "not provided" (__init__.py:197-207)
        return rasterize_gaussians(
            means3D, means2D,
            absent if shs is None else shs,
            absent if colors_precomp is None else colors_precomp,
            opacities,
            absent if scales is None else scales,
            absent if rotations is None else rotations,
            absent if cov3D_precomp is None else cov3D_precomp,
            rs)
