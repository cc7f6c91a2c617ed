"""Fused SH -> RGB of the Gaussian colours outside the rasterizer ("next" row 1).

Drop-in for the body of ``TetGS.get_points_rgb`` (Edit_core/tetgs_scene/tetgs_model.py:413-442), which every
training step of the reference runs as ~25 element-wise PyTorch kernels plus their autograd:

    render_directions = F.normalize(positions - camera_centers, dim=-1)          # or the given directions
    shs_view = sh_coordinates[:, :sh_levels**2].transpose(-1, -2).view(-1, 3, sh_levels**2)
    colors = torch.clamp_min(eval_sh(sh_levels - 1, shs_view, render_directions) + 0.5, 0.0).view(-1, 3)

One HIP kernel forward, one backward (tgs_sh_rgb_forward / tgs_sh_rgb_backward in include/tgs_raster.h).
HIP tensors only; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from .diff_gaussian_rasterization import _C as _rast_c

_lib = _rast_c._lib
_lib.tgs_sh_rgb_forward.restype = C.c_int
_lib.tgs_sh_rgb_forward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5
_lib.tgs_sh_rgb_backward.restype = C.c_int
_lib.tgs_sh_rgb_backward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 8
_lib.tgs_sh_rgb_dcrest_forward.restype = C.c_int
_lib.tgs_sh_rgb_dcrest_forward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6
_lib.tgs_sh_rgb_dcrest_backward.restype = C.c_int
_lib.tgs_sh_rgb_dcrest_backward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 10


def _check(t: torch.Tensor, name: str, dev) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise RuntimeError(f"expected scalar type Float but found {t.dtype} for {name}")
    if not t.is_cuda:
        raise RuntimeError(f"sh_color (MI355X build) has no CPU path: {name} must be on a HIP device")
    return t.to(dev).contiguous()


class _SHColor(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sh_coordinates, positions, camera_center, directions, sh_levels):
        dev = sh_coordinates.device
        sh = _check(sh_coordinates, "sh_coordinates", dev)
        P, M = int(sh.shape[0]), int(sh.shape[1])
        pos = _check(positions, "positions", dev) if positions is not None else None
        cam = _check(camera_center.reshape(-1), "camera_centers", dev) if camera_center is not None else None
        dirs = _check(directions, "directions", dev) if directions is not None else None
        colors = torch.empty((P, 3), dtype=torch.float32, device=dev)
        p = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.device(dev):
            r = _lib.tgs_sh_rgb_forward(torch.cuda.current_stream(dev).cuda_stream, P, M, int(sh_levels), p(sh), p(pos), p(cam), p(dirs), colors.data_ptr())
        if r < 0:
            raise _rast_c._err(r)
        ctx.save_for_backward(sh, pos if pos is not None else torch.Tensor([]), cam if cam is not None else torch.Tensor([]),
                              dirs if dirs is not None else torch.Tensor([]))
        ctx.levels = int(sh_levels)
        return colors

    @staticmethod
    def backward(ctx, grad_colors):
        sh, pos, cam, dirs = ctx.saved_tensors
        dev = sh.device
        P, M = int(sh.shape[0]), int(sh.shape[1])
        pos = pos if pos.numel() else None
        cam = cam if cam.numel() else None
        dirs = dirs if dirs.numel() else None
        g = _check(grad_colors, "grad_colors", dev)
        d_sh = torch.empty_like(sh)
        d_pos = torch.empty_like(pos) if pos is not None else None
        d_dir = torch.empty_like(dirs) if dirs is not None else None
        p = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.device(dev):
            r = _lib.tgs_sh_rgb_backward(torch.cuda.current_stream(dev).cuda_stream, P, M, ctx.levels, p(sh), p(pos), p(cam), p(dirs), g.data_ptr(),
                                         d_sh.data_ptr(), p(d_pos), p(d_dir))
        if r < 0:
            raise _rast_c._err(r)
        return d_sh, d_pos, None, d_dir, None


class _SHColorDcRest(torch.autograd.Function):
    """colours from the two parameter tensors the reference keeps (tetgs_model.py:234-239), no ``torch.cat`` (tgs_sh_rgb_dcrest_*)"""

    @staticmethod
    def forward(ctx, sh_dc, sh_rest, positions, camera_center, directions, sh_levels):
        dev = sh_dc.device
        dc = _check(sh_dc, "sh_coordinates_dc", dev)
        if dc.dim() != 3 or dc.shape[1] != 1 or dc.shape[2] != 3:
            raise RuntimeError("sh_coordinates_dc must have dimensions (num_points, 1, 3)")
        P = int(dc.shape[0])
        levels = int(sh_levels)
        rest = _check(sh_rest, "sh_coordinates_rest", dev) if (sh_rest is not None and sh_rest.numel()) else None
        Mr = int(rest.shape[1]) if rest is not None else 0
        if rest is not None and (rest.dim() != 3 or rest.shape[0] != P or rest.shape[2] != 3):
            raise RuntimeError("sh_coordinates_rest must have dimensions (num_points, M - 1, 3)")
        pos = _check(positions, "positions", dev) if positions is not None else None
        cam = _check(camera_center.reshape(-1), "camera_centers", dev) if camera_center is not None else None
        dirs = _check(directions, "directions", dev) if directions is not None else None
        colors = torch.empty((P, 3), dtype=torch.float32, device=dev)
        p = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.device(dev):
            r = _lib.tgs_sh_rgb_dcrest_forward(torch.cuda.current_stream(dev).cuda_stream, P, Mr, levels, p(dc), p(rest), p(pos), p(cam), p(dirs), colors.data_ptr())
        if r < 0:
            raise _rast_c._err(r)
        e = torch.Tensor([])
        ctx.save_for_backward(dc, rest if rest is not None else e, pos if pos is not None else e, cam if cam is not None else e, dirs if dirs is not None else e)
        ctx.levels = levels
        return colors

    @staticmethod
    def backward(ctx, grad_colors):
        dc, rest, pos, cam, dirs = ctx.saved_tensors
        dev = dc.device
        P = int(dc.shape[0])
        rest = rest if rest.numel() else None
        pos = pos if pos.numel() else None
        cam = cam if cam.numel() else None
        dirs = dirs if dirs.numel() else None
        Mr = int(rest.shape[1]) if rest is not None else 0
        g = _check(grad_colors, "grad_colors", dev)
        d_dc = torch.empty_like(dc)
        # levels == 1: the rest rows get exactly zero gradient -- returned as None (no 180 B of zeros per Gaussian written, read by the optimiser ...)
        d_rest = torch.empty_like(rest) if (rest is not None and ctx.levels > 1) else None
        d_pos = torch.empty_like(pos) if pos is not None else None
        d_dir = torch.empty_like(dirs) if dirs is not None else None
        p = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.device(dev):
            r = _lib.tgs_sh_rgb_dcrest_backward(torch.cuda.current_stream(dev).cuda_stream, P, Mr, ctx.levels, p(dc), p(rest), p(pos), p(cam), p(dirs), g.data_ptr(),
                                                d_dc.data_ptr(), p(d_rest), p(d_pos), p(d_dir))
        if r < 0:
            raise _rast_c._err(r)
        return d_dc, d_rest, d_pos, None, d_dir, None


def points_rgb_dc_rest(sh_coordinates_dc: torch.Tensor, sh_coordinates_rest: Optional[torch.Tensor], sh_levels: int,
                       positions: Optional[torch.Tensor] = None, camera_centers: Optional[torch.Tensor] = None,
                       directions: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``points_rgb`` on the model's own two parameters: ``_sh_coordinates_dc`` [P,1,3] and ``_sh_coordinates_rest`` [P,M-1,3] (or None for
    a one-level model) instead of ``self.sh_coordinates`` -- which is ``torch.cat([dc, rest], dim=1)`` on every access
    (tetgs_model.py:268-272): a 192-B read + 192-B write per Gaussian in front of every render and the split of its gradient behind it.
    Gradients arrive on the two parameters directly; with ``sh_levels == 1`` the rest parameter gets none (``.grad`` stays ``None`` -- the
    reference hands it a tensor of zeros there)."""
    if sh_levels > 1 and (sh_coordinates_rest is None or sh_coordinates_rest.shape[1] < sh_levels ** 2 - 1):
        raise ValueError("sh_coordinates_rest must hold at least sh_levels**2 - 1 coefficient rows")
    if camera_centers is not None:
        if positions is None:
            raise ValueError("positions are required with camera_centers")
        if camera_centers.numel() != 3:
            raise ValueError("one camera centre ([3] or [1,3]) is supported")
        return _SHColorDcRest.apply(sh_coordinates_dc, sh_coordinates_rest, positions, camera_centers, None, sh_levels)
    if directions is not None:
        return _SHColorDcRest.apply(sh_coordinates_dc, sh_coordinates_rest, None, None, directions, sh_levels)
    raise ValueError("Either camera_centers or directions must be provided.")


def points_rgb(sh_coordinates: torch.Tensor, sh_levels: int, positions: Optional[torch.Tensor] = None,
               camera_centers: Optional[torch.Tensor] = None, directions: Optional[torch.Tensor] = None) -> torch.Tensor:
    """colors[P,3] = clamp_min(eval_sh(sh_levels-1, sh_coordinates[:, :sh_levels**2], dirs) + 0.5, 0).

    ``sh_coordinates`` is the [P, M, 3] tensor the reference keeps (``TetGS.sh_coordinates``); give either
    ``positions`` + ``camera_centers`` ([3] or [1,3]) or unit ``directions`` -- get_points_rgb's two modes
    (tetgs_model.py:424-429; neither -> ValueError like the reference)."""
    if camera_centers is not None:
        if positions is None:
            raise ValueError("positions are required with camera_centers")
        if camera_centers.numel() != 3:
            raise ValueError("one camera centre ([3] or [1,3]) is supported")
        return _SHColor.apply(sh_coordinates, positions, camera_centers, None, sh_levels)
    if directions is not None:
        return _SHColor.apply(sh_coordinates, None, None, directions, sh_levels)
    raise ValueError("Either camera_centers or directions must be provided.")


class _SHColorGroups(torch.autograd.Function):
    """colours of the keep and the edit group into ONE [Pk+Pe,3] tensor (two launches of the dc / rest kernel, no torch.cat); gradients for
    the edit group only"""

    @staticmethod
    def forward(ctx, keep_dc, keep_rest, keep_pos, edit_dc, edit_rest, edit_pos, camera_center, keep_levels, edit_levels):
        dev = edit_dc.device
        cam = _check(camera_center.reshape(-1), "camera_centers", dev)
        p = lambda t: None if t is None else t.data_ptr()
        groups = []
        for dc, rest, pos, levels, tag in ((keep_dc, keep_rest, keep_pos, int(keep_levels), "keep"), (edit_dc, edit_rest, edit_pos, int(edit_levels), "edit")):
            dc = _check(dc, f"{tag}_sh_coordinates_dc", dev)
            if dc.dim() != 3 or dc.shape[1] != 1 or dc.shape[2] != 3:
                raise RuntimeError(f"{tag}_sh_coordinates_dc must have dimensions (num_points, 1, 3)")
            n = int(dc.shape[0])
            rest = _check(rest, f"{tag}_sh_coordinates_rest", dev) if (rest is not None and rest.numel() and levels > 1) else None
            if rest is not None and (rest.dim() != 3 or rest.shape[0] != n or rest.shape[2] != 3):
                raise RuntimeError(f"{tag}_sh_coordinates_rest must have dimensions (num_points, M - 1, 3)")
            pos = _check(pos, f"{tag} positions", dev)
            if pos.shape[0] != n or pos.dim() != 2 or pos.shape[1] not in (1, 3):
                raise RuntimeError(f"{tag} positions must have dimensions (num_points, 3) -- or (num_points, 1): tetgs_edit_3d.py:556-572")
            # Edit3DTetGS hands get_points_rgb its [Pe,1] normal offsets as `positions` (tetgs_edit_3d.py:556,572): `positions - camera_centers`
            # broadcasts them to [Pe,3].  Reproduced, not "fixed" (SURVEY.md section 8f).
            groups.append((dc, rest, pos.expand(n, 3).contiguous() if pos.shape[1] == 1 else pos, levels, n, int(pos.shape[1])))
        (kdc, krest, kpos, klev, Pk, _), (edc, erest, epos, elev, Pe, ecols) = groups
        colors = torch.empty((Pk + Pe, 3), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            r = _lib.tgs_sh_rgb_dcrest_forward(st, Pk, int(krest.shape[1]) if krest is not None else 0, klev, p(kdc), p(krest), p(kpos), p(cam), None, colors.data_ptr())
            if r >= 0:
                r = _lib.tgs_sh_rgb_dcrest_forward(st, Pe, int(erest.shape[1]) if erest is not None else 0, elev, p(edc), p(erest), p(epos), p(cam), None,
                                                   colors.data_ptr() + 12 * Pk)
        if r < 0:
            raise _rast_c._err(r)
        e = torch.Tensor([])
        ctx.save_for_backward(edc, erest if erest is not None else e, epos, cam)
        ctx.info = (Pk, Pe, elev, ecols)
        return colors

    @staticmethod
    def backward(ctx, grad_colors):
        edc, erest, epos, cam = ctx.saved_tensors
        Pk, Pe, elev, ecols = ctx.info
        dev = edc.device
        erest = erest if erest.numel() else None
        g = _check(grad_colors, "grad_colors", dev)
        need = ctx.needs_input_grad             # (keep_dc, keep_rest, keep_pos, edit_dc, edit_rest, edit_pos, cam, ., .)
        d_dc = torch.empty_like(edc)
        d_rest = torch.empty_like(erest) if erest is not None else None
        d_pos = torch.empty_like(epos) if need[5] else None
        p = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.device(dev):
            r = _lib.tgs_sh_rgb_dcrest_backward(torch.cuda.current_stream(dev).cuda_stream, Pe, int(erest.shape[1]) if erest is not None else 0, elev, p(edc), p(erest),
                                                p(epos), p(cam), None, g.data_ptr() + 12 * Pk, d_dc.data_ptr(), p(d_rest), p(d_pos), None)
        if r < 0:
            raise _rast_c._err(r)
        if d_pos is not None and ecols == 1:
            d_pos = d_pos.sum(dim=1, keepdim=True)      # the adjoint of the [Pe,1] -> [Pe,3] broadcast
        return None, None, None, d_dc, d_rest, d_pos, None, None, None


def points_rgb_groups(*, keep_sh_dc: torch.Tensor, keep_sh_rest: Optional[torch.Tensor], keep_sh_levels: int, keep_positions: torch.Tensor,
                      edit_sh_dc: torch.Tensor, edit_sh_rest: Optional[torch.Tensor], edit_sh_levels: int, edit_positions: torch.Tensor,
                      camera_centers: torch.Tensor) -> torch.Tensor:
    """``splat_colors`` of the editing stages' render call (tetgs_edit_2d.py:548-565, tetgs_edit_3d.py:566-582): ``get_points_rgb`` of the keep
    group at ``keep_sh_levels`` and of the edit group at ``edit_sh_levels`` (each from its dc / rest parameters), written into one
    [Pk+Pe,3] tensor, keep rows first -- without the two ``torch.cat([dc, rest])``, the ``torch.cat`` of the colours and the keep group's
    backward.  Gradients reach the edit group's dc / rest (and positions, when they require one); the keep group is frozen in the
    reference.  ``edit_positions`` may be [Pe,1] as ``Edit3DTetGS`` passes it (the offsets along the normals, broadcast against the camera
    centre -- the reference's behaviour, kept).  With ``edit_sh_levels == 1`` the edit rest parameter gets no gradient (``None``)."""
    for name, t in (("keep_sh_dc", keep_sh_dc), ("keep_sh_rest", keep_sh_rest), ("keep_positions", keep_positions)):
        if t is not None and t.requires_grad:
            raise RuntimeError(f"points_rgb_groups: {name} requires a gradient, but the keep group is frozen (tetgs_edit_2d.py:237-262)")
    for lev, rest, tag in ((keep_sh_levels, keep_sh_rest, "keep"), (edit_sh_levels, edit_sh_rest, "edit")):
        if lev > 1 and (rest is None or rest.shape[1] < lev ** 2 - 1):
            raise ValueError(f"{tag}_sh_rest must hold at least {tag}_sh_levels**2 - 1 coefficient rows")
    if camera_centers is None or camera_centers.numel() != 3:
        raise ValueError("one camera centre ([3] or [1,3]) is required")
    return _SHColorGroups.apply(keep_sh_dc, keep_sh_rest, keep_positions, edit_sh_dc, edit_sh_rest, edit_positions, camera_centers, keep_sh_levels, edit_sh_levels)
