"""The binding side of the rasterizer call, fused (BASELINE.json north_star: "tetgs_scene Gaussian model bindings").

Before every ``GaussianRasterizer`` call the reference's model classes turn their raw parameters into the rasterizer's inputs through four
properties, each a PyTorch kernel forward and one or two backward (Edit_core/tetgs_scene/tetgs_model.py):

    strengths   (:261-265)  torch.sigmoid(self.all_densities.view(-1, 1))
    scaling     (:279-281)  torch.exp(self._scales)                                    # scale_activation, :16
    quaternions (:283-286)  torch.nn.functional.normalize(self._quaternions, dim=-1)
    points      (:252-258)  self.ori_points + self.normals * self._points              # mesh-bound Gaussians, one learnable offset each

``gaussian_bind`` computes them in ONE HIP kernel and back-propagates through ONE (csrc/tgs_bind.hip, C ABI tgs_bind_forward / _backward).
HIP tensors only; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from .diff_gaussian_rasterization import _C as _rast_c

_lib = _rast_c._lib
_lib.tgs_bind_forward.restype = C.c_int
_lib.tgs_bind_forward.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 10
_lib.tgs_bind_backward.restype = C.c_int
_lib.tgs_bind_backward.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 12


def _f32(t: Optional[torch.Tensor], name: str, cols: int, P: int, dev) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise RuntimeError(f"expected scalar type Float but found {t.dtype} for {name}")
    if not t.is_cuda:
        raise RuntimeError(f"bindings (MI355X build) has no CPU path: {name} must be on a HIP device")
    if t.numel() != P * cols:
        raise RuntimeError(f"{name} must have {cols} value(s) per Gaussian ({P} Gaussians)")
    return t.to(dev).contiguous()


def _p(t):
    return None if t is None else t.data_ptr()


class _Bind(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw_density, raw_scales, raw_quats, ori_points, normals, deltas):
        first = next(t for t in (raw_density, raw_scales, raw_quats, ori_points) if t is not None)
        dev, P = first.device, int(first.shape[0])
        d, s, q = _f32(raw_density, "all_densities", 1, P, dev), _f32(raw_scales, "_scales", 3, P, dev), _f32(raw_quats, "_quaternions", 4, P, dev)
        o, n, dl = _f32(ori_points, "ori_points", 3, P, dev), _f32(normals, "normals", 3, P, dev), _f32(deltas, "_points", 1, P, dev)
        if (o is None) != (n is None) or (o is None) != (dl is None):
            raise RuntimeError("ori_points, normals and the offsets (_points) come together")
        new = lambda cols, like: torch.empty((P, cols), dtype=torch.float32, device=dev) if like is not None else None
        opacity, scales, quats, points = new(1, d), new(3, s), new(4, q), new(3, o)
        with torch.cuda.device(dev):
            r = _lib.tgs_bind_forward(torch.cuda.current_stream(dev).cuda_stream, P, _p(d), _p(s), _p(q), _p(o), _p(n), _p(dl), _p(opacity), _p(scales), _p(quats),
                                      _p(points))
        if r < 0:
            raise _rast_c._err(r)
        e = torch.Tensor([])
        ctx.save_for_backward(q if q is not None else e, n if n is not None else e, opacity if opacity is not None else e, scales if scales is not None else e)
        ctx.P = P
        ctx.shapes = tuple(None if t is None else tuple(t.shape) for t in (raw_density, raw_scales, raw_quats, deltas))
        return opacity, scales, quats, points

    @staticmethod
    def backward(ctx, g_opacity, g_scales, g_quats, g_points):
        q, n, opacity, scales = (t if t.numel() else None for t in ctx.saved_tensors)
        P = ctx.P
        dev = next(t for t in (q, n, opacity, scales) if t is not None).device
        need = ctx.needs_input_grad            # (raw_density, raw_scales, raw_quats, ori_points, normals, deltas)
        c = lambda g: None if g is None else g.contiguous()
        g_opacity, g_scales, g_quats, g_points = c(g_opacity), c(g_scales), c(g_quats), c(g_points)
        mk = lambda on, cols, g: torch.empty((P, cols), dtype=torch.float32, device=dev) if (on and g is not None) else None
        d_d, d_s, d_q, d_dl = mk(need[0], 1, g_opacity), mk(need[1], 3, g_scales), mk(need[2], 4, g_quats), mk(need[5], 1, g_points)
        if any(t is not None for t in (d_d, d_s, d_q, d_dl)):
            with torch.cuda.device(dev):
                r = _lib.tgs_bind_backward(torch.cuda.current_stream(dev).cuda_stream, P, _p(q), _p(n), _p(opacity), _p(scales), _p(g_opacity), _p(g_scales),
                                           _p(g_quats), _p(g_points), _p(d_d), _p(d_s), _p(d_q), _p(d_dl))
            if r < 0:
                raise _rast_c._err(r)
        sh = ctx.shapes
        v = lambda t, shape: None if t is None else t.view(shape)
        # ori_points / normals are buffers of the reference's models (requires_grad=False, tetgs_model.py:158-168): their gradients are not formed
        return v(d_d, sh[0]), v(d_s, sh[1]), v(d_q, sh[2]), None, None, v(d_dl, sh[3])


def gaussian_bind(all_densities: Optional[torch.Tensor] = None, raw_scales: Optional[torch.Tensor] = None, raw_quaternions: Optional[torch.Tensor] = None,
                  ori_points: Optional[torch.Tensor] = None, normals: Optional[torch.Tensor] = None, offsets: Optional[torch.Tensor] = None
                  ) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor], Optional[torch.Tensor], Optional[torch.Tensor]]:
    """(strengths [P,1], scaling [P,3], quaternions [P,4], points [P,3]) of the reference's model properties (module docstring) from the raw
    parameters, in one kernel; a group that is not given comes back as ``None``.  Differentiable with respect to ``all_densities``,
    ``raw_scales``, ``raw_quaternions`` and ``offsets`` (the model's ``_points``)."""
    if all(t is None for t in (all_densities, raw_scales, raw_quaternions, ori_points)):
        raise ValueError("gaussian_bind: nothing to do")
    return _Bind.apply(all_densities, raw_scales, raw_quaternions, ori_points, normals, offsets)


# ---- the two-group models of the editing stages -------------------------------------------------------------------------------------------
_lib.tgs_bind_groups_forward.restype = C.c_int
_lib.tgs_bind_groups_forward.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 15
_lib.tgs_bind_groups_backward.restype = C.c_int
_lib.tgs_bind_groups_backward.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 13


class _BindGroups(torch.autograd.Function):
    """inputs: keep (density, scales, quats, points) -- frozen --, edit (density, scales, quats, points | ori, normals, offsets)"""

    @staticmethod
    def forward(ctx, kd, ks, kq, kp, ed, es, eq, ep, eo, en, eoff):
        dev = kd.device
        Pk, Pe = int(kd.shape[0]), int(ed.shape[0])
        kd, ks, kq, kp = _f32(kd, "all_keep_densities", 1, Pk, dev), _f32(ks, "_keep_scales", 3, Pk, dev), _f32(kq, "_keep_quaternions", 4, Pk, dev), _f32(kp, "_keep_points", 3, Pk, dev)
        ed, es, eq = _f32(ed, "all_edit_densities", 1, Pe, dev), _f32(es, "_edit_scales", 3, Pe, dev), _f32(eq, "_edit_quaternions", 4, Pe, dev)
        ep = _f32(ep, "_edit_points", 3, Pe, dev)
        eo, en, eoff = _f32(eo, "ori_edit_points", 3, Pe, dev), _f32(en, "_edit_normals", 3, Pe, dev), _f32(eoff, "_edit_points (offsets)", 1, Pe, dev)
        P = Pk + Pe
        new = lambda cols: torch.empty((P, cols), dtype=torch.float32, device=dev)
        opacity, scales, quats, points = new(1), new(3), new(4), new(3)
        with torch.cuda.device(dev):
            r = _lib.tgs_bind_groups_forward(torch.cuda.current_stream(dev).cuda_stream, Pk, Pe, _p(kd), _p(ks), _p(kq), _p(kp), _p(ed), _p(es), _p(eq), _p(ep),
                                             _p(eo), _p(en), _p(eoff), _p(opacity), _p(scales), _p(quats), _p(points))
        if r < 0:
            raise _rast_c._err(r)
        e = torch.Tensor([])
        ctx.save_for_backward(eq, en if en is not None else e, opacity, scales)
        ctx.sizes = (Pk, Pe)
        return opacity, scales, quats, points

    @staticmethod
    def backward(ctx, g_opacity, g_scales, g_quats, g_points):
        eq, en, opacity, scales = ctx.saved_tensors
        en = en if en.numel() else None
        Pk, Pe = ctx.sizes
        dev = eq.device
        need = ctx.needs_input_grad            # (kd, ks, kq, kp, ed, es, eq, ep, eo, en, eoff)
        c = lambda g: None if g is None else g.contiguous()
        g_opacity, g_scales, g_quats, g_points = c(g_opacity), c(g_scales), c(g_quats), c(g_points)
        mk = lambda on, cols, g: torch.empty((Pe, cols), dtype=torch.float32, device=dev) if (on and g is not None) else None
        d_d, d_s, d_q = mk(need[4], 1, g_opacity), mk(need[5], 3, g_scales), mk(need[6], 4, g_quats)
        d_p, d_off = mk(need[7], 3, g_points), mk(need[10], 1, g_points)
        if any(t is not None for t in (d_d, d_s, d_q, d_p, d_off)):
            with torch.cuda.device(dev):
                r = _lib.tgs_bind_groups_backward(torch.cuda.current_stream(dev).cuda_stream, Pk, Pe, _p(eq), _p(en), _p(opacity), _p(scales), _p(g_opacity), _p(g_scales),
                                                  _p(g_quats), _p(g_points), _p(d_d), _p(d_s), _p(d_q), _p(d_p), _p(d_off))
            if r < 0:
                raise _rast_c._err(r)
        # the keep group is frozen (requires_grad=False, tetgs_edit_2d.py:237-262); ori_edit_points / _edit_normals are buffers (tetgs_edit_3d.py:121-136)
        return None, None, None, None, d_d, d_s, d_q, d_p, None, None, d_off


def gaussian_bind_groups(*, keep_points: torch.Tensor, keep_densities: torch.Tensor, keep_scales: torch.Tensor, keep_quaternions: torch.Tensor,
                         edit_densities: torch.Tensor, edit_scales: torch.Tensor, edit_quaternions: torch.Tensor,
                         edit_points: Optional[torch.Tensor] = None, ori_edit_points: Optional[torch.Tensor] = None,
                         edit_normals: Optional[torch.Tensor] = None, edit_offsets: Optional[torch.Tensor] = None
                         ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """(strengths [Pk+Pe,1], scaling [Pk+Pe,3], quaternions [Pk+Pe,4], points [Pk+Pe,3]) of ``EditTetGS`` (tetgs_edit_2d.py:280-318:
    ``edit_points`` [Pe,3]) and ``Edit3DTetGS`` (tetgs_edit_3d.py:272-331: ``ori_edit_points`` + ``edit_normals`` * ``edit_offsets`` [Pe,1]) --
    the properties those classes build with ``torch.cat([keep, edit])`` + activation on every access -- in one kernel, keep rows first.
    Differentiable with respect to the edit group only: the keep group is frozen in the reference (``requires_grad=False``); a keep
    tensor that requires a gradient is an error here rather than a silently missing gradient."""
    for name, t in (("keep_points", keep_points), ("keep_densities", keep_densities), ("keep_scales", keep_scales), ("keep_quaternions", keep_quaternions)):
        if t.requires_grad:
            raise RuntimeError(f"gaussian_bind_groups: {name} requires a gradient, but the keep group is frozen (tetgs_edit_2d.py:237-262); "
                               "bind a learnable group with gaussian_bind")
    plain, bound = edit_points is not None, any(t is not None for t in (ori_edit_points, edit_normals, edit_offsets))
    if plain == bound or (bound and any(t is None for t in (ori_edit_points, edit_normals, edit_offsets))):
        raise ValueError("gaussian_bind_groups: give either edit_points [Pe,3], or ori_edit_points + edit_normals + edit_offsets")
    return _BindGroups.apply(keep_densities, keep_scales, keep_quaternions, keep_points, edit_densities, edit_scales, edit_quaternions, edit_points,
                             ori_edit_points, edit_normals, edit_offsets)
