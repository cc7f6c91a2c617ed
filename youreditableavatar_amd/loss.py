"""The trainers' photometric loss on the MI355X ("next" row 2): drop-in for ``utils/loss_utils.py`` of the reference.

``l1_loss`` / ``ssim`` keep the reference's names and argument meaning (loss_utils.py:17-18, :39-63);
``l1_ssim_loss`` is the composition every trainer builds from them (tetgs_texture/refine.py:245-247).  Value and
gradient come from two passes over the image (csrc/tgs_loss.hip) instead of five depthwise convolutions, ~15
element-wise kernels and their autograd graph.  HIP device only: there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import torch

from .diff_gaussian_rasterization import _C as _rast_c

_lib = _rast_c._lib
_lib.tgs_l1_ssim_workspace_bytes.restype = C.c_size_t
_lib.tgs_l1_ssim_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
_lib.tgs_l1_ssim.restype = C.c_int
_lib.tgs_l1_ssim.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]


def _check(img: torch.Tensor, gt: torch.Tensor) -> Tuple[int, int, int]:
    if not img.is_cuda or not gt.is_cuda:
        raise RuntimeError("youreditableavatar_amd.loss has no CPU path: images must be on a HIP device")
    if img.shape != gt.shape or img.dim() not in (3, 4):
        raise RuntimeError(f"expected two images of the same [C,H,W] or [B,C,H,W] shape, got {tuple(img.shape)} and {tuple(gt.shape)}")
    if img.dtype != torch.float32 or gt.dtype != torch.float32:
        raise RuntimeError("expected float32 images")
    planes = int(img.shape[0]) if img.dim() == 3 else int(img.shape[0] * img.shape[1])
    return planes, int(img.shape[-2]), int(img.shape[-1])


def l1_ssim_value_and_grad(img: torch.Tensor, gt: torch.Tensor, dssim_factor: float = 0.2, need_grad: bool = True):
    """-> (out3, grad): out3 = device tensor [loss, ssim, l1]; grad = d loss / d img (None unless ``need_grad``).
    Nothing is synchronised; use this directly as the ``upstream`` of ``multiview.SyncFreeBatch``."""
    planes, H, W = _check(img, gt)
    dev = img.device
    a, b = img.detach().contiguous(), gt.detach().contiguous()
    with torch.cuda.device(dev):
        nbytes = int(_lib.tgs_l1_ssim_workspace_bytes(planes, H, W))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        out3 = torch.empty(3, dtype=torch.float32, device=dev)
        grad = torch.empty_like(a) if need_grad else None
        r = _lib.tgs_l1_ssim(torch.cuda.current_stream(dev).cuda_stream, planes, H, W, a.data_ptr(), b.data_ptr(), float(dssim_factor),
                             out3.data_ptr(), grad.data_ptr() if need_grad else None, ws.data_ptr(), nbytes)
    if r < 0:
        raise _rast_c._err(r)
    return out3, grad


_lib.tgs_l1_ssim_backward.restype = C.c_int
_lib.tgs_l1_ssim_backward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]


class _L1SSIM(torch.autograd.Function):
    """Forward: the statistics pass + the reduction (value only); the gradient pass runs in backward with the incoming gradient as a device
    scalar folded in (tgs_l1_ssim_backward) -- no image-sized ``grad * g`` pass, and no gradient image is computed for a loss that is never
    back-propagated."""

    @staticmethod
    def forward(ctx, img, gt, dssim_factor, which):
        planes, H, W = _check(img, gt)
        dev = img.device
        a, b = img.detach().contiguous(), gt.detach().contiguous()
        with torch.cuda.device(dev):
            nbytes = int(_lib.tgs_l1_ssim_workspace_bytes(planes, H, W))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            out3 = torch.empty(3, dtype=torch.float32, device=dev)
            r = _lib.tgs_l1_ssim(torch.cuda.current_stream(dev).cuda_stream, planes, H, W, a.data_ptr(), b.data_ptr(), float(dssim_factor),
                                 out3.data_ptr(), None, ws.data_ptr(), nbytes)
        if r < 0:
            raise _rast_c._err(r)
        ctx.save_for_backward(a, b, ws)
        ctx.dims, ctx.f, ctx.shape = (planes, H, W), float(dssim_factor), img.shape
        return out3[which]

    @staticmethod
    def backward(ctx, g):
        a, b, ws = ctx.saved_tensors
        planes, H, W = ctx.dims
        dev = a.device
        g = g.detach().to(device=dev, dtype=torch.float32).contiguous()
        grad = torch.empty_like(a)
        with torch.cuda.device(dev):
            r = _lib.tgs_l1_ssim_backward(torch.cuda.current_stream(dev).cuda_stream, planes, H, W, a.data_ptr(), b.data_ptr(), ctx.f, g.data_ptr(),
                                          grad.data_ptr(), ws.data_ptr(), ws.numel())
        if r < 0:
            raise _rast_c._err(r)
        return grad.view(ctx.shape), None, None, None


def l1_ssim_loss(network_output: torch.Tensor, gt: torch.Tensor, dssim_factor: float = 0.2) -> torch.Tensor:
    """``(1 - dssim_factor) * l1_loss(x, gt) + dssim_factor * (1 - ssim(x, gt))`` (refine.py:245-247), differentiable in x."""
    return _L1SSIM.apply(network_output, gt, float(dssim_factor), 0)


def ssim(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11, size_average: bool = True) -> torch.Tensor:
    """loss_utils.py:39-48 (window 11, mean over everything), differentiable in ``img1``."""
    if window_size != 11 or not size_average:
        raise NotImplementedError("the fused kernel implements the configuration the reference uses: window_size=11, size_average=True")
    return 1.0 - _L1SSIM.apply(img1, img2, 1.0, 0)


def l1_loss(network_output: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """loss_utils.py:17-18."""
    return _L1SSIM.apply(network_output, gt, 0.0, 0)
