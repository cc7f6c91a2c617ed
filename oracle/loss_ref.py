"""TEST INFRASTRUCTURE ONLY -- torch restatement of the reference's photometric loss (never imported by the product).

Pinned: tests/golden/ref_loss_fixture.npz holds outputs and autograd gradients of the reference's own
`utils/loss_utils.py` (l1_loss :17-18, ssim :39-63) imported in the build container (tests/make_ref_loss_fixture.py).

    loss = (1 - dssim_factor) * l1_loss(pred, gt) + dssim_factor * (1 - ssim(pred, gt))
(Edit_core/tetgs_texture/refine.py:245-247, refine_3dgs.py:277-279, paint_2dgs.py:345-347; dssim_factor = 0.2).
"""
import math

import torch
import torch.nn.functional as F


def gaussian_window(window_size: int = 11, sigma: float = 1.5, dtype=torch.float32) -> torch.Tensor:
    """loss_utils.py:23-25: normalised 1-D Gaussian (computed in Python doubles, stored as fp32 like torch.Tensor([...]))."""
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)], dtype=torch.float32)
    return (g / g.sum()).to(dtype)


def ssim_map(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11) -> torch.Tensor:
    """loss_utils.py:27-58 for [C,H,W] or [B,C,H,W] images: zero-padded depthwise 11x11 Gaussian statistics."""
    C = img1.size(-3)
    w1 = gaussian_window(window_size, 1.5, torch.float32).unsqueeze(1)
    w2 = w1.mm(w1.t()).float().to(img1.dtype)                 # the reference forms the 2-D window in fp32 (:29)
    window = w2.unsqueeze(0).unsqueeze(0).expand(C, 1, window_size, window_size).contiguous().to(img1.device)
    conv = lambda x: F.conv2d(x, window, padding=window_size // 2, groups=C)
    mu1, mu2 = conv(img1), conv(img2)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = conv(img1 * img1) - mu1_sq
    sigma2_sq = conv(img2 * img2) - mu2_sq
    sigma12 = conv(img1 * img2) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))


def ssim(img1, img2, window_size: int = 11):
    return ssim_map(img1, img2, window_size).mean()


def l1_loss(a, b):
    return torch.abs(a - b).mean()


def l1_ssim_loss(pred, gt, dssim_factor: float = 0.2):
    return (1.0 - dssim_factor) * l1_loss(pred, gt) + dssim_factor * (1.0 - ssim(pred, gt))
