"""Caller-side 3D covariance (TEST INFRASTRUCTURE): what tetgs_model.py:559-577 computes in PyTorch
when compute_covariance_in_rasterizer=False, in float32 numpy."""
import numpy as np


def cov3d_from(scales, rots, mod=1.0):
    r, x, y, z = rots[:, 0], rots[:, 1], rots[:, 2], rots[:, 3]
    Rm = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
                   np.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
                   np.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], -2).astype(np.float32)
    S = (mod * scales).astype(np.float32)
    RS = Rm * S[:, None, :]
    Sig = RS @ RS.transpose(0, 2, 1)
    return np.stack([Sig[:, 0, 0], Sig[:, 0, 1], Sig[:, 0, 2], Sig[:, 1, 1], Sig[:, 1, 2], Sig[:, 2, 2]], -1).astype(np.float32)
