"""``knn_points`` for the one way the reference uses it: the K nearest neighbours of a point cloud among itself.

``tetgs_scene/tetgs_model.py`` calls ``pytorch3d.ops.knn_points(points[None], points[None], K=4)`` to initialise the Gaussian
scales (:36-49, ``.dists[..., 1:]`` drops the point itself) and ``K=knn_to_track`` (16) to keep neighbour indices for its
regularisers (:180-182).  ``knn_points_self`` returns the same ``dists`` (squared, ascending, the point itself first at 0) and
``idx`` (int64) tensors from ``tgs_knn_self`` (csrc/tgs_knn.hip).  HIP device only.
"""
from __future__ import annotations

import ctypes as C
from collections import namedtuple

import torch

from .diff_gaussian_rasterization import _C as _rast_c

_lib = _rast_c._lib
_lib.tgs_dist2_workspace_bytes.restype = C.c_size_t
_lib.tgs_dist2_workspace_bytes.argtypes = [C.c_int]
_lib.tgs_knn_self.restype = C.c_int
_lib.tgs_knn_self.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]

KNN = namedtuple("KNN", "dists idx knn")        # pytorch3d.ops.knn._KNN


def knn_points_self(points: torch.Tensor, K: int = 1) -> KNN:
    """``points`` [N,P,3] or [P,3] float32 on the HIP device -> KNN(dists [N,P,K], idx [N,P,K], knn=None), neighbours taken from
    the same cloud (``knn_points(p, p, K=K)``)."""
    if not points.is_cuda:
        raise RuntimeError("youreditableavatar_amd.knn has no CPU path: points must be on a HIP device")
    if points.dtype != torch.float32 or points.shape[-1] != 3 or points.dim() not in (2, 3):
        raise RuntimeError(f"expected float32 points of shape [N,P,3] or [P,3], got {points.dtype} {tuple(points.shape)}")
    if not (0 < K <= 32):
        raise RuntimeError("K must be in 1..32")
    batched = points.dim() == 3
    clouds = points if batched else points[None]
    N, P = int(clouds.shape[0]), int(clouds.shape[1])
    dev = points.device
    dists = torch.empty((N, P, K), dtype=torch.float32, device=dev)
    idx = torch.empty((N, P, K), dtype=torch.int64, device=dev)
    if P:
        with torch.cuda.device(dev):
            nbytes = int(_lib.tgs_dist2_workspace_bytes(P))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            st = torch.cuda.current_stream(dev).cuda_stream
            for n in range(N):
                pts = clouds[n].detach().contiguous()
                r = _lib.tgs_knn_self(st, P, int(K), pts.data_ptr(), dists[n].data_ptr(), idx[n].data_ptr(), ws.data_ptr(), nbytes)
                if r < 0:
                    raise _rast_c._err(r)
    return KNN(dists if batched else dists[0], idx if batched else idx[0], None)
