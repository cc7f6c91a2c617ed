"""``simple_knn._C`` served by libtgs_raster.so (tgs_dist2 in include/tgs_raster.h)."""
import ctypes as C

import torch

from ..diff_gaussian_rasterization import _C as _rast_c

_lib = _rast_c._lib
_lib.tgs_dist2_workspace_bytes.restype = C.c_size_t
_lib.tgs_dist2_workspace_bytes.argtypes = [C.c_int]
_lib.tgs_dist2.restype = C.c_int
_lib.tgs_dist2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """points[P,3] (fp32, HIP device) -> float[P]: mean squared distance to the 3 nearest other points
    (simple-knn/spatial.cu:15-26)."""
    if not points.is_cuda:
        raise RuntimeError("simple_knn (MI355X build) has no CPU path: points must be on a HIP device")
    if points.dtype != torch.float32:
        raise RuntimeError(f"expected scalar type Float but found {points.dtype}")
    pts = points.contiguous()
    P = int(pts.size(0))
    dev = pts.device
    means = torch.full((P,), 0.0, dtype=torch.float32, device=dev)
    if P == 0:
        return means
    with torch.cuda.device(dev):
        nbytes = int(_lib.tgs_dist2_workspace_bytes(P))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        r = _lib.tgs_dist2(torch.cuda.current_stream(dev).cuda_stream, P, pts.data_ptr(), means.data_ptr(), ws.data_ptr(), nbytes)
    if r < 0:
        raise _rast_c._err(r)
    return means
