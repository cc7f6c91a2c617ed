"""TEST INFRASTRUCTURE ONLY -- CPU oracle for simple-knn's distCUDA2 (never imported by the product path).

PARITY UNPINNED: the reference holds no test or golden vector for simple-knn and its CUDA source (cub/thrust) cannot
be built in this image.  The oracle therefore restates WHAT the reference computes, which its algorithm determines
exactly: `boxMeanDist` (Edit_core/thirdparties/simple-knn/simple_knn.cu:147-183) scans every box it cannot reject by a
conservative AABB distance bound, skipping only the point itself (`i == idx`, :173), so its result is the exact mean of
the 3 smallest squared distances to the OTHER points (duplicates count with distance 0), with each distance evaluated
in fp32 as dx*dx+dy*dy+dz*dz (:137-138) and FLT_MAX standing in for missing neighbours (:155, P < 4).
"""
import numpy as np

FLT_MAX = np.float32(3.4028234663852886e38)


def _d2_f32(p, q):
    d = (q - p).astype(np.float32)
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]).astype(np.float32) + d[..., 2] * d[..., 2]


def dist2_bruteforce(points: np.ndarray) -> np.ndarray:
    """O(P^2) definition, chunked; use for P up to a few 10k."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    P = pts.shape[0]
    out = np.empty(P, np.float32)
    step = max(1, (1 << 24) // max(P, 1))
    for s in range(0, P, step):
        e = min(P, s + step)
        d = _d2_f32(pts[s:e, None, :], pts[None, :, :])                   # [chunk, P]
        d[np.arange(e - s), np.arange(s, e)] = FLT_MAX                    # i == idx is skipped
        if P < 4:
            d = np.concatenate([d, np.full((e - s, 3), FLT_MAX, np.float32)], 1)
        best = np.sort(np.partition(d, 2, axis=1)[:, :3], axis=1)
        with np.errstate(over="ignore"):
            out[s:e] = ((best[:, 0] + best[:, 1]).astype(np.float32) + best[:, 2]).astype(np.float32) / np.float32(3.0)
    return out


def dist2_kdtree(points: np.ndarray) -> np.ndarray:
    """Same quantity through scipy's cKDTree (candidate search in fp64, distances re-evaluated in fp32); P >= 4."""
    from scipy.spatial import cKDTree
    pts = np.ascontiguousarray(points, dtype=np.float32)
    P = pts.shape[0]
    assert P >= 4
    _, idx = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    d = np.sort(_d2_f32(pts[:, None, :], pts[idx]), axis=1)[:, 1:4]       # drop one zero: the point itself
    return ((d[:, 0] + d[:, 1]).astype(np.float32) + d[:, 2]).astype(np.float32) / np.float32(3.0)


def knn_self_bruteforce(points: np.ndarray, K: int):
    """pytorch3d.ops.knn_points(p[None], p[None], K) as used at tetgs_scene/tetgs_model.py:36,180 (not importable here: pytorch3d
    is absent): squared distances in fp32, ascending, the point itself included; -> (dists [P,K], idx [P,K]).  Ties may come in
    any order, so compare indices through the distances they give."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    P = pts.shape[0]
    d_out = np.full((P, K), FLT_MAX, np.float32)
    i_out = np.full((P, K), -1, np.int64)
    step = max(1, (1 << 24) // max(P, 1))
    for s in range(0, P, step):
        e = min(P, s + step)
        d = _d2_f32(pts[s:e, None, :], pts[None, :, :])
        k = min(K, P)
        part = np.argpartition(d, k - 1, axis=1)[:, :k]
        dd = np.take_along_axis(d, part, 1)
        order = np.argsort(dd, axis=1, kind="stable")
        d_out[s:e, :k] = np.take_along_axis(dd, order, 1)
        i_out[s:e, :k] = np.take_along_axis(part, order, 1)
    return d_out, i_out
