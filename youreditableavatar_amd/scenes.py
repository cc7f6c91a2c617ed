"""Synthetic Gaussian clouds and cameras for tests and bench.

Host-side numpy only, seeded with PCG64 so the CPU oracle and the HIP path see bit-identical
inputs on every machine.  The camera helpers restate how the reference's callers assemble the
rasterizer settings (Edit_core/tetgs_scene/tetgs_model.py:480-521 with
Edit_core/utils/graphics_utils.py:39-86): row-vector convention, i.e. ``viewmatrix`` is the
transposed world-to-camera matrix and ``projmatrix`` is ``viewmatrix @ P^T``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np

SH_C0 = 0.28209479177387814


@dataclass
class Camera:
    """Everything GaussianRasterizationSettings needs, as float32 numpy arrays."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    viewmatrix: np.ndarray   # [4,4], transposed w2c (tetgs_model.py:490-491)
    projmatrix: np.ndarray   # [4,4], viewmatrix @ proj^T (tetgs_model.py:501)
    campos: np.ndarray       # [3]
    bg: np.ndarray = field(default_factory=lambda: np.ones(3, np.float32))
    scale_modifier: float = 1.0


def projection_matrix(znear: float, zfar: float, fovx: float, fovy: float) -> np.ndarray:
    """graphics_utils.py:66-86 (getProjectionMatrix), float32 like torch.zeros(4, 4)."""
    tan_y = math.tan(fovy / 2)
    tan_x = math.tan(fovx / 2)
    top, right = tan_y * znear, tan_x * znear
    bottom, left = -top, -right
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def world_to_view(R: np.ndarray, t: np.ndarray) -> np.ndarray:
    """graphics_utils.py:39-51 (getWorld2View): Rt[:3,:3] = R^T, Rt[:3,3] = t."""
    Rt = np.zeros((4, 4), np.float32)
    Rt[:3, :3] = R.T
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    return Rt


def orbit_camera(width: int, height: int, azimuth_deg: float = 0.0, elevation_deg: float = 5.0,
                 radius: float = 3.0, fovy_deg: float = 45.0, znear: float = 1e-4, zfar: float = 100.0,
                 bg=(1.0, 1.0, 1.0), target=(0.0, 0.0, 0.0)) -> Camera:
    """Orbit camera looking at ``target`` (COLMAP axes: x right, y down, z forward), same family as
    sample_circle_gs_cameras (Edit_core/tetgs_scene/cameras.py:443-527)."""
    az, el = math.radians(azimuth_deg), math.radians(elevation_deg)
    C = np.array([radius * math.cos(el) * math.sin(az), radius * math.sin(el),
                  radius * math.cos(el) * math.cos(az)], np.float64) + np.asarray(target, np.float64)
    z = np.asarray(target, np.float64) - C
    z /= np.linalg.norm(z)
    down = np.array([0.0, -1.0, 0.0])
    x = np.cross(down, z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    c2w_R = np.stack([x, y, z], axis=1)            # columns = camera axes in world
    w2c_R = c2w_R.T
    w2c_t = -w2c_R @ C
    # tetgs_model.py:486-491: R = w2c[:3,:3].T, T = w2c[:3,3]; view = getWorld2View(R, T).T
    view = world_to_view(w2c_R.T.astype(np.float32), w2c_t.astype(np.float32)).T.copy()
    fovy = math.radians(fovy_deg)
    focal = height / (2.0 * math.tan(fovy / 2))
    fovx = 2.0 * math.atan(width / (2.0 * focal))   # graphics_utils.py:91-92 focal2fov
    proj = projection_matrix(znear, zfar, fovx, fovy).T.copy()
    proj[2, 0] = -0.0                               # tetgs_model.py:498-499: -K[0,0,2] (centred)
    proj[2, 1] = -0.0
    full = (view.astype(np.float32) @ proj.astype(np.float32)).astype(np.float32)
    return Camera(image_height=height, image_width=width, tanfovx=math.tan(fovx / 2),
                  tanfovy=math.tan(fovy / 2), viewmatrix=view.astype(np.float32), projmatrix=full,
                  campos=C.astype(np.float32), bg=np.asarray(bg, np.float32))


def make_cloud(P: int, sh_degree: int = 3, seed: int = 1234, scale_mult: float = 1.0,
               flat_fraction: float = 0.5, tiny_fraction: float = 0.0, n_oversized: int = 0,
               oversize: float = 50.0, M: Optional[int] = None) -> Dict[str, np.ndarray]:
    """SURVEY.md section 8d cloud: 70 % on an ellipsoid shell (0.35, 0.9, 0.25), 30 % in the unit
    ball; log-normal scales around 0.9/sqrt(P); half of the Gaussians flat (one axis x0.05);
    ``tiny_fraction`` with smallest scale 1e-8 (TetGS 2-D Gaussians, tetgs_edit_2d.py:203);
    ``n_oversized`` splats scaled by ``oversize`` (tile-list overflow stress, config 5)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n_shell = int(round(0.7 * P))
    d = rng.standard_normal((n_shell, 3))
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-12)
    shell = d * np.array([0.35, 0.9, 0.25])
    b = rng.standard_normal((P - n_shell, 3))
    b /= np.maximum(np.linalg.norm(b, axis=1, keepdims=True), 1e-12)
    ball = b * rng.random((P - n_shell, 1)) ** (1.0 / 3.0)
    means = np.concatenate([shell, ball], 0)
    means = means[rng.permutation(P)]
    sbar = 0.9 / math.sqrt(max(P, 1)) * scale_mult
    scales = np.exp(rng.normal(math.log(sbar), 0.5, (P, 3)))
    flat = rng.random(P) < flat_fraction
    axis = rng.integers(0, 3, P)
    scales[np.arange(P)[flat], axis[flat]] *= 0.05
    if tiny_fraction > 0:
        tiny = rng.random(P) < tiny_fraction
        scales[np.arange(P)[tiny], axis[tiny]] = 1e-8
    if n_oversized > 0:
        big = rng.choice(P, size=min(n_oversized, P), replace=False)
        scales[big] *= oversize
    q = rng.standard_normal((P, 4))
    q /= np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-12)
    opac = 1.0 / (1.0 + np.exp(-rng.normal(0.0, 2.0, (P, 1))))
    opac = np.clip(opac, 0.01, 0.999)
    Mc = (sh_degree + 1) ** 2 if M is None else M
    sh = rng.normal(0.0, 0.1, (P, Mc, 3))
    sh[:, 0, :] = rng.standard_normal((P, 3))
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return {"means3D": f32(means), "scales": f32(scales), "rotations": f32(q), "opacities": f32(opac),
            "shs": f32(sh), "sh_degree": sh_degree}


def concentrate(cloud: Dict[str, np.ndarray], n: int, centre=(0.0, 0.0, 0.0), sigma: float = 0.004, seed: int = 5,
                opacity=(0.004, 0.02)) -> Dict[str, np.ndarray]:
    """Moves the first ``n`` Gaussians of ``cloud`` into a blob of standard deviation ``sigma`` around ``centre`` and makes them
    faint -- thousands of splats whose centres fall into ONE 16x16 tile: a tile list far beyond the LDS sort budget (8192
    entries), the case the reference's global radix sort handles like any other (rasterizer_impl.cu:303-308)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in cloud.items()}
    out["means3D"][:n] = (np.asarray(centre, np.float64) + sigma * rng.standard_normal((n, 3))).astype(np.float32)
    out["opacities"][:n, 0] = rng.uniform(opacity[0], opacity[1], n).astype(np.float32)
    return out


def morton_order(cloud: Dict[str, np.ndarray], bits: int = 10) -> Dict[str, np.ndarray]:
    """The same cloud with its Gaussians re-numbered along a 3-D Morton curve of their centres: consecutive indices are spatial
    neighbours, the way mesh-bound Gaussians come (TetGS binds 1 or 3 Gaussians to every face of a marching-tetrahedra surface,
    tetgs_scene/tetgs_model.py:335-377, and faces come out cell by cell).  make_cloud's own order is a random permutation -- the worst
    case for everything that profits from neighbours sharing tiles."""
    m = cloud["means3D"].astype(np.float64)
    lo, hi = m.min(0), m.max(0)
    q = np.clip(((m - lo) / np.maximum(hi - lo, 1e-12) * ((1 << bits) - 1)).astype(np.uint64), 0, (1 << bits) - 1)
    code = np.zeros(len(m), np.uint64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    perm = np.argsort(code, kind="stable")
    return {k: (np.ascontiguousarray(v[perm]) if isinstance(v, np.ndarray) and v.shape[:1] == (len(m),) else v) for k, v in cloud.items()}


def upstream_gradient(width: int, height: int, seed: int = 99) -> np.ndarray:
    """dL/d out_color ~ N(0,1)/(3HW)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return (rng.standard_normal((3, height, width)) / (3.0 * height * width)).astype(np.float32)


def sh_to_rgb_numpy(shs: np.ndarray, means: np.ndarray, campos: np.ndarray, deg: int) -> np.ndarray:
    """What the callers do when compute_color_in_rasterizer=False (tetgs_model.py:413-442):
    eval_sh (utils/spherical_harmonics.py:117-172) + 0.5, clamp_min 0.  float32 numpy."""
    d = means - campos.reshape(1, 3)
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    C1 = 0.4886025119029199
    C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
    C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
          -0.4570457994644658, 1.445305721320277, -0.5900435899266435]
    r = SH_C0 * shs[:, 0]
    if deg > 0:
        r = r - C1 * y * shs[:, 1] + C1 * z * shs[:, 2] - C1 * x * shs[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = (r + C2[0] * xy * shs[:, 4] + C2[1] * yz * shs[:, 5] + C2[2] * (2 * zz - xx - yy) * shs[:, 6]
             + C2[3] * xz * shs[:, 7] + C2[4] * (xx - yy) * shs[:, 8])
    if deg > 2:
        r = (r + C3[0] * y * (3 * xx - yy) * shs[:, 9] + C3[1] * xy * z * shs[:, 10]
             + C3[2] * y * (4 * zz - xx - yy) * shs[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12]
             + C3[4] * x * (4 * zz - xx - yy) * shs[:, 13] + C3[5] * z * (xx - yy) * shs[:, 14]
             + C3[6] * x * (xx - 3 * yy) * shs[:, 15])
    return np.maximum(r + 0.5, 0.0).astype(np.float32)


# BASELINE.json configs (index -> kwargs); seeds = 1234 + index
CONFIGS = {
    1: dict(P=10_000, width=256, height=256, sh_degree=0, views=1, seed=1235),
    2: dict(P=100_000, width=800, height=800, sh_degree=3, views=1, seed=1236),
    3: dict(P=500_000, width=1920, height=1080, sh_degree=3, views=1, seed=1237),
    4: dict(P=500_000, width=1920, height=1080, sh_degree=3, views=64, seed=1238),
    # "tile-overflow stress": SURVEY.md 8d's 1000 oversized splats (scale x50, ~100 tiles each) put only ~6 extra entries into a tile, so
    # on their own no list comes near the LDS sort budget; 20 000 faint splats piled into one tile make the config do what it is named for
    5: dict(P=2_000_000, width=2048, height=2048, sh_degree=3, views=8, seed=1239,
            tiny_fraction=0.01, n_oversized=1000, n_pileup=20_000),
}


def config_cloud(index: int) -> Dict[str, np.ndarray]:
    """The Gaussian cloud of a BASELINE.json config (bench.py and the tests build the same one)."""
    c = CONFIGS[index]
    cloud = make_cloud(c["P"], c["sh_degree"], c["seed"], tiny_fraction=c.get("tiny_fraction", 0.0), n_oversized=c.get("n_oversized", 0))
    if c.get("n_pileup", 0):
        cloud = concentrate(cloud, c["n_pileup"], centre=(0.01, -0.01, 0.3), sigma=0.0015, seed=c["seed"] + 7)
    return cloud


def config_scene(index: int):
    """Returns (cloud dict, [Camera per view], upstream gradient) for a BASELINE.json config."""
    c = CONFIGS[index]
    cloud = config_cloud(index)
    cams = [orbit_camera(c["width"], c["height"], azimuth_deg=k * 360.0 / c["views"]) for k in range(c["views"])]
    return cloud, cams, upstream_gradient(c["width"], c["height"], seed=c["seed"] + 1000)
