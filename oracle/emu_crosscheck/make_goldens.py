#!/usr/bin/env python3
"""Generates tests/golden/*.npz by executing the reference's kernel text through the CUDA
execution-model emulation (cuda_emu.h / driver.cpp; build.sh first).  BUILD CONTAINER ONLY: needs
/root/reference and writes nothing of it into the repository -- a fixture is data (inputs + outputs).

Also prints how far ../tgs_oracle.c and the two arithmetic variants of the harness (FMA contraction
on/off, bounding what nvcc's default -fmad could change) are from each other.

Usage: oracle/emu_crosscheck/build.sh && python oracle/emu_crosscheck/make_goldens.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from youreditableavatar_amd import scenes  # noqa: E402
from oracle import oracle  # noqa: E402

EMU_DIR = os.environ.get("OUT", "/tmp/tgs_emu")
GOLD = os.path.join(ROOT, "tests", "golden")


def load(variant):
    L = C.CDLL(os.path.join(EMU_DIR, f"libtgs_emu_{variant}.so"))
    L.emu_run.restype = C.c_int
    return L


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def emu_run(L, inp, dL):
    P = inp["means3D"].shape[0]
    H, W = int(inp["image_height"]), int(inp["image_width"])
    shs = inp.get("shs"); cp = inp.get("colors_precomp")
    M = shs.shape[1] if shs is not None else 0
    T = ((W + 15) // 16) * ((H + 15) // 16)
    z = lambda shape, dt=np.float32: np.zeros(shape, dt)
    o = dict(color=z((3, H, W)), radii=z(P, np.int32), n_contrib=z((H, W), np.uint32), final_T=z((H, W)),
             means2D=z((P, 2)), depths=z(P), conic_opacity=z((P, 4)), rgb=z((P, 3)), tiles_touched=z(P, np.uint32),
             ranges=z((T, 2), np.uint32))
    cap = 1 << 24
    point_list = z(cap, np.uint32)
    g = dict(dL_dmeans2D=z((P, 3)), dL_dconic=z((P, 4)), dL_dopacity=z((P, 1)), dL_dcolors=z((P, 3)),
             dL_dmeans3D=z((P, 3)), dL_dcov3D=z((P, 6)), dL_dsh=z((P, M, 3)), dL_dscales=z((P, 3)), dL_drotations=z((P, 4)))
    f = lambda k: ptr(np.ascontiguousarray(inp[k], np.float32)) if inp.get(k) is not None else None
    keep = {k: np.ascontiguousarray(inp[k], np.float32) for k in
            ("bg", "means3D", "shs", "colors_precomp", "opacities", "scales", "rotations", "cov3D_precomp",
             "viewmatrix", "projmatrix", "campos") if inp.get(k) is not None}
    kp = lambda k: ptr(keep[k]) if k in keep else None
    L.emu_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_float] + \
        [C.c_void_p] * 5 + [C.c_float, C.c_float] + [C.c_void_p] * 11 + [C.c_longlong] + [C.c_void_p] * 10
    R = L.emu_run(P, int(inp["sh_degree"]), M, kp("bg"), W, H, kp("means3D"), kp("shs"), kp("colors_precomp"),
                  kp("opacities"), kp("scales"), float(inp["scale_modifier"]), kp("rotations"), kp("cov3D_precomp"),
                  kp("viewmatrix"), kp("projmatrix"), kp("campos"), float(inp["tanfovx"]), float(inp["tanfovy"]),
                  ptr(o["color"]), ptr(o["radii"]), ptr(o["n_contrib"]), ptr(o["final_T"]), ptr(o["means2D"]),
                  ptr(o["depths"]), ptr(o["conic_opacity"]), ptr(o["rgb"]), ptr(o["tiles_touched"]), ptr(o["ranges"]),
                  ptr(point_list), cap, ptr(dL), ptr(g["dL_dmeans2D"]), ptr(g["dL_dconic"]), ptr(g["dL_dopacity"]),
                  ptr(g["dL_dcolors"]), ptr(g["dL_dmeans3D"]), ptr(g["dL_dcov3D"]), ptr(g["dL_dsh"]), ptr(g["dL_dscales"]),
                  ptr(g["dL_drotations"]))
    assert R >= 0
    o["num_rendered"] = np.int64(R)
    o["point_list"] = point_list[:R].copy()
    o.update(g)
    return o


def cov3d_from(scales, rots, mod=1.0):
    """What the callers do for compute_covariance_in_rasterizer=False (tetgs_model.py:559-577)."""
    r, x, y, z = rots[:, 0], rots[:, 1], rots[:, 2], rots[:, 3]
    Rm = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
                   np.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
                   np.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], -2).astype(np.float32)
    S = (mod * scales).astype(np.float32)
    RS = Rm * S[:, None, :]
    Sig = RS @ RS.transpose(0, 2, 1)
    return np.stack([Sig[:, 0, 0], Sig[:, 0, 1], Sig[:, 0, 2], Sig[:, 1, 1], Sig[:, 1, 2], Sig[:, 2, 2]], -1).astype(np.float32)


def make_input(P, W, H, deg, seed, mode="sh", cov_mode="scale_rot", scale_mult=6.0, az=20.0, el=5.0, radius=3.0,
               bg=(1, 1, 1), scale_modifier=1.0, M=None, tweak=None, **cloud_kw):
    cloud = scenes.make_cloud(P, deg, seed=seed, scale_mult=scale_mult, M=M, **cloud_kw)
    cam = scenes.orbit_camera(W, H, azimuth_deg=az, elevation_deg=el, radius=radius, bg=bg)
    if tweak:
        tweak(cloud, cam)
    inp = dict(bg=cam.bg, means3D=cloud["means3D"], opacities=cloud["opacities"], viewmatrix=cam.viewmatrix,
               projmatrix=cam.projmatrix, campos=cam.campos, tanfovx=np.float32(cam.tanfovx), tanfovy=np.float32(cam.tanfovy),
               image_height=np.int32(H), image_width=np.int32(W), sh_degree=np.int32(deg),
               scale_modifier=np.float32(scale_modifier))
    if mode == "sh":
        inp["shs"] = cloud["shs"]
    else:
        inp["colors_precomp"] = scenes.sh_to_rgb_numpy(cloud["shs"], cloud["means3D"], cam.campos, deg)
    if cov_mode == "scale_rot":
        inp["scales"], inp["rotations"] = cloud["scales"], cloud["rotations"]
    else:
        inp["cov3D_precomp"] = cov3d_from(cloud["scales"], cloud["rotations"])
    return inp


def t_ties(cloud, cam):       # exact depth ties + duplicated positions: the stable (tile, depth, idx) order matters
    n = cloud["means3D"].shape[0]
    cloud["means3D"][n // 2:] = cloud["means3D"][: n - n // 2]


def t_opaque(cloud, cam):     # opacities at/over the 0.99 clamp, early termination everywhere
    cloud["opacities"][:] = 0.9999


def t_giant(cloud, cam):      # one splat covering every tile + a few normal ones
    cloud["scales"][0] = 5.0
    cloud["means3D"][0] = 0.0


def t_behind(cloud, cam):     # everything behind the camera -> all culled, R = 0
    cloud["means3D"][:] = cloud["means3D"] * 0.1 + cam.campos * 2.0


def t_clampxy(cloud, cam):    # points far outside the frustum: the 1.3*tanfov clamp of forward.cu:82-87
    cloud["means3D"][::3, 0] += 2.5
    cloud["scales"][::3] *= 8.0


SCENES = {
    "g01_sh3_scale_rot":      dict(P=400, W=96, H=80, deg=3, seed=11),
    "g02_sh0_nonmult16":      dict(P=300, W=70, H=50, deg=0, seed=12, bg=(0.2, 0.5, 0.9)),
    "g03_precomp_scale_rot":  dict(P=400, W=64, H=64, deg=2, seed=13, mode="precomp"),
    "g04_sh2_cov3d":          dict(P=300, W=64, H=48, deg=2, seed=14, cov_mode="cov3d"),
    "g05_precomp_cov3d":      dict(P=300, W=48, H=64, deg=1, seed=15, mode="precomp", cov_mode="cov3d", bg=(0, 0, 0)),
    "g06_sh1_stride16":       dict(P=300, W=64, H=64, deg=1, seed=16, M=16),
    "g07_depth_ties":         dict(P=200, W=64, H=64, deg=0, seed=17, tweak=t_ties),
    "g08_opaque_termination": dict(P=600, W=64, H=64, deg=0, seed=18, scale_mult=10.0, tweak=t_opaque),
    "g09_giant_splat":        dict(P=100, W=80, H=64, deg=1, seed=19, tweak=t_giant),
    "g10_all_culled":         dict(P=50, W=32, H=32, deg=0, seed=20, tweak=t_behind),
    "g11_flat_1e-8":          dict(P=400, W=64, H=64, deg=3, seed=21, tiny_fraction=0.5, scale_mult=8.0),
    "g12_frustum_clamp_mod":  dict(P=300, W=96, H=48, deg=3, seed=22, tweak=t_clampxy, scale_modifier=1.3, az=200.0, el=-20.0),
    "g13_dense_2k":           dict(P=2000, W=128, H=96, deg=3, seed=23, scale_mult=3.0),
    "g14_tiny_splats":        dict(P=1500, W=112, H=112, deg=0, seed=24, scale_mult=1.0, radius=2.0),
}


def f64_truth(inp, dL):
    import torch
    from oracle import torch_splat
    dt = torch.float64
    t = lambda x, g=False: torch.tensor(np.asarray(x), dtype=dt, requires_grad=g)
    P = inp["means3D"].shape[0]
    leaves = {"means3D": t(inp["means3D"], True), "means2D": torch.zeros(P, 3, dtype=dt, requires_grad=True), "opacities": t(inp["opacities"], True)}
    for k in ("shs", "colors_precomp", "scales", "rotations", "cov3D_precomp"):
        if inp.get(k) is not None:
            leaves[k] = t(inp[k], True)
    color, _ = torch_splat.splat(viewmatrix=t(inp["viewmatrix"]), projmatrix=t(inp["projmatrix"]), campos=t(inp["campos"]), bg=t(inp["bg"]),
                                 tanfovx=float(inp["tanfovx"]), tanfovy=float(inp["tanfovy"]), image_height=int(inp["image_height"]),
                                 image_width=int(inp["image_width"]), sh_degree=int(inp["sh_degree"]),
                                 scale_modifier=float(inp["scale_modifier"]), **leaves)
    color.backward(torch.tensor(dL, dtype=dt))
    return {k: v.grad.numpy() for k, v in leaves.items() if v.grad is not None}


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-12))


def main():
    os.makedirs(GOLD, exist_ok=True)
    Lf, Ln = load("fma"), load("nofma")
    worst = {}
    for name, kw in SCENES.items():
        inp = make_input(**kw)
        H, W = int(inp["image_height"]), int(inp["image_width"])
        dL = scenes.upstream_gradient(W, H, seed=1000 + kw["seed"])
        a = emu_run(Lf, inp, dL)
        b = emu_run(Ln, inp, dL)
        # the restatement, same inputs
        okw = {k: inp.get(k) for k in ("shs", "colors_precomp", "scales", "rotations", "cov3D_precomp")}
        common = dict(bg=inp["bg"], means3D=inp["means3D"], viewmatrix=inp["viewmatrix"], projmatrix=inp["projmatrix"],
                      campos=inp["campos"], tanfovx=float(inp["tanfovx"]), tanfovy=float(inp["tanfovy"]),
                      scale_modifier=float(inp["scale_modifier"]), **okw)
        color, radii, st = oracle.forward(opacities=inp["opacities"], image_height=H, image_width=W,
                                          sh_degree=int(inp["sh_degree"]), **common)
        g = oracle.backward(st, dL, **common)
        keys = ["color"] + [k for k in a if k.startswith("dL_")]
        line = []
        for k in keys:
            ref = a[k]
            mine = color if k == "color" else g[k]
            e_or, e_fm = rel(mine, ref), rel(b[k], ref)
            worst[k] = max(worst.get(k, (0, 0))[0], e_or), max(worst.get(k, (0, 0))[1], e_fm)
            line.append(f"{k.replace('dL_d', '')}:{e_or:.1e}/{e_fm:.1e}")
        exact = (np.array_equal(radii, a["radii"]) and st.num_rendered == int(a["num_rendered"])
                 and np.array_equal(st.field("point_list"), a["point_list"])
                 and np.array_equal(st.field("n_contrib").reshape(H, W), a["n_contrib"]))
        nc_fm = float((a["n_contrib"] == b["n_contrib"]).mean())
        print(f"{name}: R={int(a['num_rendered'])} F={int(a['n_contrib'].sum())} exact(radii,R,list,n_contrib)={exact} "
              f"n_contrib fma-vs-nofma {nc_fm:.5f}\n    oracle/nofma vs emu-fma: " + " ".join(line))
        out = {("in_" + k): v for k, v in inp.items() if v is not None}
        out["in_dL_dout_color"] = dL
        out.update({("out_" + k): v for k, v in a.items()})
        # the same kernel text without FMA contraction: the reference's own arithmetic noise floor for the
        # cancellation-prone tensors (tests scale their tolerance with it)
        for k in ("dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations"):
            out["out_nofma_" + k] = b[k]
        # exact-arithmetic value of the same gradients (independent fp64 autograd splat, oracle/torch_splat.py):
        # how far the reference's OWN fp32 result is from the truth bounds what "parity" can mean
        tr = f64_truth(inp, dL)
        for k, leaf in (("dL_dmeans3D", "means3D"), ("dL_dscales", "scales"), ("dL_drotations", "rotations"), ("dL_dcov3D", "cov3D_precomp")):
            if leaf in tr:
                out["out_f64_" + k] = tr[leaf].astype(np.float32)
                print(f"    reference fp32 vs fp64 truth {k}: {rel(a[k], tr[leaf]):.2e}")
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print("worst rel-L2 (oracle vs emu-fma, emu-nofma vs emu-fma):")
    for k, (x, y) in worst.items():
        print(f"  {k:16s} {x:.2e} {y:.2e}")


if __name__ == "__main__":
    main()
