#!/usr/bin/env python3
"""Mid-size goldens (tests/golden/m*.npz): the reference's kernel text under the CUDA execution-model emulation (cuda_emu.h /
driver.cpp; build.sh first) on scenes too large to commit in full -- 20 000 Gaussians at 256x256 and BASELINE config 2
(100 000 Gaussians, 800x800, SH degree 3).  BUILD CONTAINER ONLY (needs /root/reference through build.sh's extraction).

A fixture is data and has to stay small, so it holds a RECIPE instead of the inputs (the seeded generator of
youreditableavatar_amd/scenes.py reproduces them bit for bit; a checksum of the generated inputs is stored to detect drift) and of
the outputs: everything integer in full (radii, n_contrib, num_rendered, tiles_touched, the point list as a checksum per tile), the image
as its full-frame moments plus a 256x256 window, and every gradient tensor on a seeded subset of 4096 Gaussians plus its full-tensor
norm -- rel-L2 on the subset is the comparison.  Also stored: the same kernel text without FMA contraction on the subset
(the reference's own arithmetic noise, tests scale the tolerance of the cancellation-prone tensors with it).

Usage: oracle/emu_crosscheck/build.sh && python oracle/emu_crosscheck/make_mid_goldens.py
"""
import os
import sys
import time
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from youreditableavatar_amd import scenes  # noqa: E402
import make_goldens as mg  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SUBSET = 4096
WINDOW = 256

# name -> recipe (everything scenes.make_cloud / orbit_camera / upstream_gradient need)
RECIPES = {
    "m01_mid_20k_256": dict(P=20_000, W=256, H=256, deg=3, seed=31, scale_mult=2.0, az=35.0, dl_seed=1031),
    "m02_cfg2_800": dict(P=100_000, W=800, H=800, deg=3, seed=scenes.CONFIGS[2]["seed"], scale_mult=1.0, az=0.0, dl_seed=scenes.CONFIGS[2]["seed"] + 1000),
    # BASELINE config 3 = the headline workload at full size (500 000 Gaussians, 1920x1080, SH 3), view 0
    "m03_cfg3_1080p": dict(P=500_000, W=1920, H=1080, deg=3, seed=scenes.CONFIGS[3]["seed"], scale_mult=1.0, az=0.0, dl_seed=scenes.CONFIGS[3]["seed"] + 1000),
}


def build_input(r):
    """the scene of a recipe: (inp dict as the goldens use it, dL)"""
    inp = mg.make_input(P=r["P"], W=r["W"], H=r["H"], deg=r["deg"], seed=r["seed"], scale_mult=r["scale_mult"], az=r["az"])
    return inp, scenes.upstream_gradient(r["W"], r["H"], seed=r["dl_seed"])


def input_checksum(inp, dL):
    c = 0
    for k in sorted(inp):
        if isinstance(inp[k], np.ndarray):
            c = zlib.crc32(np.ascontiguousarray(inp[k]).tobytes(), c)
    return np.uint32(zlib.crc32(np.ascontiguousarray(dL).tobytes(), c))


def list_checksums(point_list, ranges):
    """per tile: crc32 of the tile's sorted list of Gaussian indices (0 for an empty tile)"""
    out = np.zeros(len(ranges), np.uint32)
    for t, (a, b) in enumerate(np.asarray(ranges).reshape(-1, 2)):
        if b > a:
            out[t] = zlib.crc32(np.ascontiguousarray(point_list[a:b], np.uint32).tobytes())
    return out


def subset_indices(P, seed):
    return np.sort(np.random.Generator(np.random.PCG64(seed + 4242)).choice(P, size=min(SUBSET, P), replace=False)).astype(np.int64)


def main():
    Lf, Ln = mg.load("fma"), mg.load("nofma")
    only = sys.argv[1:]
    for name, r in RECIPES.items():
        if only and name not in only:
            continue
        inp, dL = build_input(r)
        H, W, P = r["H"], r["W"], r["P"]
        t0 = time.time()
        a = mg.emu_run(Lf, inp, dL)
        b = mg.emu_run(Ln, inp, dL)
        print(f"{name}: emulation {time.time() - t0:.1f} s, R={int(a['num_rendered'])}, F={int(a['n_contrib'].astype(np.int64).sum())}")
        idx = subset_indices(P, r["seed"])
        y0, x0 = (H - min(WINDOW, H)) // 2, (W - min(WINDOW, W)) // 2
        out = {("recipe_" + k): np.asarray(v) for k, v in r.items()}
        out["input_crc"] = input_checksum(inp, dL)
        out["subset"] = idx
        out["window"] = np.asarray([y0, x0, min(WINDOW, H), min(WINDOW, W)], np.int32)
        out["out_num_rendered"] = np.int64(a["num_rendered"])
        out["out_radii"] = a["radii"].astype(np.int32)
        out["out_tiles_touched"] = a["tiles_touched"].astype(np.uint32)
        out["out_n_contrib"] = a["n_contrib"].astype(np.uint16 if a["n_contrib"].max() < 65536 else np.uint32)
        out["out_ranges"] = a["ranges"].astype(np.uint32)
        out["out_list_crc"] = list_checksums(a["point_list"], a["ranges"])
        col = a["color"].astype(np.float64)
        out["out_color_sum"] = col.sum(axis=(1, 2))
        out["out_color_sumsq"] = (col * col).sum(axis=(1, 2))
        out["out_color_window"] = a["color"][:, y0:y0 + WINDOW, x0:x0 + WINDOW].copy()
        out["out_final_T_window"] = a["final_T"][y0:y0 + WINDOW, x0:x0 + WINDOW].copy()
        for k in ("dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"):
            out["out_" + k + "_subset"] = a[k][idx].copy()
            out["out_" + k + "_norm"] = np.float64(np.linalg.norm(a[k].astype(np.float64)))
        for k in ("dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations"):
            out["out_nofma_" + k + "_subset"] = b[k][idx].copy()
        for k in ("means2D", "depths", "conic_opacity", "rgb"):
            out["out_" + k + "_subset"] = a[k][idx].copy()
        path = os.path.join(GOLD, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"    wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB); n_contrib fma-vs-nofma equal on {float((a['n_contrib'] == b['n_contrib']).mean()):.6f}")


if __name__ == "__main__":
    main()
