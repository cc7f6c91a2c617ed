"""Diagnose one scene of fuzz_vs_oracle.py: python tests/tools/fuzz_diagnose.py <seed> <iteration>.  Prints where the gradient
difference sits (a few Gaussians = a threshold flip on a pixel; spread out = a bug)."""
import sys, numpy as np, torch
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", ".."))
from youreditableavatar_amd import scenes
from tests import util
rng = np.random.default_rng(int(sys.argv[1]))
target = int(sys.argv[2])
for it in range(target + 1):
    P = int(rng.integers(50, 6000)); W = int(rng.integers(17, 300)); H = int(rng.integers(17, 220)); D = int(rng.integers(0, 4))
    sm = float(rng.choice([0.3, 1.0, 3.0, 8.0])); ff = float(rng.uniform(0, 1)); tf = float(rng.choice([0.0, 0.05]))
    seed = int(rng.integers(1 << 30)); nov = int(rng.choice([0, 0, 3]))
    az = float(rng.uniform(0, 360)); el = float(rng.uniform(-30, 30))
cloud = scenes.make_cloud(P, D, seed=seed, scale_mult=sm, flat_fraction=ff, tiny_fraction=tf, n_oversized=nov)
cam = scenes.orbit_camera(W, H, azimuth_deg=az, elevation_deg=el)
inp = util.scene_input(cloud, cam)
dL = scenes.upstream_gradient(W, H, seed=target)
ref = util.oracle_run(inp, dL)
mine = util.hip_run(inp, dL)
print("scene", P, W, H, D, sm, "R", mine["num_rendered"], ref["num_rendered"])
print("color rel", util.rel_l2(mine["color"], ref["color"]), "max abs", np.abs(mine["color"] - ref["color"]).max())
lc_m, lc_r = util.last_contributor_ids(mine), util.last_contributor_ids(ref)
bad_px = np.argwhere(lc_m != lc_r)
print("pixels with another last contributor:", len(bad_px), bad_px[:5].tolist())
dT = np.abs(mine["final_T"] - ref["final_T"].reshape(H, W))
print("final_T max abs diff", dT.max(), "at", np.unravel_index(dT.argmax(), dT.shape))
for k in ("dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors", "dL_dscales", "dL_drotations", "dL_dmeans3D"):
    a = np.asarray(mine[k], np.float64).reshape(P, -1); b = np.asarray(ref[k], np.float64).reshape(P, -1)[:, :a.shape[1]]
    d = ((a - b) ** 2).sum(1)
    order = np.argsort(-d)[:4]
    print(k, "rel", np.sqrt(d.sum() / (b ** 2).sum()), "top Gaussians", [(int(i), f"{d[i] / d.sum():.2f}") for i in order])
    i = order[0]
    print("   ", i, "mine", a[i][:4], "ref", b[i][:4], "radius", ref["radii"][i], "mean2D", mine["means2D"][i], "conic_op", mine["conic_opacity"][i])
