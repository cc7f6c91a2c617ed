import sys, numpy as np, torch
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", ".."))
from youreditableavatar_amd import scenes
from tests import util
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    P = int(rng.integers(50, 6000)); W = int(rng.integers(17, 300)); H = int(rng.integers(17, 220)); D = int(rng.integers(0, 4))
    sm = float(rng.choice([0.3, 1.0, 3.0, 8.0])); ff = float(rng.uniform(0, 1)); tf = float(rng.choice([0.0, 0.05]))
    cloud = scenes.make_cloud(P, D, seed=int(rng.integers(1 << 30)), scale_mult=sm, flat_fraction=ff, tiny_fraction=tf, n_oversized=int(rng.choice([0, 0, 3])))
    cam = scenes.orbit_camera(W, H, azimuth_deg=float(rng.uniform(0, 360)), elevation_deg=float(rng.uniform(-30, 30)))
    inp = util.scene_input(cloud, cam)
    dL = scenes.upstream_gradient(W, H, seed=it)
    try:
        ref = util.oracle_run(inp, dL)
        mine = util.hip_run(inp, dL)
        rep = util.compare(mine, ref)
        print(it, P, W, H, D, sm, f"R={mine['num_rendered']}/{ref['num_rendered']}", "ok", {k: f"{v:.1e}" for k, v in rep.items() if k in ("color", "lists_equal", "instances_dropped", "n_contrib_equal")})
    except AssertionError as e:
        bad += 1
        print(it, P, W, H, D, sm, "FAIL", str(e)[:200])
print("failures", bad)
