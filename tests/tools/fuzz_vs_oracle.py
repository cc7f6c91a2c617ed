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
        # where does the difference sit?  (one Gaussian / one pixel = a threshold flip or an ill-conditioned splat, not a defect)
        note = ""
        try:
            dT = np.abs(np.asarray(mine["final_T"], np.float64) - np.asarray(ref["final_T"], np.float64).reshape(np.asarray(mine["final_T"]).shape))
            conc = {}
            for k in ("dL_dmeans2D", "dL_dconic", "dL_dscales", "dL_drotations"):
                if k in mine and k in ref:
                    a = np.asarray(mine[k], np.float64).reshape(P, -1); b = np.asarray(ref[k], np.float64).reshape(P, -1)[:, :a.shape[1]]
                    d = ((a - b) ** 2).sum(1)
                    conc[k] = float(d.max() / max(d.sum(), 1e-300))
            note = f" | pixels with |dT| > 1e-3: {int((dT > 1e-3).sum())}, share of the squared error in ONE Gaussian: " + ", ".join(f"{k} {v:.2f}" for k, v in conc.items())
        except Exception as ex:
            note = f" | (no diagnosis: {ex})"
        print(it, P, W, H, D, sm, "FAIL", str(e)[:120] + note)
print("failures", bad)
