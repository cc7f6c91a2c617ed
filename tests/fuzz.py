"""Random small scenes against the CPU oracle (the body of tests/tools/fuzz_vs_oracle.py as a function, so that the GPU suite can assert
on it).  A *miss* is a scene on which util.compare raises: some tensor further than max(1e-4, 2 x the distance of the reference's own fp32
arithmetic from exact arithmetic on that scene) from BOTH the fp32 oracle and the oracle's text evaluated in double.  The suite asserts
ZERO misses."""
from __future__ import annotations

import numpy as np

from tests import util


def random_scene(rng, it: int):
    from youreditableavatar_amd import scenes
    P = int(rng.integers(50, 6000)); W = int(rng.integers(17, 300)); H = int(rng.integers(17, 220)); D = int(rng.integers(0, 4))
    sm = float(rng.choice([0.3, 1.0, 3.0, 8.0])); ff = float(rng.uniform(0, 1)); tf = float(rng.choice([0.0, 0.05]))
    cloud = scenes.make_cloud(P, D, seed=int(rng.integers(1 << 30)), scale_mult=sm, flat_fraction=ff, tiny_fraction=tf, n_oversized=int(rng.choice([0, 0, 3])))
    cam = scenes.orbit_camera(W, H, azimuth_deg=float(rng.uniform(0, 360)), elevation_deg=float(rng.uniform(-30, 30)))
    return (P, W, H, D, sm), util.scene_input(cloud, cam), scenes.upstream_gradient(W, H, seed=it)


def diagnose(mine: dict, ref: dict, P: int) -> str:
    """where does the difference sit?  (one Gaussian / one pixel = a threshold flip or an ill-conditioned splat, not a defect)"""
    try:
        dT = np.abs(np.asarray(mine["final_T"], np.float64) - np.asarray(ref["final_T"], np.float64).reshape(np.asarray(mine["final_T"]).shape))
        conc = {}
        for k in ("dL_dmeans2D", "dL_dconic", "dL_dscales", "dL_drotations"):
            if k in mine and k in ref:
                a = np.asarray(mine[k], np.float64).reshape(P, -1); b = np.asarray(ref[k], np.float64).reshape(P, -1)[:, :a.shape[1]]
                d = ((a - b) ** 2).sum(1)
                conc[k] = float(d.max() / max(d.sum(), 1e-300))
        return f"pixels with |dT| > 1e-3: {int((dT > 1e-3).sum())}, share of the squared error in ONE Gaussian: " + ", ".join(f"{k} {v:.2f}" for k, v in conc.items())
    except Exception as ex:      # noqa: BLE001
        return f"(no diagnosis: {ex})"


def run(seed: int, n_scenes: int, log=print, light_tiles=None) -> dict:
    """-> {scenes, misses, miss_rate, worst_rel_l2, worst (text), largest_ok: per-tensor maximum over the scenes that passed,
    over_1e4: how many (scene, tensor) pairs needed the scene's own noise floor, hip_closer_to_exact: in how many of those the HIP result
    is closer to exact arithmetic than the fp32 oracle, passed_through_exact_only: how many passed by their distance to the double evaluation
    while further than the bar from the fp32 oracle, scenes_beyond_the_oracle_route: scenes in which some tensor passed by the second
    or third route of util.compare, routes: {route: tensors over 1e-4 that passed by it}}
    light_tiles: tgs_options_t::light_tiles for forward and backward (None: the entry points' default)"""
    import re
    rng = np.random.default_rng(seed)
    misses, worst, worst_txt, largest = 0, 0.0, "", {}
    over = closer = via_exact = 0
    scenes_beyond, routes = 0, {}
    for it in range(n_scenes):
        desc, inp, dL = random_scene(rng, it)
        ref = util.oracle_run(inp, dL)
        mine = util.hip_run(inp, dL, light_tiles=light_tiles)
        try:
            rep = util.compare(mine, ref)
            scenes_beyond += rep.get("routes_beyond_oracle", 0) > 0
            for k, v in rep.items():
                if k.endswith("|route"):
                    routes[v] = routes.get(v, 0) + 1
            for k, v in rep.items():
                if "|" in k:
                    continue
                if k.startswith("dL_") or k == "color":
                    largest[k] = max(largest.get(k, 0.0), float(v))
                    if (k + "|vs_f64") in rep:
                        over += 1
                        closer += rep[k + "|vs_f64"] <= ref["_noise"][k]
                        via_exact += v > rep[k + "|bar"]
            log(it, *desc, f"R={mine['num_rendered']}/{ref['num_rendered']}", "ok")
        except AssertionError as e:
            misses += 1
            m = re.search(r"rel-L2 ([0-9.e+-]+)", str(e))
            val = float(m.group(1)) if m else float("nan")
            txt = f"{it} {desc} {str(e)[:110]} | {diagnose(mine, ref, desc[0])}"
            if not (val <= worst):
                worst, worst_txt = val, txt
            log("MISS", txt)
    return dict(scenes=n_scenes, misses=misses, miss_rate=misses / max(n_scenes, 1), worst_rel_l2=worst, worst=worst_txt, seed=seed,
                over_1e4=over, hip_closer_to_exact=closer, passed_through_exact_only=via_exact,
                scenes_beyond_the_oracle_route=scenes_beyond, routes=routes, largest_ok={k: float(f"{v:.3g}") for k, v in largest.items()})
