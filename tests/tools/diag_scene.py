#!/usr/bin/env python3
"""Where does one fuzz scene's HIP result leave the oracle's?   python tests/tools/diag_scene.py <seed> <scene> [<scene> ...]
For each scene: rel-L2 of the per-pixel backward's outputs against the fp32 oracle under the library defaults, with the fixed-order backward,
without instance pruning and without light groups; n_contrib / final_T agreement; and the Gaussians that carry the difference."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tests import util, fuzz

seed, want = int(sys.argv[1]), [int(x) for x in sys.argv[2:]]
rng = np.random.default_rng(seed)
for it in range(max(want) + 1):
    desc, inp, dL = fuzz.random_scene(rng, it)
    if it not in want:
        continue
    ref = util.oracle_run(inp, dL)
    print(f"== seed {seed} scene {it} {desc}")
    for name, kw in (("default", {}), ("deterministic", dict(deterministic=True)), ("no pruning", dict(pruning=False)), ("no light groups", dict(light_tiles=False))):
        mine = util.hip_run(inp, dL, **kw)
        r = {k: util.rel_l2(np.asarray(mine[k]).reshape(np.asarray(ref[k]).shape), ref[k]) for k in ("color", "dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors") if k in mine and k in ref}
        nc = (mine["n_contrib"] == ref["n_contrib"]).mean() if "n_contrib" in mine and "n_contrib" in ref else float("nan")
        ft = np.abs(mine["final_T"] - ref["final_T"]).max() if "final_T" in mine and "final_T" in ref else float("nan")
        print(f"  {name:16s} " + " ".join(f"{k} {v:.2e}" for k, v in r.items()) + f" | n_contrib equal {nc:.6f} max|dT_final| {ft:.2e}")
        if name == "default":
            d = np.abs(np.asarray(mine["dL_dconic"]).reshape(np.asarray(ref["dL_dconic"]).shape) - ref["dL_dconic"]).sum(axis=1)
            top = np.argsort(d)[::-1][:5]
            tot = np.abs(ref["dL_dconic"]).sum()
            for g in top:
                print(f"    gaussian {g}: |d dL_dconic| {d[g]:.3e} of {np.abs(ref['dL_dconic'][g]).sum():.3e} (tensor total {tot:.3e}); tiles_touched hip {int(mine['tiles_touched'][g])} oracle {int(ref['tiles_touched'][g]) if 'tiles_touched' in ref else -1};"
                      f" opacity {float(np.ravel(inp['opacities'])[g]):.4f} radius {int(mine['radii'][g])}")
            if "n_contrib" in ref:
                bad = np.argwhere(mine["n_contrib"] != ref["n_contrib"])
                print(f"    pixels with another n_contrib: {len(bad)}" + (f", first {bad[:4].tolist()} hip {[int(mine['n_contrib'][tuple(b)]) for b in bad[:4]]} oracle {[int(ref['n_contrib'][tuple(b)]) for b in bad[:4]]}" if len(bad) else ""))
