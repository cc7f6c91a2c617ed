"""Three-way adjudication of a parity difference (test infrastructure).

The HIP path and the fp32 oracle are two roundings of one function.  Where they differ by more than 1e-4 the question is which side is
further from that function in exact arithmetic.  ``oracle/tgs_oracle.c`` compiles to seven libraries from one text (six enter the measure below; `f64_s32`, double arithmetic on fp32-rounded state, is a diagnostic variant): fp32 without FMA
contraction (the restatement), fp32 with contraction and the reference's fp32 accumulation of the cross-pixel sums (what nvcc and
atomicAdd do to the reference), fp32 with exp evaluated as 2^(x log2 e) (what a GPU math library does), fp32 with the compositing loop's
two cut-offs decided the other way -- every pair within fp32's evaluation noise of alpha >= 1/255 (max(1e-6, 8 ulp of the sum of the quadratic
form's term magnitudes), relative) blended / skipped, T >= 1e-4 moved by 1e-6 of its value -- (the function is discontinuous
there, and fp32's own evaluation noise -- 4e-7 of alpha on the pair that prompted this -- decides such pairs either way), and **double** (every intermediate
and every array; the fp32 literals, the fp32 inputs and the fp32 depth bits of the sort key stay).  For every tensor this module reports

    hip_vs_f64      rel_l2(HIP, f64)          how far the product is from exact arithmetic
    ref_vs_f64      rel_l2(oracle fp32, f64)  how far the reference's arithmetic is (max over the five fp32 builds)
    hip_vs_ref      rel_l2(HIP, oracle fp32)  what the parity bar is stated on

and the bar of the GPU suite is  min(hip_vs_ref, hip_vs_f64) <= max(1e-4, 2 x ref_vs_f64)  with no failure budget (tests/util.py: compare).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from tests import util

TENSORS = ("color",) + util.GRAD_KEYS + ("dL_dconic",)


def oracle_variant(inp: dict, dL: Optional[np.ndarray], variant: str) -> Dict[str, np.ndarray]:
    """forward (+ backward) of one oracle build on the scene ``inp`` (tests.util.scene_input)"""
    from oracle import oracle
    kw = dict(bg=inp["bg"], means3D=inp["means3D"], viewmatrix=inp["viewmatrix"], projmatrix=inp["projmatrix"], campos=inp["campos"],
              tanfovx=float(inp["tanfovx"]), tanfovy=float(inp["tanfovy"]), scale_modifier=float(inp.get("scale_modifier", 1.0)))
    for k in ("shs", "colors_precomp", "scales", "rotations", "cov3D_precomp"):
        kw[k] = inp.get(k)
    color, radii, st = oracle.forward(opacities=inp["opacities"], image_height=int(inp["image_height"]), image_width=int(inp["image_width"]),
                                      sh_degree=int(inp["sh_degree"]), variant=variant, **kw)
    out = dict(color=color, radii=radii, num_rendered=st.num_rendered)
    if dL is not None:
        out.update(oracle.backward(st, dL, **kw))
    return out


def reference_noise(inp: dict, dL: np.ndarray, ref32: Optional[dict] = None) -> Dict[str, float]:
    """{tensor: distance of the reference's fp32 arithmetic from the same function in double}: the largest of the five fp32 builds
    (no FMA contraction / contraction + fp32 accumulation / exp as 2^(x log2 e) / the loop's two cut-offs decided either way inside fp32's noise band: tgs_oracle.c, TGS_ORACLE_CUT);
    ``ref32``: an existing fp32 result (tests.util.oracle_run) to reuse."""
    f64 = oracle_variant(inp, dL, "f64")
    a = ref32 if ref32 is not None else oracle_variant(inp, dL, "f32")
    b = oracle_variant(inp, dL, "f32_fma")
    others = [oracle_variant(inp, dL, v) for v in ("f32_ex2", "f32_in", "f32_out")]
    noise = {}
    for k in TENSORS:
        if k in f64 and k in a:
            noise[k] = max([util.rel_l2(a[k], f64[k]), util.rel_l2(b[k], f64[k])] + [util.rel_l2(o[k], f64[k]) for o in others])
    noise["_f64"] = f64
    noise["_fma"] = b
    return noise


def three_way(mine: dict, ref32: dict, noise: dict) -> Dict[str, Dict[str, float]]:
    f64 = noise["_f64"]
    rep = {}
    for k in TENSORS:
        if k in mine and k in ref32 and k in f64:
            a = np.asarray(mine[k], np.float64).reshape(np.asarray(f64[k]).shape)
            rep[k] = dict(hip_vs_f64=util.rel_l2(a, f64[k]), ref_vs_f64=noise[k], hip_vs_ref=util.rel_l2(a, np.asarray(ref32[k]).reshape(a.shape)),
                          fma_vs_ref=util.rel_l2(noise["_fma"][k], ref32[k]))
    return rep


def run_fuzz(seed: int, n_scenes: int, light_tiles=None, log=print) -> dict:
    """the fuzz of tests/fuzz.py with the three-way numbers for every scene that exceeds 1e-4 on some tensor"""
    from tests import fuzz
    rng = np.random.default_rng(seed)
    over, hip_farther, misses, worst = [], 0, 0, {}
    for it in range(n_scenes):
        desc, inp, dL = fuzz.random_scene(rng, it)
        ref = util.oracle_run(inp, dL)
        mine = util.hip_run(inp, dL, light_tiles=light_tiles)
        direct = {k: util.rel_l2(np.asarray(mine[k]).reshape(np.asarray(ref[k]).shape), ref[k]) for k in TENSORS if k in mine and k in ref}
        for k, v in direct.items():
            worst[k] = max(worst.get(k, 0.0), v)
        if max(direct.values()) <= util.REL_TOL:
            continue
        tw = three_way(mine, ref, reference_noise(inp, dL, ref))
        for k, r in tw.items():
            if r["hip_vs_ref"] > util.REL_TOL:
                bar = max(util.REL_TOL, 2.0 * r["ref_vs_f64"])
                verdict = "ok" if r["hip_vs_ref"] <= bar else ("ok (within the bar of exact arithmetic)" if r["hip_vs_f64"] <= bar else "MISS")
                if verdict == "MISS" and k in util.PERGAUSS_KEYS and "dL_dconic" in mine:
                    # the per-Gaussian half on the product's own per-pixel gradients, in double (tests/util.py: compare, third route)
                    r["hip_vs_own_chain"] = util.rel_l2(np.asarray(mine[k]).reshape(np.asarray(ref[k]).shape), util.own_chain(mine, ref)[k])
                    upstream_ok = all(tw[u]["hip_vs_ref"] <= max(util.REL_TOL, 2.0 * tw[u]["ref_vs_f64"]) or tw[u]["hip_vs_f64"] <= max(util.REL_TOL, 2.0 * tw[u]["ref_vs_f64"])
                                      for u in ("dL_dmeans2D", "dL_dconic", "dL_dcolors") if u in tw)
                    if upstream_ok and r["hip_vs_own_chain"] <= util.CHAIN_TOL:
                        verdict = f"ok (the per-Gaussian half is exact on its own inputs: {r['hip_vs_own_chain']:.1e}; their noise, inside the bar, amplified)"
                misses += verdict == "MISS"
                hip_farther += r["hip_vs_f64"] > r["ref_vs_f64"]
                over.append((seed, it, desc, k, r, verdict))
                log(f"seed {seed} scene {it} {desc} {k}: hip_vs_ref {r['hip_vs_ref']:.3e} hip_vs_f64 {r['hip_vs_f64']:.3e} ref_vs_f64 {r['ref_vs_f64']:.3e} "
                    f"fma_vs_ref {r['fma_vs_ref']:.3e} -> {verdict}")
    return dict(seed=seed, scenes=n_scenes, tensors_over_1e4=len(over), hip_farther_from_f64=hip_farther, misses_at_new_bar=misses,
                worst_direct={k: float(f"{v:.3g}") for k, v in worst.items()})


def run_config(cfg: int, log=print) -> dict:
    """three-way numbers of a BASELINE configuration at full size (view 0)"""
    from youreditableavatar_amd import scenes
    cloud, cams, dL = scenes.config_scene(cfg)
    inp = util.scene_input(cloud, cams[0])
    mine = util.hip_run(inp, dL, introspect=False)
    ref = oracle_variant(inp, dL, "f32")
    tw = three_way(mine, ref, reference_noise(inp, dL, ref))
    for k, r in tw.items():
        log(f"cfg{cfg} {k}: " + " ".join(f"{n} {v:.3e}" for n, v in r.items()))
    return {k: {n: float(f"{v:.4g}") for n, v in r.items()} for k, r in tw.items()}


if __name__ == "__main__":
    import json
    import sys
    if sys.argv[1] == "cfg":
        print(json.dumps({f"cfg{c}": run_config(int(c)) for c in sys.argv[2:]}))
    else:
        seed, n = int(sys.argv[1]), int(sys.argv[2])
        lt = None if len(sys.argv) < 4 else bool(int(sys.argv[3]))
        print(json.dumps(run_fuzz(seed, n, lt)))
