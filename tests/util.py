"""Shared helpers of the test-suite: golden loading, oracle / HIP runners, the parity metric."""
from __future__ import annotations

import glob
import os
from typing import Dict, Optional

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

REL_TOL = 1e-4          # BASELINE.json: "within 1e-4 rel-L2 of reference"
NOISY = ("dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations")
GRAD_KEYS = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")


def rel_l2(x, ref) -> float:
    """SURVEY.md section 8d parity metric: ||x - ref|| / max(||ref||, 1e-12)."""
    x = np.asarray(x, np.float64); ref = np.asarray(ref, np.float64)
    return float(np.linalg.norm(x - ref) / max(np.linalg.norm(ref), 1e-12))


def golden_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "g[0-9]*.npz")))


def load_golden(name: str):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    inp = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    out = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
    return inp, out


def tolerance(key: str, golden_out: Optional[dict]) -> float:
    """1e-4, except for the cancellation-prone tensors where the fixture shows that the reference's own
    fp32 arithmetic is less accurate than that: FMA contraction on/off of the SAME kernel text
    (nofma_*) moves them by more, or the reference is further than 0.5e-4 from exact arithmetic (f64_*)."""
    tol = REL_TOL
    if golden_out is not None and key in NOISY:
        if ("nofma_" + key) in golden_out:
            tol = max(tol, 3.0 * rel_l2(golden_out["nofma_" + key], golden_out[key]))
        if ("f64_" + key) in golden_out:      # the reference's own fp32 error against exact arithmetic (fp64 autograd)
            tol = max(tol, 2.0 * rel_l2(golden_out[key], golden_out["f64_" + key]))
    return tol


def oracle_run(inp: dict, dL: Optional[np.ndarray] = None):
    from oracle import oracle
    kw = dict(bg=inp["bg"], means3D=inp["means3D"], viewmatrix=inp["viewmatrix"], projmatrix=inp["projmatrix"],
              campos=inp["campos"], tanfovx=float(inp["tanfovx"]), tanfovy=float(inp["tanfovy"]),
              scale_modifier=float(inp.get("scale_modifier", 1.0)))
    for k in ("shs", "colors_precomp", "scales", "rotations", "cov3D_precomp"):
        kw[k] = inp.get(k)
    color, radii, st = oracle.forward(opacities=inp["opacities"], image_height=int(inp["image_height"]),
                                      image_width=int(inp["image_width"]), sh_degree=int(inp["sh_degree"]), **kw)
    H, W = int(inp["image_height"]), int(inp["image_width"])
    out = dict(color=color, radii=radii, num_rendered=st.num_rendered, n_contrib=st.field("n_contrib").reshape(H, W),
               final_T=st.field("final_T").reshape(H, W), point_list=st.field("point_list"),
               ranges=st.field("ranges").reshape(-1, 2), means2D=st.field("means2D").reshape(-1, 2),
               depths=st.field("depths"), conic_opacity=st.field("conic_opacity").reshape(-1, 4),
               rgb=st.field("rgb").reshape(-1, 3), tiles_touched=st.field("tiles_touched"))
    if dL is not None:
        out.update(oracle.backward(st, dL, **kw))
    out["_st"], out["_kw"], out["_dL"], out["_opacities"], out["_sh_degree"] = st, kw, dL, inp["opacities"], int(inp["sh_degree"])
    return out


def scene_input(cloud: dict, cam, mode: str = "sh", cov_mode: str = "scale_rot") -> dict:
    from youreditableavatar_amd import scenes
    inp = dict(bg=cam.bg, means3D=cloud["means3D"], opacities=cloud["opacities"], viewmatrix=cam.viewmatrix,
               projmatrix=cam.projmatrix, campos=cam.campos, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
               image_height=cam.image_height, image_width=cam.image_width, sh_degree=cloud["sh_degree"],
               scale_modifier=cam.scale_modifier)
    if mode == "sh":
        inp["shs"] = cloud["shs"]
    else:
        inp["colors_precomp"] = scenes.sh_to_rgb_numpy(cloud["shs"], cloud["means3D"], cam.campos, cloud["sh_degree"])
    if cov_mode == "scale_rot":
        inp["scales"], inp["rotations"] = cloud["scales"], cloud["rotations"]
    else:
        inp["cov3D_precomp"] = cloud["cov3D_precomp"]
    return inp


def hip_run(inp: dict, dL: Optional[np.ndarray] = None, device="cuda:0", debug=False, introspect=True, pruning: Optional[bool] = None,
            deterministic: Optional[bool] = None, light_tiles: Optional[bool] = None, light_tiles_bwd: Optional[bool] = None,
            backward_twice: bool = False, sort_lds_cap: int = 0) -> Dict[str, np.ndarray]:
    """The HIP path through the reference's ``_C`` surface (the compiled module over the C ABI of include/tgs_raster.h).  ``pruning`` /
    ``deterministic`` / ``light_tiles``: explicit per-call options (tgs_options_t); None = the library defaults.  ``light_tiles_bwd``: another
    light-group option for the backward than the forward had; ``backward_twice``: back-propagate the same frame a second time (its result is
    returned)."""
    import torch
    from diff_gaussian_rasterization import _C
    dev = torch.device(device)
    t = lambda k: (torch.from_numpy(np.ascontiguousarray(inp[k], np.float32)).to(dev) if inp.get(k) is not None else torch.Tensor([]))
    bg, means3D, opac, view, proj, campos = t("bg"), t("means3D"), t("opacities"), t("viewmatrix"), t("projmatrix"), t("campos")
    sh, colors, scales, rots, cov = t("shs"), t("colors_precomp"), t("scales"), t("rotations"), t("cov3D_precomp")
    H, W, D = int(inp["image_height"]), int(inp["image_width"]), int(inp["sh_degree"])
    sm, tfx, tfy = float(inp.get("scale_modifier", 1.0)), float(inp["tanfovx"]), float(inp["tanfovy"])
    R, color, radii, geom, binning, img = _C.rasterize_gaussians(bg, means3D, colors, opac, scales, rots, sm, cov, view, proj,
                                                                 tfx, tfy, H, W, sh, D, campos, False, debug, pruning=pruning, light_tiles=light_tiles, sort_lds_cap=int(sort_lds_cap))
    P = means3D.shape[0]
    out = dict(color=color.cpu().numpy(), radii=radii.cpu().numpy(), num_rendered=R)
    has_sh, has_sr = inp.get("shs") is not None, inp.get("scales") is not None
    if introspect and P > 0:
        f = lambda n: _C.state_field(n, P, W, H, R, has_sh, has_sr, geom, binning, img).cpu().numpy()
        out["n_contrib"] = f("n_contrib").astype(np.uint32).reshape(H, W)
        out["final_T"] = f("final_T").reshape(H, W)
        out["ranges"] = f("ranges").astype(np.uint32).reshape(-1, 2)
        out["point_list"] = f("point_list").astype(np.uint32)
        out["means2D"] = f("means2D").reshape(-1, 2)
        out["depths"] = f("depths")
        out["conic_opacity"] = f("conic_opacity").reshape(-1, 4)
        out["tiles_touched"] = f("tiles_touched").astype(np.uint32)
        out["quad_masks"] = f("quad_masks").astype(np.uint64)
        if has_sh:
            out["rgb"] = f("rgb").reshape(-1, 3)
    if dL is not None:
        for _ in range(2 if backward_twice else 1):
            g = _C.rasterize_gaussians_backward(bg, means3D, radii, colors, scales, rots, sm, cov, view, proj, tfx, tfy,
                                                torch.from_numpy(np.ascontiguousarray(dL, np.float32)).to(dev), sh, D, campos, geom, R, binning, img, debug,
                                                _with_conic=True, deterministic=deterministic, light_tiles=light_tiles if light_tiles_bwd is None else light_tiles_bwd)
        names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations", "dL_dconic")
        out.update({n: v.cpu().numpy() for n, v in zip(names, g)})
    torch.cuda.synchronize()
    return out


def _alpha_max_in_tile(ref: dict, ids: np.ndarray, tile: int, gx: int) -> np.ndarray:
    """max over the 256 pixel centres of a tile of the alpha each Gaussian of ``ids`` would get there (float64;
    forward.cu:333-343: power > 0 and alpha < 1/255 are skipped)"""
    m2 = np.asarray(ref["means2D"], np.float64).reshape(-1, 2)[ids]
    co = np.asarray(ref["conic_opacity"], np.float64).reshape(-1, 4)[ids]
    px = (tile % gx) * 16 + np.arange(16, dtype=np.float64)
    py = (tile // gx) * 16 + np.arange(16, dtype=np.float64)
    dx = m2[:, 0, None, None] - px[None, None, :]
    dy = m2[:, 1, None, None] - py[None, :, None]
    power = -0.5 * (co[:, 0, None, None] * dx * dx + co[:, 2, None, None] * dy * dy) - co[:, 1, None, None] * dx * dy
    alpha = np.where(power > 0, 0.0, np.minimum(0.99, co[:, 3, None, None] * np.exp(np.minimum(power, 0.0))))
    return alpha.reshape(len(ids), -1).max(axis=1)


def check_point_lists(mine: dict, ref: dict, rep: dict):
    """Per-tile lists.  The HIP path gives a splat no instance in a tile of its 3-sigma rectangle where it stays below
    alpha = 1/255 on every pixel (the reference creates the instance and skips it pixel by pixel, forward.cu:340-343), so
    a tile's list must be the reference's list minus entries that provably cannot contribute, in the same order: sorted by
    (my own fp32 depth, index), and equal to the reference order wherever the depths agree bitwise (a last-bit difference
    in view-space z -- FMA contraction -- may legitimately swap two near-coincident entries)."""
    a, b = np.asarray(mine["point_list"]), np.asarray(ref["point_list"])
    rg = np.asarray(ref["ranges"]).reshape(-1, 2).astype(np.int64)
    mrg = np.asarray(mine["ranges"]).reshape(-1, 2).astype(np.int64)
    H, W = ref["n_contrib"].shape
    gx = (W + 15) // 16
    dm = np.asarray(mine["depths"]).view(np.uint32).astype(np.uint64)
    dr = np.asarray(ref["depths"]).view(np.uint32).astype(np.uint64)
    dropped = same = 0
    for t in range(rg.shape[0]):
        lb = b[rg[t, 0]:rg[t, 1]]
        la = a[mrg[t, 0]:mrg[t, 1]] if mrg[t, 1] > mrg[t, 0] else a[:0]
        if len(lb) == 0:
            assert len(la) == 0, "an instance in a tile the reference leaves empty"
            continue
        keep = np.isin(lb, la)
        assert len(la) == int(keep.sum()) and np.array_equal(np.sort(la), np.sort(lb[keep])), "a tile's list is not a subset of the reference's"
        gone = lb[~keep]
        if len(gone):
            amax = _alpha_max_in_tile(ref, gone, t, gx)
            assert np.all(amax < (1.0 / 255.0) * (1.0 - 1e-3)), f"tile {t}: dropped an instance that reaches alpha {amax.max():.6f}"
            dropped += len(gone)
        lbk = lb[keep]
        if np.array_equal(la, lbk):
            same += len(la)
            continue
        ka = (dm[la] << np.uint64(32)) | la.astype(np.uint64)
        assert np.all(ka[1:] > ka[:-1]), "a tile's list is not sorted by (depth, index)"
        diff = la != lbk
        assert np.all(dm[la[diff]] != dr[la[diff]]) or np.all(np.abs(dm[la[diff]].astype(np.int64) - dr[la[diff]].astype(np.int64)) <= 2), \
            "order differs although depths agree"
        same += int((~diff).sum())
    assert int(mine["num_rendered"]) == len(b) - dropped == len(a), "num_rendered is not the reference's minus the non-contributing instances"
    rep["lists_equal"] = same / len(a) if len(a) else 1.0
    rep["instances_dropped"] = dropped / max(len(b), 1)
    assert rep["lists_equal"] > 0.99
    if "tiles_touched" in mine and "tiles_touched" in ref:
        tm, tr = np.asarray(mine["tiles_touched"]).astype(np.int64), np.asarray(ref["tiles_touched"]).astype(np.int64)
        assert np.all(tm <= tr) and tm.sum() == len(a)


def last_contributor_ids(d: dict) -> np.ndarray:
    """Gaussian id of the last entry blended into each pixel (-1: none) from n_contrib, ranges and point_list."""
    nc = np.asarray(d["n_contrib"]).astype(np.int64)
    H, W = nc.shape
    gx = (W + 15) // 16
    start = np.asarray(d["ranges"]).reshape(-1, 2).astype(np.int64)[:, 0]
    yy, xx = np.mgrid[0:H, 0:W]
    tile = (yy // 16) * gx + xx // 16
    pl = np.asarray(d["point_list"]).astype(np.int64)
    pos = np.clip(start[tile] + nc - 1, 0, max(len(pl) - 1, 0))
    return np.where(nc > 0, pl[pos] if len(pl) else -1, -1)


def record_parity(name: str, rep: dict, extra: Optional[dict] = None) -> None:
    """Appends the achieved rel-L2 per tensor of one comparison to gpurun_out/parity.json (merged back from the GPU box; the round's
    copy is tracked as profiles/parity_rNN.json)."""
    import json
    out_dir = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, "parity.json")
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[name] = {k: (float(f"{v:.4g}") if isinstance(v, float) else v) for k, v in {**rep, **(extra or {})}.items()}
        json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass


PERGAUSS_KEYS = ("dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations")
CHAIN_TOL = 2e-5        # the per-Gaussian half against its evaluation in double on the same per-pixel gradients (those reach the oracle rounded to fp32: 6e-8 x the map's amplification)


def own_chain(mine: dict, ref: dict) -> dict:
    """the reference's per-Gaussian backward (computeCov2DCUDA / preprocessCUDA / computeCov3D backward) in double on the PRODUCT's own
    dL_dmeans2D / dL_dconic / dL_dcolors, with the fp32 oracle's state of the scene (oracle.pergauss_f64); cached in ``mine``"""
    if "_own_chain" not in mine:
        from oracle import oracle
        P = np.asarray(ref["dL_dmeans2D"]).shape[0]
        mine["_own_chain"] = oracle.pergauss_f64(ref["_st"], np.asarray(mine["dL_dmeans2D"]).reshape(P, 3), np.asarray(mine["dL_dconic"]).reshape(P, 4),
                                                np.asarray(mine["dL_dcolors"]).reshape(P, 3), **ref["_kw"])
    return mine["_own_chain"]


def reference_noise_of(ref: dict) -> dict:
    """{tensor: rel_l2(reference arithmetic in fp32, the same function in double)} of the scene behind an oracle result (tests.util.oracle_run):
    the largest of the FIVE fp32 builds of oracle/tgs_oracle.c (no FMA contraction + double accumulation / contraction + the reference's fp32
    accumulation / exp as 2^(x log2 e) like a GPU math library / the compositing loop's two cut-offs, alpha >= 1/255 and T >= 1e-4, decided
    either way for every pair inside fp32's own evaluation noise of them: tgs_oracle.c, TGS_ORACLE_CUT) against the double build.  Computed on demand
    and cached in ``ref``."""
    if "_noise" not in ref:
        from oracle import oracle
        kw = ref["_kw"]
        inp = dict(kw, opacities=ref["_opacities"], image_height=ref["n_contrib"].shape[0], image_width=ref["n_contrib"].shape[1], sh_degree=ref["_sh_degree"])
        outs = {}
        fp32_variants = ("f32_fma", "f32_ex2", "f32_in", "f32_out")
        for variant in ("f64",) + fp32_variants:
            color, _, s2 = oracle.forward(variant=variant, **inp)
            outs[variant] = dict(oracle.backward(s2, ref["_dL"], **kw), color=color)
        f64 = outs["f64"]
        ref["_noise"] = {k: max([rel_l2(ref[k], f64[k])] + [rel_l2(outs[v][k], f64[k]) for v in fp32_variants])
                         for k in ("color", "dL_dconic") + GRAD_KEYS if k in ref and k in f64}
        ref["_f64"] = f64
    return ref["_noise"]


BAR_CAP = 1e-3          # the bar never grows beyond this, however noisy the reference's own fp32 arithmetic is on a scene (round 5: frozen)
ROUTES = ("oracle", "f64", "own_chain")


def compare(mine: dict, ref: dict, golden_out: Optional[dict] = None, nc_frac: float = 0.999, check_lists: bool = True):
    """Asserts the parity bar; returns {tensor: rel_l2} for reporting.

    The bar (FROZEN in round 5: no new routes, no new oracle builds), for the colour and every gradient:
        bar = min(max(1e-4, 2 x eta), BAR_CAP = 1e-3)
    eta = the reference arithmetic's own distance from exact arithmetic on this very scene (reference_noise_of: the largest of five fp32
    builds of the oracle's C text against the same text compiled in double).  A tensor passes by ONE of three routes, recorded per tensor as
    `<tensor>|route`:
      "oracle"     rel_l2(HIP, fp32 oracle) <= bar                                   (every tensor at every BASELINE configuration: <= 1e-4)
      "f64"        rel_l2(HIP, the double evaluation) <= bar -- the fp32 oracle is ONE rounding of the reference's function; a result within
                   the bar of that function in exact arithmetic is as good a rounding of it as the bar allows the reference to be
      "own_chain"  the four per-Gaussian tensors only: within CHAIN_TOL of the reference's per-Gaussian half evaluated in double on the
                   product's own per-pixel gradients, which must pass the bar themselves
    `routes_beyond_oracle` counts the tensors of the scene that needed the second or third route; the fuzz test holds a budget on the
    scenes that do (tests/test_gpu_parity.py).  Against a FIXTURE the bar is what the fixture recorded (tolerance()): fixed, no routes."""
    H, W = ref["n_contrib"].shape
    rep = {}
    assert int(mine["num_rendered"]) <= int(ref["num_rendered"]), "more instances than the reference"   # exact relation: check_point_lists
    vis = np.asarray(ref["radii"]) > 0
    assert np.array_equal(np.asarray(mine["radii"]), np.asarray(ref["radii"])), "radii differ"
    if check_lists and "point_list" in mine:
        check_point_lists(mine, ref, rep)
    if "n_contrib" in mine and "point_list" in mine and "point_list" in ref:
        # n_contrib counts list positions, and the lists here lack the reference's non-contributing instances: compare WHO the
        # last contributor of every pixel is
        frac = float((last_contributor_ids(mine) == last_contributor_ids(ref)).mean())
        rep["n_contrib_equal"] = frac
        assert frac >= nc_frac, f"last contributor equal on {frac:.5f} of pixels"
    against_oracle = "_st" in ref

    def bar(k: str) -> float:
        if against_oracle:
            return min(max(REL_TOL, 2.0 * reference_noise_of(ref).get(k, 0.0)), BAR_CAP)
        return tolerance(k, golden_out)

    e = rel_l2(mine["color"], ref["color"]); rep["color"] = e
    assert e <= REL_TOL or e <= bar("color"), f"color rel-L2 {e:.3e}"
    keys = GRAD_KEYS + (("dL_dconic",) if against_oracle and "dL_dconic" in mine else ())
    beyond = 0
    for k in keys:
        if k in mine and k in ref:
            a, b = np.asarray(mine[k]), np.asarray(ref[k])
            if k == "dL_dmeans2D":
                assert np.all(a[:, 2] == 0)
            if k == "dL_dconic":
                a = a.reshape(-1, 4)
            e = rel_l2(a, b); rep[k] = e
            if e > REL_TOL:                                  # only then is the scene's own noise floor needed (five more oracle runs)
                tol = bar(k)
                rep[k + "|bar"] = tol
                if against_oracle:
                    e64 = rep[k + "|vs_f64"] = rel_l2(a, ref["_f64"][k])
                    if e <= tol:
                        route = "oracle"
                    elif e64 <= tol:
                        route = "f64"
                    elif k in PERGAUSS_KEYS:
                        # Third route, per-Gaussian tensors only: dL_dmeans3D / dL_dcov3D / dL_dscales / dL_drotations are a function of the
                        # per-pixel pass's outputs (dL_dmeans2D, dL_dconic, dL_dcolors), and for splats several times wider than the image that
                        # function multiplies their fp32 noise by hundreds.  When those three pass the bar themselves (asserted in this loop) and
                        # the tensor equals the reference's per-Gaussian half evaluated IN DOUBLE on the product's own three, the distance is
                        # amplified input noise that is inside the bar, not an error of this half.
                        ech = rep[k + "|vs_own_chain"] = rel_l2(a, own_chain(mine, ref)[k])
                        assert ech <= CHAIN_TOL, (f"{k}: rel-L2 to the fp32 oracle {e:.3e}, to exact arithmetic {e64:.3e} (bar {tol:.2e}) and to the per-Gaussian half "
                                                  f"in double on the product's own per-pixel gradients {ech:.3e} (bar {CHAIN_TOL:.0e})")
                        route = "own_chain"
                    else:
                        raise AssertionError(f"{k}: rel-L2 to the fp32 oracle {e:.3e} and to exact arithmetic {e64:.3e} both exceed {tol:.2e} = "
                                             "min(max(1e-4, 2 x the reference arithmetic's own distance from exact arithmetic), 1e-3)")
                    rep[k + "|route"] = route
                    beyond += route != "oracle"
                else:
                    assert e <= tol, f"{k} rel-L2 {e:.3e} > {tol:.2e}"
            if k != "dL_dconic":                             # (on every route)
                assert np.all(a[~vis] == 0), f"{k}: culled Gaussians must have zero gradient"
    if against_oracle:
        rep["routes_beyond_oracle"] = beyond
    return rep


# ---- mid-size goldens (tests/golden/m*.npz, oracle/emu_crosscheck/make_mid_goldens.py): recipe + outputs on a seeded subset ----
def mid_golden_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "m[0-9]*.npz")))


def load_mid_golden(name: str):
    """-> (inp, dL, fixture): the inputs are regenerated from the fixture's recipe by the seeded generators of scenes.py and checked
    against the stored checksum (a fixture made from other inputs would pin nothing)."""
    import zlib
    from youreditableavatar_amd import scenes
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    r = {k[7:]: z[k].item() for k in z.files if k.startswith("recipe_")}
    cloud = scenes.make_cloud(int(r["P"]), int(r["deg"]), seed=int(r["seed"]), scale_mult=float(r["scale_mult"]))
    cam = scenes.orbit_camera(int(r["W"]), int(r["H"]), azimuth_deg=float(r["az"]))
    inp = scene_input(cloud, cam)
    dL = scenes.upstream_gradient(int(r["W"]), int(r["H"]), seed=int(r["dl_seed"]))
    c = 0
    for k in sorted(inp):
        if isinstance(inp[k], np.ndarray):
            c = zlib.crc32(np.ascontiguousarray(inp[k]).tobytes(), c)
    c = zlib.crc32(np.ascontiguousarray(dL).tobytes(), c)
    assert np.uint32(c) == z["input_crc"], f"{name}: the regenerated inputs are not the ones the fixture was made from"
    return inp, dL, {k: z[k] for k in z.files}


def list_checksums(point_list, ranges) -> np.ndarray:
    import zlib
    rg = np.asarray(ranges).reshape(-1, 2)
    out = np.zeros(len(rg), np.uint32)
    for t, (a, b) in enumerate(rg):
        if b > a:
            out[t] = zlib.crc32(np.ascontiguousarray(point_list[a:b], np.uint32).tobytes())
    return out


def compare_mid(mine: dict, fx: dict, exact_lists: bool, nc_frac: float = 0.999, list_frac: float = 0.99) -> dict:
    """mine (oracle_run / hip_run output) against a mid-size fixture; exact_lists: the integer state must equal the reference's
    (the oracle, or the HIP path with instance pruning off), otherwise it must be the pruned subset."""
    rep = {}
    idx = fx["subset"]
    y0, x0, wh, ww = [int(v) for v in fx["window"]]
    assert np.array_equal(np.asarray(mine["radii"]), fx["out_radii"]), "radii differ"
    if exact_lists:
        assert int(mine["num_rendered"]) == int(fx["out_num_rendered"])
        assert np.array_equal(np.asarray(mine["tiles_touched"]).astype(np.uint32), fx["out_tiles_touched"])
        mr, fr = np.asarray(mine["ranges"]).reshape(-1, 2).astype(np.int64), fx["out_ranges"].reshape(-1, 2).astype(np.int64)
        ne = fr[:, 1] > fr[:, 0]                          # (the reference leaves an empty tile at {0, 0}, rasterizer_impl.cu:310; here it is [start, start))
        assert np.array_equal(mr[ne], fr[ne]) and np.all(mr[~ne, 1] == mr[~ne, 0])
        rep["n_contrib_equal"] = float((np.asarray(mine["n_contrib"]).astype(np.int64) == fx["out_n_contrib"].astype(np.int64)).mean())
        assert rep["n_contrib_equal"] >= nc_frac
        crc = list_checksums(mine["point_list"], mine["ranges"])
        rep["tile_lists_equal"] = float((crc == fx["out_list_crc"]).mean())
        assert rep["tile_lists_equal"] >= list_frac
    else:
        assert int(mine["num_rendered"]) <= int(fx["out_num_rendered"])
        assert np.all(np.asarray(mine["tiles_touched"]).astype(np.int64) <= fx["out_tiles_touched"].astype(np.int64))
    col = np.asarray(mine["color"], np.float64)
    rep["color_window"] = rel_l2(col[:, y0:y0 + wh, x0:x0 + ww], fx["out_color_window"])
    assert rep["color_window"] <= REL_TOL
    assert np.allclose(col.sum(axis=(1, 2)), fx["out_color_sum"], rtol=1e-5) and np.allclose((col * col).sum(axis=(1, 2)), fx["out_color_sumsq"], rtol=1e-5)
    if "final_T" in mine:
        assert rel_l2(np.asarray(mine["final_T"])[y0:y0 + wh, x0:x0 + ww], fx["out_final_T_window"]) <= 1e-4
    vis = fx["out_radii"][idx] > 0
    for k, tol in (("means2D", 1e-6), ("depths", 1e-6), ("conic_opacity", 1e-4)):
        if k in mine:
            assert rel_l2(np.asarray(mine[k])[idx][vis], fx["out_" + k + "_subset"][vis]) <= tol, k
    for k in GRAD_KEYS + ("dL_dconic",):
        if k not in mine:
            continue
        a, b = np.asarray(mine[k]).reshape(len(fx["out_radii"]), -1)[idx], fx["out_" + k + "_subset"].reshape(len(idx), -1)
        tol = REL_TOL
        if k in NOISY:
            tol = max(tol, 3.0 * rel_l2(fx["out_nofma_" + k + "_subset"].reshape(len(idx), -1), b))
        e = rel_l2(a, b); rep[k] = e
        assert e <= tol, f"{k} (subset of {len(idx)} Gaussians) rel-L2 {e:.3e} > {tol:.1e}"
        n = float(np.linalg.norm(np.asarray(mine[k], np.float64)))
        assert abs(n - float(fx["out_" + k + "_norm"])) <= 2 * tol * max(float(fx["out_" + k + "_norm"]), 1e-30), f"{k}: full-tensor norm"
    return rep
