"""CPU: the oracle (oracle/tgs_oracle.c) against every committed golden vector, and against the
independent fp64 autograd splat.  These pin the checker that the GPU parity tests rely on."""
import numpy as np
import pytest

from tests import util


@pytest.mark.parametrize("name", util.golden_names())
def test_oracle_matches_golden(name):
    inp, gold = util.load_golden(name)
    H, W = int(inp["image_height"]), int(inp["image_width"])
    mine = util.oracle_run(inp, inp["dL_dout_color"])
    ref = dict(gold)
    ref["n_contrib"] = gold["n_contrib"].reshape(H, W)
    # integer/index state is bit-exact
    assert mine["num_rendered"] == int(gold["num_rendered"])
    assert np.array_equal(mine["radii"], gold["radii"])
    assert np.array_equal(mine["point_list"], gold["point_list"])
    assert np.array_equal(mine["n_contrib"], ref["n_contrib"])
    assert np.array_equal(mine["tiles_touched"], gold["tiles_touched"])
    util.compare(mine, ref, gold, nc_frac=1.0)
    vis = gold["radii"] > 0
    assert util.rel_l2(mine["means2D"][vis], gold["means2D"][vis]) <= 1e-6
    assert util.rel_l2(mine["conic_opacity"][vis], gold["conic_opacity"][vis]) <= 1e-5
    assert util.rel_l2(mine["final_T"], gold["final_T"]) <= 1e-5


@pytest.mark.parametrize("name", util.mid_golden_names())
def test_oracle_matches_mid_size_golden(name):
    """20 000 Gaussians at 256x256, BASELINE config 2 (100 000 Gaussians, 800x800, SH 3) and config 3 -- the headline workload: 500 000
    Gaussians, 1920x1080, SH 3 -- at full size: the oracle against the reference's
    kernel text under the emulation -- integer state (radii, num_rendered, tiles_touched, ranges, n_contrib) bit-exact, every tile's list
    equal except where two depths differ in the last bit between the FMA-contracted emulation build and the oracle (1 tile of 2500 at
    config 2: two entries whose view-space z is equal in one build and one ulp apart in the other), image and gradients (on the fixture's
    subset of Gaussians) within the parity bar."""
    inp, dL, fx = util.load_mid_golden(name)
    mine = util.oracle_run(inp, dL)
    # config 3 at full size: 31 of 8160 tile lists hold such a pair and 5 of 2 073 600 pixels end on one of them -- exactly the difference
    # between the emulation's own two builds (FMA contraction on / off: n_contrib equal on 0.999998 of the pixels)
    rep = util.compare_mid(mine, fx, exact_lists=True, nc_frac=0.99999, list_frac=0.995)
    assert name == "m03_cfg3_1080p" or (rep["n_contrib_equal"] == 1.0 and rep["tile_lists_equal"] >= 0.999)
    print(name, {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in rep.items()})


def test_goldens_cover_the_edge_cases():
    names = util.golden_names()
    assert len(names) >= 14
    inp, gold = util.load_golden("g10_all_culled")
    assert int(gold["num_rendered"]) == 0 and np.all(gold["radii"] == 0)
    inp, gold = util.load_golden("g09_giant_splat")
    T = ((int(inp["image_width"]) + 15) // 16) * ((int(inp["image_height"]) + 15) // 16)
    assert gold["tiles_touched"].max() == T           # one splat covers every tile
    inp, gold = util.load_golden("g02_sh0_nonmult16")
    assert int(inp["image_width"]) % 16 != 0 and int(inp["image_height"]) % 16 != 0
    inp, gold = util.load_golden("g07_depth_ties")
    d = gold["depths"][gold["radii"] > 0]
    assert len(np.unique(d)) < len(d)                  # exact ties exist


@pytest.mark.parametrize("P,W,H,deg,mode,sm", [(300, 64, 48, 3, "sh", 6.0), (400, 64, 64, 1, "precomp", 8.0)])
def test_oracle_matches_fp64_autograd(P, W, H, deg, mode, sm):
    """Independent check of the analytic backward: textbook forward differentiated by autograd in fp64."""
    from oracle import oracle, torch_splat
    from youreditableavatar_amd import scenes
    cloud = scenes.make_cloud(P, deg, seed=P, scale_mult=sm)
    cam = scenes.orbit_camera(W, H, azimuth_deg=30)
    dL = scenes.upstream_gradient(W, H)
    color, radii, st, g = oracle.run_scene(cloud, cam, dL, mode=mode)
    r = torch_splat.run_scene(cloud, cam, dL, mode=mode)
    assert st.num_rendered == r["num_rendered"]
    assert np.array_equal(radii, r["radii"])
    assert (st.field("n_contrib").reshape(H, W) == r["n_contrib"]).mean() >= 0.999
    assert util.rel_l2(color, r["color"]) <= 1e-5
    pairs = [("dL_dmeans3D", "grad_means3D"), ("dL_dmeans2D", "grad_means2D"), ("dL_dopacity", "grad_opacities"),
             ("dL_dscales", "grad_scales"), ("dL_drotations", "grad_rotations")]
    pairs += [("dL_dsh", "grad_shs")] if mode == "sh" else [("dL_dcolors", "grad_colors_precomp")]
    for a, b in pairs:
        assert util.rel_l2(g[a], r[b]) <= 2e-5, (a, util.rel_l2(g[a], r[b]))


def test_oracle_mark_visible():
    from oracle import oracle
    from youreditableavatar_amd import scenes
    cloud = scenes.make_cloud(500, 0, seed=2)
    cam = scenes.orbit_camera(64, 64)
    m = cloud["means3D"].copy()
    m[::2] = m[::2] * 0.1 + cam.campos * 2.0
    vis = oracle.mark_visible(m, cam.viewmatrix, cam.projmatrix)
    z = (np.concatenate([m, np.ones((500, 1), np.float32)], 1) @ cam.viewmatrix)[:, 2]
    assert np.array_equal(vis, z > 0.2)
    assert not vis[::2].any() and vis[1::2].all()


def test_oracle_cutoff_builds_move_only_undecidable_pairs():
    """The two cut-off builds of the oracle (tgs_oracle.c: TGS_ORACLE_CUT; part of the reference-noise measure of the parity bar) against the
    plain fp32 build on the two fuzz scenes that prompted them: the build that decides pairs inside fp32's noise band of alpha >= 1/255 as
    BLENDED moves dL_dconic by what one edge pair of a large splat is worth (6.7e-4 / 1.1e-4), the build that decides them as SKIPPED is the
    plain build there (those pairs are skipped by it already) -- and on a scene without such a pair all three are the same function."""
    from tests import fuzz, adjudicate
    for seed, scene, lo, hi in ((23, 93, 5e-4, 8e-4), (37, 89, 8e-5, 2e-4)):
        rng = np.random.default_rng(seed)
        for it in range(scene + 1):
            desc, inp, dL = fuzz.random_scene(rng, it)
        plain, more, fewer = (adjudicate.oracle_variant(inp, dL, v) for v in ("f32", "f32_in", "f32_out"))
        d_in, d_out = util.rel_l2(more["dL_dconic"], plain["dL_dconic"]), util.rel_l2(fewer["dL_dconic"], plain["dL_dconic"])
        assert lo <= d_in <= hi and d_out <= 1e-6, (desc, d_in, d_out)
    inp, gold = util.load_golden(util.golden_names()[0])
    dL = inp["dL_dout_color"]
    plain, more, fewer = (adjudicate.oracle_variant(inp, dL, v) for v in ("f32", "f32_in", "f32_out"))
    for k in ("color", "dL_dconic", "dL_dmeans2D"):
        assert util.rel_l2(more[k], plain[k]) <= 1e-4 and util.rel_l2(fewer[k], plain[k]) <= 1e-4, k
