"""GPU parity tests proper: the HIP path (through the C ABI) against the committed goldens and the
CPU oracle on identical inputs.  Run on an MI355X with ``pytest -m gpu``."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", util.golden_names())
def test_golden(name, gpu_device):
    inp, gold = util.load_golden(name)
    mine = util.hip_run(inp, inp["dL_dout_color"])
    H, W = int(inp["image_height"]), int(inp["image_width"])
    ref = dict(gold)
    ref["n_contrib"] = gold["n_contrib"].reshape(H, W)
    rep = util.compare(mine, ref, gold)
    print(name, {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in rep.items()})
    # per-Gaussian intermediates of visible Gaussians
    vis = gold["radii"] > 0
    assert util.rel_l2(mine["means2D"][vis], gold["means2D"][vis]) <= 1e-6
    assert util.rel_l2(mine["depths"][vis], gold["depths"][vis]) <= 1e-6
    assert util.rel_l2(mine["conic_opacity"][vis], gold["conic_opacity"][vis]) <= 1e-4
    # tile bookkeeping: lists are the reference's minus the instances that cannot contribute (util.check_point_lists, called
    # by compare): never more per Gaussian or per tile, ranges contiguous, empty tiles of the reference stay empty
    assert np.all(mine["tiles_touched"] <= gold["tiles_touched"])
    gr, mr = gold["ranges"].reshape(-1, 2).astype(np.int64), mine["ranges"].reshape(-1, 2).astype(np.int64)
    assert np.all((mr[:, 1] - mr[:, 0]) <= (gr[:, 1] - gr[:, 0]))
    ne = mr[:, 1] > mr[:, 0]
    assert np.all(mr[ne][1:, 0] == mr[ne][:-1, 1]) and (not ne.any() or (mr[ne][0, 0] == 0 and mr[ne][-1, 1] == mine["num_rendered"]))


@pytest.mark.parametrize("name", util.mid_golden_names())
def test_mid_size_golden(name, gpu_device):
    """The HIP path against the mid-size emulated-reference fixtures (20k Gaussians / 256x256; BASELINE config 2 at 800x800): with instance
    pruning off the integer state is the reference's (num_rendered, tiles_touched, ranges; n_contrib and the per-tile lists up to the
    near-coincident depths a last-bit difference in view-space z may swap); with the default pruning image and gradients."""
    inp, dL, fx = util.load_mid_golden(name)
    full = util.hip_run(inp, None, pruning=False)           # explicit per-call option (tgs_options_t), no process-wide knob
    rep = util.compare_mid(full, fx, exact_lists=True)
    mine = util.hip_run(inp, dL)
    rep.update(util.compare_mid(mine, fx, exact_lists=False))
    util.record_parity(name, rep)
    print(name, {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in rep.items()})


@pytest.mark.parametrize("P,W,H,deg,mode,cov_mode,scale_mult", [
    (10_000, 256, 256, 0, "sh", "scale_rot", 1.0),       # BASELINE config 1
    (20_000, 320, 200, 3, "sh", "scale_rot", 2.0),
    (20_000, 300, 300, 2, "precomp", "scale_rot", 3.0),
    (5_000, 200, 120, 3, "sh", "cov3d", 4.0),
    (4_000, 320, 240, 1, "sh", "scale_rot", 8.0),        # splats on 16 .. 127 tiles: every per-wave threshold of the slab sums' cooperative pass (round 5)
])
def test_vs_oracle_seeded(P, W, H, deg, mode, cov_mode, scale_mult, gpu_device):
    from youreditableavatar_amd import scenes
    from oracle.emu_crosscheck_cov import cov3d_from
    cloud = scenes.make_cloud(P, deg, seed=7 + P + W, scale_mult=scale_mult)
    if cov_mode == "cov3d":
        cloud["cov3D_precomp"] = cov3d_from(cloud["scales"], cloud["rotations"])
    cam = scenes.orbit_camera(W, H, azimuth_deg=40.0)
    inp = util.scene_input(cloud, cam, mode, cov_mode)
    dL = scenes.upstream_gradient(W, H, seed=5)
    ref = util.oracle_run(inp, dL)
    mine = util.hip_run(inp, dL)
    rep = util.compare(mine, ref)
    if (P, W, H, deg) == (10_000, 256, 256, 0):
        util.record_parity("cfg1", rep, extra=dict(num_rendered=int(mine["num_rendered"]), num_rendered_reference=int(ref["num_rendered"])))
    print({k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in rep.items()})


def test_spatially_ordered_cloud_vs_oracle(gpu_device):
    """Gaussians numbered along a Morton curve (index neighbours are spatial neighbours, the way mesh-bound Gaussians come): many lanes
    of a wave of k_bin_count / k_scatter then add to the SAME counter of the chunk's LDS table in one instruction, which a randomly
    ordered cloud almost never does.  Lists, image and gradients against the oracle."""
    from youreditableavatar_amd import scenes
    cloud = scenes.morton_order(scenes.make_cloud(60_000, 2, seed=314, scale_mult=1.5))
    cam = scenes.orbit_camera(640, 400, azimuth_deg=65.0)
    inp = util.scene_input(cloud, cam)
    dL = scenes.upstream_gradient(640, 400, seed=15)
    ref = util.oracle_run(inp, dL)
    mine = util.hip_run(inp, dL)
    rep = util.compare(mine, ref)
    assert rep["lists_equal"] > 0.999
    util.record_parity("morton_ordered_60k", rep)
    # how often index neighbours share their first tile here (the combined path needs >= 6 such lanes in a wave)
    m2 = ref["means2D"]; vis = ref["radii"] > 0
    t = (np.floor(m2[:, 0] / 16).astype(np.int64) + 40 * np.floor(m2[:, 1] / 16).astype(np.int64))
    same = (t[1:] == t[:-1]) & vis[1:] & vis[:-1]
    assert same.mean() > 0.3


def test_tile_grid_beyond_one_lds_table(gpu_device):
    """26 125 tiles (3344 x 2000): more than the 24 576 counters one pass of the binning chunks' LDS table holds, so k_bin_count and
    k_scatter walk the tile grid in two bands (and ask for 96 KB of dynamic LDS).  Splats of every emission class -- <= 4 tiles, 5..64
    (spread over the wave), larger (walked by the wave) -- against the oracle."""
    from youreditableavatar_amd import scenes
    W, H = 3344, 2000
    cloud = scenes.make_cloud(6_000, 1, seed=2611, scale_mult=0.6)
    cloud["means3D"] = cloud["means3D"].copy(); cloud["means3D"][:, 1] -= 0.5        # towards the bottom of the image: the last band is not empty
    cloud["scales"] = cloud["scales"].copy(); cloud["scales"][::150] *= 6.0           # some splats on more than 64 tiles
    cam = scenes.orbit_camera(W, H, azimuth_deg=20.0)
    inp = util.scene_input(cloud, cam)
    dL = scenes.upstream_gradient(W, H, seed=26)
    ref = util.oracle_run(inp, dL)
    mine = util.hip_run(inp, dL)
    rep = util.compare(mine, ref)
    tt = ref["tiles_touched"]
    assert (tt > 64).sum() > 50 and ((tt > 4) & (tt <= 64)).sum() > 1000 and ((tt > 0) & (tt <= 4)).sum() > 100
    r = ref["ranges"].reshape(-1, 2)
    assert len(r) > 24576 and (r[24576:, 1] > r[24576:, 0]).sum() > 500       # the second band holds instances
    assert rep["lists_equal"] == 1.0
    print({k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in rep.items()})


def test_backward_is_bitwise_reproducible(gpu_device):
    """Deterministic mode: no float atomics anywhere, two runs give identical bits (the reference's do not)."""
    from youreditableavatar_amd import scenes
    cloud = scenes.make_cloud(20_000, 3, seed=3, scale_mult=3.0)
    cam = scenes.orbit_camera(256, 192)
    inp = util.scene_input(cloud, cam)
    dL = scenes.upstream_gradient(256, 192)
    a = util.hip_run(inp, dL, introspect=False, deterministic=True)
    b = util.hip_run(inp, dL, introspect=False, deterministic=True)
    c = util.hip_run(inp, dL, introspect=False, deterministic=False)
    for k in ("color",) + util.GRAD_KEYS:
        assert np.array_equal(a[k], b[k]), k
        assert util.rel_l2(c[k], a[k]) <= 1e-5, k          # the default (LDS-atomic) kernel agrees with it


@pytest.mark.parametrize("name", ["g01_sh3_scale_rot", "g08_opaque_termination", "g09_giant_splat", "g13_dense_2k"])
def test_golden_deterministic_kernel(name, gpu_device):
    inp, gold = util.load_golden(name)
    mine = util.hip_run(inp, inp["dL_dout_color"], deterministic=True)
    ref = dict(gold)
    ref["n_contrib"] = gold["n_contrib"].reshape(int(inp["image_height"]), int(inp["image_width"]))
    util.compare(mine, ref, gold)


def test_wave_reduce36_on_hardware(gpu_device):
    """The 36-value wave reduction (v_permlane32/16_swap + DPP) against a float64 sum."""
    import torch
    from diff_gaussian_rasterization import _C
    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, 36, generator=g)
    out = _C.selftest_reduce36(x.to(gpu_device)).cpu().double()
    ref = x.double().sum(0).reshape(4, 9)
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-5), (out - ref).abs().max()


@pytest.mark.parametrize("cap", [2, 64, 512])
def test_tile_list_overflow_path(cap, gpu_device):
    """Lists longer than the LDS budget are sorted in global memory by many workgroups; with the budget
    lowered to ``cap`` entries nearly every tile takes that path.  Results must not change."""
    from youreditableavatar_amd import scenes
    cloud = scenes.make_cloud(20_000, 1, seed=77, scale_mult=3.0, n_oversized=20, oversize=30.0)
    cam = scenes.orbit_camera(200, 136, azimuth_deg=10.0)
    inp = util.scene_input(cloud, cam)
    dL = scenes.upstream_gradient(200, 136, seed=6)
    ref = util.oracle_run(inp, dL)
    lens = ref["ranges"][:, 1] - ref["ranges"][:, 0]
    assert lens.max() > 512                      # the scene really has long lists
    mine = util.hip_run(inp, dL, sort_lds_cap=cap)
    util.compare(mine, ref)
    base = util.hip_run(inp, dL)
    assert np.array_equal(mine["point_list"], base["point_list"])
    assert np.array_equal(mine["color"], base["color"])


def _one_tile_pileup(n_blob, width=208, height=144, P_extra=3000):
    """n_blob splats piled into one tile (centre of the image = inside tile (6, 4) of 13 x 9) + an ordinary cloud around them"""
    from youreditableavatar_amd import scenes
    cloud = scenes.concentrate(scenes.make_cloud(n_blob + P_extra, 1, seed=91, scale_mult=2.0), n_blob, centre=(0.01, -0.01, 0.0), sigma=0.003)
    cam = scenes.orbit_camera(width, height, azimuth_deg=20.0)
    return cloud, cam


@pytest.mark.parametrize("n_blob", [12_000, 40_000])
def test_real_overflow_lists_vs_oracle(n_blob, gpu_device):
    """A tile list longer than the LDS sort (8192 keys) at the DEFAULT budget -- 12k: one merge level above it, 40k: three -- goes through
    k_tile_sort's overflow workers (global-memory bitonic network, grid barrier between steps) at their real sizes; lists, images and
    gradients against the oracle like any other scene."""
    from youreditableavatar_amd import scenes
    cloud, cam = _one_tile_pileup(n_blob)
    inp = util.scene_input(cloud, cam)
    dL = scenes.upstream_gradient(cam.image_width, cam.image_height, seed=8)
    ref = util.oracle_run(inp, dL)
    lens = ref["ranges"][:, 1] - ref["ranges"][:, 0]
    assert lens.max() > n_blob * 0.9 and lens.max() > 8192
    mine = util.hip_run(inp, dL)
    mlens = mine["ranges"][:, 1].astype(np.int64) - mine["ranges"][:, 0]
    assert mlens.max() > 8192                                   # still beyond the LDS sort after instance pruning
    rep = util.compare(mine, ref)
    util.record_parity(f"one_tile_pileup_{n_blob}", rep, extra=dict(longest_list=int(mlens.max()), longest_list_reference=int(lens.max())))
    assert rep["lists_equal"] > 0.999
    again = util.hip_run(inp, dL)                               # the grid-barrier network is deterministic
    assert np.array_equal(mine["point_list"], again["point_list"]) and np.array_equal(mine["color"], again["color"])


def test_instance_pruning_off_gives_the_reference_lists(gpu_device):
    """tgs_set_instance_pruning(0): every tile of the 3-sigma rectangle gets its instance like in the reference
    (rasterizer_impl.cu:98-109) -- num_rendered, tiles_touched, ranges and n_contrib equal the golden state; the image and the
    gradients are the pruned path's up to summation order."""
    inp, gold = util.load_golden("g13_dense_2k")
    H, W = int(inp["image_height"]), int(inp["image_width"])
    pruned = util.hip_run(inp, inp["dL_dout_color"], deterministic=True)
    full = util.hip_run(inp, inp["dL_dout_color"], deterministic=True, pruning=False)
    assert full["num_rendered"] == int(gold["num_rendered"]) > pruned["num_rendered"]
    assert np.array_equal(full["tiles_touched"], gold["tiles_touched"])
    assert (full["n_contrib"] == gold["n_contrib"].reshape(H, W)).mean() >= 0.999
    gr = gold["ranges"].reshape(-1, 2)
    ne = gr[:, 1] > gr[:, 0]
    assert np.array_equal(full["ranges"][ne], gr[ne])
    # same contributions, but the 512-entry rounds and 4-entry groups fall elsewhere in the longer lists: another summation order
    assert util.rel_l2(full["color"], pruned["color"]) <= 1e-6
    for k in util.GRAD_KEYS:
        assert util.rel_l2(full[k], pruned[k]) <= 2e-5, k


def test_instance_pruning_off_with_large_splats(gpu_device):
    """The same switch on a scene whose splats mostly cover 5..64 tiles (64-bit live-tile masks, wave-cooperative k_scatter) and
    some more than 64: with pruning off every tile of every rectangle has its instance -- num_rendered, tiles_touched and the tile
    ranges equal the oracle's (the reference's definition); with pruning on the lists are the oracle's minus non-contributors."""
    from youreditableavatar_amd import scenes
    cloud = scenes.make_cloud(20_000, 2, seed=97, scale_mult=4.0, n_oversized=5, oversize=20.0)
    cam = scenes.orbit_camera(320, 208, azimuth_deg=65.0)
    inp = util.scene_input(cloud, cam)
    dL = scenes.upstream_gradient(320, 208, seed=4)
    ref = util.oracle_run(inp, dL)
    tt = ref["tiles_touched"]
    assert ((tt > 4) & (tt <= 64)).sum() > 2000 and (tt > 64).sum() >= 5
    pruned = util.hip_run(inp, dL)
    util.compare(pruned, ref)
    full = util.hip_run(inp, dL, pruning=False)
    assert full["num_rendered"] == ref["num_rendered"] > pruned["num_rendered"]
    assert np.array_equal(full["tiles_touched"], ref["tiles_touched"])
    assert np.array_equal(full["ranges"].reshape(-1, 2), ref["ranges"].reshape(-1, 2))
    assert util.rel_l2(full["color"], pruned["color"]) <= 1e-6
    for k in util.GRAD_KEYS:
        assert util.rel_l2(full[k], pruned[k]) <= 5e-5, k


@pytest.mark.parametrize("name", ["g01_sh3_scale_rot", "g09_giant_splat", "g13_dense_2k"])
def test_quadrant_masks_are_conservative(name, gpu_device):
    """The render kernels visit an entry only in the 2x2-pixel quadrants its 64-bit mask names (k_finalize, bit 8*row + column of the
    tile's 8x8 quadrant grid): every pixel where the entry passes the reference's tests (forward.cu:336-343, evaluated in float64 on the
    kernel's own means2D / conic_opacity) must lie in a named quadrant.  Also reports how tight the masks are."""
    inp, gold = util.load_golden(name)
    mine = util.hip_run(inp)
    W, H = int(inp["image_width"]), int(inp["image_height"])
    gx = (W + 15) // 16
    rg = mine["ranges"].astype(np.int64)
    pl, qm = mine["point_list"].astype(np.int64), mine["quad_masks"]
    m2, co = mine["means2D"].astype(np.float64), mine["conic_opacity"].astype(np.float64)
    named = alive_total = missed = 0
    for tile in np.nonzero(rg[:, 1] > rg[:, 0])[0]:
        ids = pl[rg[tile, 0]:rg[tile, 1]]
        px = (tile % gx) * 16 + np.arange(16, dtype=np.float64)
        py = (tile // gx) * 16 + np.arange(16, dtype=np.float64)
        dx = m2[ids, 0, None, None] - px[None, None, :]
        dy = m2[ids, 1, None, None] - py[None, :, None]
        power = -0.5 * (co[ids, 0, None, None] * dx * dx + co[ids, 2, None, None] * dy * dy) - co[ids, 1, None, None] * dx * dy
        alpha = np.minimum(0.99, co[ids, 3, None, None] * np.exp(np.minimum(power, 0.0)))
        alive = (power <= 0) & (alpha >= 1.0 / 255.0)                                   # [n, y, x]
        alive_q = alive.reshape(-1, 8, 2, 8, 2).any(axis=(2, 4)).reshape(-1, 64)        # [n, 8*row + column]
        bits = ((qm[rg[tile, 0]:rg[tile, 1], None] >> np.arange(64, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(bool)
        missed += int((alive_q & ~bits).sum())
        named += int(bits.sum()); alive_total += int(alive_q.sum())
    print(name, f"quadrants named {named}, with a live pixel {alive_total} ({alive_total / max(named, 1):.3f})")
    assert missed == 0, f"{missed} quadrants with a contributing pixel are not in the mask"
    assert alive_total >= 0.9 * named                    # and the masks are tight: a quadrant is named only near the footprint


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16, 17])
def test_random_small_scenes_vs_oracle(seed, gpu_device):
    """A slice of tests/tools/fuzz_vs_oracle.py with fixed seeds: random sizes (not multiples of 16), SH degrees, scale multipliers,
    flat / tiny / oversized splats, camera angles -- the full parity bar of util.compare, pruned tile lists included."""
    from youreditableavatar_amd import scenes
    rng = np.random.default_rng(seed)
    P = int(rng.integers(50, 5000)); W = int(rng.integers(17, 280)); H = int(rng.integers(17, 200)); D = int(rng.integers(0, 4))
    sm = float(rng.choice([0.3, 1.0, 3.0, 8.0])); ff = float(rng.uniform(0, 1)); tf = float(rng.choice([0.0, 0.05]))
    cloud = scenes.make_cloud(P, D, seed=int(rng.integers(1 << 30)), scale_mult=sm, flat_fraction=ff, tiny_fraction=tf, n_oversized=int(rng.choice([0, 0, 3])))
    cam = scenes.orbit_camera(W, H, azimuth_deg=float(rng.uniform(0, 360)), elevation_deg=float(rng.uniform(-30, 30)))
    inp = util.scene_input(cloud, cam)
    dL = scenes.upstream_gradient(W, H, seed=seed)
    rep = util.compare(util.hip_run(inp, dL), util.oracle_run(inp, dL))
    print(seed, P, W, H, D, sm, {k: f"{v:.1e}" for k, v in rep.items() if k in ("color", "lists_equal", "instances_dropped", "n_contrib_equal")})


@pytest.mark.parametrize("mode,seed", [("sh", 91), ("precomp", 92), ("sh", "cfg1"), ("precomp", "cfg1")])
def test_hip_vs_independent_fp64_autograd(mode, seed, gpu_device):
    """The HIP path against oracle/torch_splat.py -- a forward written from the textbook formulas (Sigma = R S^2 R^T, EWA projection,
    front-to-back compositing) and differentiated by autograd in float64, i.e. code that shares NOTHING with the reference-derived C
    oracle or the emulated-reference fixtures -- at 6000 Gaussians / 256 x 256 / SH 3 and (round 5) at BASELINE configuration 1 exactly
    (10 k Gaussians, 256 x 256, SH degree 0: scenes.config_scene(1)), both colour modes.  The bar is the plain 1e-4 on every
    tensor, no relaxation: exact arithmetic is the yardstick here, and the fp32 paths sit 1e-6 .. 2e-5 from it on these scenes."""
    from oracle import torch_splat
    from youreditableavatar_amd import scenes
    if seed == "cfg1":
        cloud, cams, dL = scenes.config_scene(1)
        cam, name, min_r = cams[0], "cfg1_10k_256_sh0", 10_000
    else:
        cloud = scenes.make_cloud(6000, 3, seed=seed, scale_mult=3.0)
        cam = scenes.orbit_camera(256, 256, azimuth_deg=33.0)
        dL = scenes.upstream_gradient(256, 256, seed=5)
        name, min_r = "6000_256", 40_000
    ref = torch_splat.run_scene(cloud, cam, dL, mode=mode)
    inp = util.scene_input(cloud, cam, mode)
    full = util.hip_run(inp, None, pruning=False)           # the reference's instance lists: n_contrib is comparable as a list position
    assert int(full["num_rendered"]) == int(ref["num_rendered"]) > min_r
    assert np.array_equal(full["radii"], ref["radii"])
    assert (full["n_contrib"].astype(np.int64) == ref["n_contrib"]).mean() >= 0.999
    assert util.rel_l2(full["final_T"], ref["final_T"]) <= 1e-5
    mine = util.hip_run(inp, dL)
    rep = {"color": util.rel_l2(mine["color"], ref["color"])}
    assert np.array_equal(mine["radii"], ref["radii"])
    pairs = [("dL_dmeans3D", "grad_means3D"), ("dL_dopacity", "grad_opacities"), ("dL_dscales", "grad_scales"), ("dL_drotations", "grad_rotations")]
    pairs += [("dL_dsh", "grad_shs")] if mode == "sh" else [("dL_dcolors", "grad_colors_precomp")]
    for a, b in pairs:
        rep[a] = util.rel_l2(mine[a], ref[b])
    rep["dL_dmeans2D"] = util.rel_l2(mine["dL_dmeans2D"][:, :2], ref["grad_means2D"][:, :2])
    util.record_parity(f"fp64_autograd_{mode}_{name}", rep)
    print(mode, {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in rep.items()})
    for k, v in rep.items():
        assert v <= util.REL_TOL, (k, v)


def test_fuzz_vs_oracle_has_no_miss(gpu_device):
    """128 random scenes against the CPU oracle at the bar of util.compare -- every tensor within max(1e-4, 2 x the distance of the
    reference's own fp32 arithmetic from exact arithmetic on that scene, measured with the oracle's C text compiled in double) -- and NO
    failure budget: zero misses.  (Round 3 asserted "<= 5 of 128" at a blanket tolerance; the three-way adjudication of round 4,
    profiles/r04_adjudication.txt, showed that in most scenes over 1e-4 the fp32 ORACLE is the side further from exact arithmetic, and
    the per-Gaussian chain of the HIP path was moved to double where it was not.)"""
    from tests import fuzz
    res = fuzz.run(seed=2026, n_scenes=128, log=lambda *a: None)
    util.record_parity("fuzz_128_scenes", res)
    print({k: v for k, v in res.items() if k != "largest_ok"})
    assert res["misses"] == 0, res
    # the criterion is frozen (round 5): the routes beyond "within the bar of the fp32 oracle" stay the exception -- at most 2 % of the scenes
    assert res["scenes_beyond_the_oracle_route"] <= 0.02 * res["scenes"], res


@pytest.mark.parametrize("seed,scene,tensor,flipped,bound", [(23, 93, "dL_dconic", "f32_in", 2e-5), (37, 89, "dL_dconic", "f32_in", 5e-5)])
def test_cutoff_flip_scenes_are_the_reference_with_one_decision_taken_the_other_way(seed, scene, tensor, flipped, bound, gpu_device):
    """The two fuzz scenes of round 4 in which ONE (pixel, entry) pair at the edge of a large splat lies within fp32's evaluation noise of the
    cut-off alpha >= 1/255 and the kernels decide it the other way than the fp32 oracle ("Cut-off flips"): the HIP
    result is > 1e-4 from the fp32 oracle and from the double build, passes the bar (the oracle's cut-off builds are part of the reference
    noise), and is the SAME function as the oracle build that decides such pairs as blended: 6.7e-4 from the fp32 oracle and < 2e-5 from that build
    in the first scene; in the second (a needle: that build flips a few more pairs than the kernels do) 1.0e-4 and 4e-5."""
    from tests import fuzz, adjudicate
    rng = np.random.default_rng(seed)
    for it in range(scene + 1):
        desc, inp, dL = fuzz.random_scene(rng, it)
    ref = util.oracle_run(inp, dL)
    mine = util.hip_run(inp, dL)
    direct = util.rel_l2(np.asarray(mine[tensor]).reshape(np.asarray(ref[tensor]).shape), ref[tensor])
    assert direct > util.REL_TOL, (desc, direct)           # (otherwise the scene no longer shows what it is kept for)
    util.compare(mine, ref)
    other = adjudicate.oracle_variant(inp, dL, flipped)
    same = util.rel_l2(np.asarray(mine[tensor]).reshape(np.asarray(other[tensor]).shape), other[tensor])
    util.record_parity(f"cutoff_flip_seed{seed}_scene{scene}", {"vs_fp32_oracle": direct, f"vs_{flipped}": same})
    print(desc, tensor, f"vs fp32 oracle {direct:.2e}, vs {flipped} {same:.2e}")
    assert same <= bound and same < 0.5 * direct, (desc, same, direct)


@pytest.mark.parametrize("seed,scene,tensor,before,bound", [(94, 71, "dL_dcov3D", 1.875e-4, 1.17e-4), (104, 25, "dL_dconic", 2.575e-4, 1e-5)])
def test_forward_and_backward_take_the_same_cutoff_decisions(seed, scene, tensor, before, bound, gpu_device):
    """The two scenes of 7 584 fuzzed ones (rounds 5-6) that missed the frozen criterion -- each ONE splat's gradient:
    * seed 94 / scene 71 (428 Gaussians, 270 x 219, splats x 8; round 5): dL_dcov3D 1.875e-4 from the fp32 oracle, bar 2 eta = 1.17e-4;
    * seed 104 / scene 25 (2 752 Gaussians, 29 x 138, splats x 8; found by round 6's soak over twelve new seeds, with round 5's library too): dL_dconic of an
      image-filling splat 2.575e-4 from every build of the oracle, bar 1.95e-4.
    Bisected over the image (the backward is linear in dL/d image: tests/tools/diag_pixels.py), the second one was ONE pixel at which the BACKWARD blended an
    entry the FORWARD had skipped: both kernels evaluate alpha from the same expression, but the compiler had contracted it differently in each (the forward fused
    t * dx with the rounded c dy^2, the backward rounded t * dx and fused c dy * dy), and on that pair the two roundings fell on different sides of alpha = 1/255 --
    every entry in front of it at that pixel saw T off by 1/255.  pair_power2 (tgs_device.hpp) is now ONE instruction sequence for both kernels, and BOTH scenes
    pass (the first had been blamed on the pairing of the four-pixel sums in round 5: it was the same thing, the pass trim had changed the backward's contraction).
    Regression test: the criterion holds on both, and the tensor that missed is well inside what it was."""
    from tests import fuzz
    rng = np.random.default_rng(seed)
    for it in range(scene + 1):
        desc, inp, dL = fuzz.random_scene(rng, it)
    ref = util.oracle_run(inp, dL)
    mine = util.hip_run(inp, dL)
    rep = util.compare(mine, ref)
    d = util.rel_l2(np.asarray(mine[tensor]).reshape(np.asarray(ref[tensor]).shape), ref[tensor])
    util.record_parity(f"former_miss_seed{seed}_scene{scene}", {f"{tensor}_vs_fp32_oracle": d, "before_the_fix": before})
    print(desc, f"{tensor} vs fp32 oracle {d:.3e} (before the fix {before:.3e})", {k: v for k, v in rep.items() if "|route" in k})
    assert d <= bound, (desc, tensor, d)
    # the forward's decisions are the backward's: final_T replayed by the backward's own alphas would otherwise differ by a factor (1 - alpha) somewhere
    light = util.hip_run(inp, dL, light_tiles=True)
    for k in util.GRAD_KEYS:
        if k in mine:
            assert util.rel_l2(light[k], mine[k]) <= 1e-5, k        # (the light groups share pair_power2: same decisions, another summation order)


KNOWN_CAP_MISSES = [      # (fuzz seed, scene index, the tensor that misses, its recorded distance from the fp32 oracle)
    pytest.param(207, 90, "dL_dconic", 1.025e-3, id="seed207_scene90"),
    pytest.param(308, 9, "dL_dconic", 2.787e-3, id="seed308_scene9"),
    pytest.param(419, 91, "dL_dmeans2D", 1.812e-3, id="seed419_scene91"),
    pytest.param(420, 89, "dL_dcov3D", 1.028e-3, id="seed420_scene89"),
]


@pytest.mark.xfail(strict=True, raises=AssertionError, reason="the four recorded misses of round 6's 9 216 scenes on the fixed kernels: one-splat differences on scenes whose reference noise exceeds the 1e-3 cap "
                                       "(profiles/r06_fuzz_soak_e.txt, _f.txt: soaks E, F and G)")
@pytest.mark.parametrize("seed,scene,tensor,recorded", KNOWN_CAP_MISSES)
def test_known_cap_misses_are_still_the_recorded_ones(gpu_device, seed, scene, tensor, recorded):
    """Fuzz seed 207 / scene 90 (4 924 Gaussians, 283 x 219, SH 2, splats x 8): at ONE pixel the forward (and, consistently, the backward: default and
    fixed-order backward agree to 4e-7) decides a pair of a needle splat on the other side of alpha = 1/255 than the fp32 oracle -- final_T differs by 2.7e-3
    there.  The reference's own builds are ~1e-3 apart on this scene (dL_dconic: HIP is 4.3e-4 from the cut-off-in build, 4.6e-4 from the FMA build, 1.025e-3 from
    the plain fp32 and the double builds), so the bar is its cap, 1e-3, and the HIP path is 2.5 % over it.
    Fuzz seed 308 / scene 9 (1 500 Gaussians, 272 x 173, SH 3, splats x 3): the same at one pixel (final_T differs by 3.9e-3), all of the error in one Gaussian;
    dL_dconic is 2.787e-3 from the plain fp32, the exp2 and the double builds and 3.9e-6 / 4.4e-6 from the cut-off-out and the FMA-contracted builds -- the
    reference's own builds are 2.8e-3 apart, the bar is the cap.
    Fuzz seed 419 / scene 91 (4 577 Gaussians, 152 x 203, SH 0, splats x 8): one splat, dL_dmeans2D 1.812e-3 from the plain fp32, FMA, exp2 and double builds and
    1.1e-4 from the cut-off-IN build -- the same kind of flip in the other direction.
    Fuzz seed 420 / scene 89 (67 Gaussians, 87 x 154, SH 0, splats x 8): no flip -- dL_dconic is 2.5e-6 and dL_dmeans2D 4.9e-5 from the fp32 oracle; one needle splat's
    dL_dcov3D (an intermediate here: the caller gets dL_dscales / dL_drotations) is 1.028e-3 from the fp32 evaluation of the ill-conditioned chain and 1.59e-3 from the
    exact one: 2.8 % over the cap.
    Kept as STRICT expected failures: a change that clears one turns the test red (update the record); one that doubles the distance raises."""
    from tests import fuzz
    rng = np.random.default_rng(seed)
    for it in range(scene + 1):
        desc, inp, dL = fuzz.random_scene(rng, it)
    ref = util.oracle_run(inp, dL)
    mine = util.hip_run(inp, dL)
    d = util.rel_l2(np.asarray(mine[tensor]).reshape(np.asarray(ref[tensor]).shape), ref[tensor])
    util.record_parity(f"known_miss_seed{seed}_scene{scene}", {f"{tensor}_vs_fp32_oracle": d, "recorded": recorded})
    print(desc, f"{tensor} vs fp32 oracle {d:.3e} (recorded {recorded:.3e})")
    if d > 2.05 * recorded:
        raise RuntimeError(f"seed {seed} scene {scene} got worse: {tensor} {d:.3e} from the fp32 oracle (recorded {recorded:.3e})")
    util.compare(mine, ref)                               # the expected failure: AssertionError from the frozen criterion


def test_nonfinite_upstream_gradient_stays_with_the_splats_that_cover_its_pixel(gpu_device):
    """ADVICE of round 5: the per-pixel backward loads the inputs of a tile's padding lanes (W or H not a multiple of 16) from a clamped address
    -- pixel (0, 0) -- and used to zero them with a multiplication: 0 * NaN = NaN, so a non-finite dL_dpixel at (0, 0) reached every splat on the
    right / bottom edge tiles.  With a select, and in the reference, only splats that blend into pixel (0, 0) see it."""
    from youreditableavatar_amd import scenes
    W, H = 150, 90                                        # neither a multiple of 16: both edges have padding lanes
    cloud = scenes.make_cloud(4000, 1, seed=5, scale_mult=2.0)
    cam = scenes.orbit_camera(W, H, azimuth_deg=20.0)
    inp = util.scene_input(cloud, cam, "sh")
    dL = scenes.upstream_gradient(W, H, seed=3)
    clean = util.hip_run(inp, dL)
    dLn = dL.copy(); dLn[:, 0, 0] = np.nan
    for lt in (False, True):
        bad = util.hip_run(inp, dLn, light_tiles=lt)
        ref = util.oracle_run(inp, dLn)
        hit = ~np.isfinite(np.asarray(ref["dL_dopacity"]).reshape(-1))
        mine_hit = ~np.isfinite(np.asarray(bad["dL_dopacity"]).reshape(-1))
        # splats whose instance in tile 0 lies in front of pixel (0, 0)'s last contributor may differ by the pruned / quadrant-culled instances: the
        # HIP path must not poison MORE than the reference does, and everything else must be the clean run's value
        assert not np.any(mine_hit & ~hit), f"{int((mine_hit & ~hit).sum())} splats poisoned beyond the reference's {int(hit.sum())} (light_tiles={lt})"
        ok = ~hit
        for k in ("dL_dopacity", "dL_dmeans2D", "dL_dconic"):
            a = np.asarray(bad[k]).reshape(len(hit), -1)[ok]; b = np.asarray(clean[k]).reshape(len(hit), -1)[ok]
            assert np.all(np.isfinite(a)) and util.rel_l2(a, b) <= 1e-5, (k, lt)


@pytest.mark.parametrize("P,W,H,deg,mode,scale_mult", [(20_000, 320, 200, 3, "sh", 1.0), (60_000, 500, 333, 1, "precomp", 2.0), (3_000, 100, 60, 2, "sh", 6.0)])
def test_light_tile_groups_vs_oracle(P, W, H, deg, mode, scale_mult, gpu_device):
    """The light groups of the render kernels (tiles with fewer than 128 instances composited four / three per workgroup: fwd_light_group,
    bwd_light_group) through the single-view entry points, where they are off by default: the full parity bar against the oracle, the same
    image bit for bit as with one workgroup per tile, and gradients equal up to the in-tile summation order.  Scenes: mostly light tiles, a
    mix of light and heavy ones, ragged image sizes; the last one has no light tile at all."""
    from youreditableavatar_amd import scenes
    cloud = scenes.make_cloud(P, deg, seed=21 + P, scale_mult=scale_mult)
    cam = scenes.orbit_camera(W, H, azimuth_deg=75.0)
    inp = util.scene_input(cloud, cam, mode)
    dL = scenes.upstream_gradient(W, H, seed=8)
    ref = util.oracle_run(inp, dL)
    mine = util.hip_run(inp, dL, light_tiles=True)
    rep = util.compare(mine, ref)
    n = (mine["ranges"][:, 1].astype(np.int64) - mine["ranges"][:, 0].astype(np.int64))
    print(P, W, H, "light tiles", int(((n > 0) & (n < 128)).sum()), "others", int((n >= 128).sum()), {k: f"{v:.1e}" for k, v in rep.items() if k.startswith("dL_") or k == "color"})
    one = util.hip_run(inp, dL, light_tiles=False)
    assert np.array_equal(mine["color"], one["color"]) and np.array_equal(mine["n_contrib"], one["n_contrib"]) and np.array_equal(mine["final_T"], one["final_T"])
    for k in util.GRAD_KEYS:
        if k in mine:
            assert util.rel_l2(mine[k], one[k]) <= 1e-5, k


@pytest.mark.parametrize("fwd_light,bwd_light", [(True, False), (False, True)])
def test_backward_with_other_light_option_than_forward(fwd_light, bwd_light, gpu_device):
    """The backward may be called with other options than the frame's forward (include/tgs_raster.h, light_tiles): a tile's deepest blended
    position travels in BOTH copies of its descriptor (tile_desc / light_desc, written by k_render_fwd), light groups are only formed when
    the forward wrote light_desc, and a frame can be back-propagated twice.  Gradients: the oracle's, to the parity bar."""
    from youreditableavatar_amd import scenes
    cloud = scenes.make_cloud(20_000, 2, seed=77)
    cam = scenes.orbit_camera(320, 200, azimuth_deg=40.0)
    inp = util.scene_input(cloud, cam, "sh")
    dL = scenes.upstream_gradient(320, 200, seed=9)
    ref = util.oracle_run(inp, dL)
    mine = util.hip_run(inp, dL, light_tiles=fwd_light, light_tiles_bwd=bwd_light, backward_twice=True)
    rep = util.compare(mine, ref)
    print({k: f"{v:.1e}" for k, v in rep.items() if k.startswith("dL_")})
    same = util.hip_run(inp, dL, light_tiles=fwd_light)
    for k in util.GRAD_KEYS:
        if k in mine:
            assert util.rel_l2(mine[k], same[k]) <= 1e-5, k
