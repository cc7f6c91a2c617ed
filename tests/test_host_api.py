"""CPU: the host-side mirror of the reference interface (names, argument meaning, error behaviour)
and the scene/camera helpers against fixtures recorded from the reference's own Python helpers."""
import os

import numpy as np
import pytest
import torch

from tests.util import GOLDEN_DIR, ROOT


def test_settings_tuple_matches_reference_field_order():
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    # diff_gaussian_rasterization/__init__.py:157-169
    assert GaussianRasterizationSettings._fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier",
                                                     "viewmatrix", "projmatrix", "sh_degree", "campos", "prefiltered", "debug")


def _rast():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    rs = GaussianRasterizationSettings(16, 16, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0, torch.zeros(3), False, False)
    return GaussianRasterizer(rs)


def test_validation_messages_are_the_reference_ones():
    r = _rast()
    m, z, o = torch.zeros(4, 3), torch.zeros(4, 3), torch.zeros(4, 1)
    with pytest.raises(Exception, match="Please provide excatly one of either SHs or precomputed colors!"):
        r(m, z, o)
    with pytest.raises(Exception, match="Please provide excatly one of either SHs or precomputed colors!"):
        r(m, z, o, shs=torch.zeros(4, 1, 3), colors_precomp=torch.zeros(4, 3))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(m, z, o, colors_precomp=torch.zeros(4, 3))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(m, z, o, colors_precomp=torch.zeros(4, 3), scales=torch.ones(4, 3), rotations=torch.ones(4, 4), cov3D_precomp=torch.zeros(4, 6))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(m, z, o, colors_precomp=torch.zeros(4, 3), scales=torch.ones(4, 3))


def test_no_cpu_fallback_fails_loudly():
    r = _rast()
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(torch.zeros(4, 3), torch.zeros(4, 3), torch.zeros(4, 1), colors_precomp=torch.zeros(4, 3), scales=torch.ones(4, 3), rotations=torch.ones(4, 4))
    with pytest.raises(RuntimeError, match="no CPU path"):
        r.markVisible(torch.zeros(4, 3))


def test_means3d_shape_error_is_the_reference_one():
    from diff_gaussian_rasterization import _C
    e = torch.Tensor([])
    with pytest.raises(RuntimeError, match=r"means3D must have dimensions \(num_points, 3\)"):
        _C.rasterize_gaussians(torch.zeros(3), torch.zeros(4, 2), e, torch.zeros(4, 1), e, e, 1.0, e, torch.eye(4), torch.eye(4), 1.0, 1.0,
                               16, 16, e, 0, torch.zeros(3), False, False)


def test_import_surface():
    import diff_gaussian_rasterization as dgr
    assert hasattr(dgr, "GaussianRasterizationSettings") and hasattr(dgr, "GaussianRasterizer") and hasattr(dgr, "rasterize_gaussians")
    for f in ("rasterize_gaussians", "rasterize_gaussians_backward", "mark_visible"):      # ext.cpp:15-19
        assert hasattr(dgr._C, f)
    assert os.path.exists(dgr._C.loaded_library())


# ---- scene helpers vs the reference's own helpers (fixture made by tests/make_ref_utils_fixture.py) ----
@pytest.fixture(scope="module")
def ref_fix():
    return np.load(os.path.join(GOLDEN_DIR, "ref_utils_fixture.npz"))


def test_projection_matrix_matches_reference(ref_fix):
    from youreditableavatar_amd import scenes
    for args, out in zip(ref_fix["proj_args"], ref_fix["proj_out"]):
        assert np.allclose(scenes.projection_matrix(*args), out, rtol=1e-6, atol=0)


def test_world_to_view_matches_reference(ref_fix):
    from youreditableavatar_amd import scenes
    for R, t, out in zip(ref_fix["w2v_R"], ref_fix["w2v_t"], ref_fix["w2v_out"]):
        assert np.allclose(scenes.world_to_view(R.astype(np.float32), t.astype(np.float32)), out, rtol=1e-6, atol=1e-7)


def test_caller_side_sh_to_rgb_matches_reference_eval_sh(ref_fix):
    from youreditableavatar_amd import scenes
    sh, d = ref_fix["sh_coeffs"], ref_fix["sh_dirs"]
    for deg in range(4):
        ours = scenes.sh_to_rgb_numpy(sh, d, np.zeros(3, np.float32), deg)      # campos 0 => dirs = means / |means| = d
        ref = np.maximum(ref_fix[f"sh_out_deg{deg}"] + 0.5, 0.0)               # tetgs_model.py:436-441
        assert np.allclose(ours, ref, rtol=2e-5, atol=2e-6), deg


def test_orbit_camera_geometry():
    from youreditableavatar_amd import scenes
    cam = scenes.orbit_camera(1920, 1080, azimuth_deg=33.0, elevation_deg=5.0, radius=3.0)
    origin_h = np.array([0, 0, 0, 1], np.float32)
    pv = origin_h @ cam.viewmatrix
    assert abs(pv[0]) < 1e-5 and abs(pv[1]) < 1e-5 and abs(pv[2] - 3.0) < 1e-5       # look-at origin, depth = radius
    ph = origin_h @ cam.projmatrix
    assert abs(ph[0] / ph[3]) < 1e-5 and abs(ph[1] / ph[3]) < 1e-5                    # projects to the image centre
    assert abs(cam.tanfovx / cam.tanfovy - 1920 / 1080) < 1e-6                       # square pixels
    assert np.allclose(np.linalg.norm(cam.campos), 3.0, atol=1e-5)
    # config table is BASELINE.json's
    assert scenes.CONFIGS[3]["P"] == 500_000 and (scenes.CONFIGS[3]["width"], scenes.CONFIGS[3]["height"]) == (1920, 1080)
    assert scenes.CONFIGS[5]["P"] == 2_000_000 and scenes.CONFIGS[4]["views"] == 64


def test_cloud_is_deterministic():
    from youreditableavatar_amd import scenes
    a, b = scenes.make_cloud(1000, 3, seed=9), scenes.make_cloud(1000, 3, seed=9)
    for k in ("means3D", "scales", "rotations", "opacities", "shs"):
        assert np.array_equal(a[k], b[k])
    assert a["shs"].shape == (1000, 16, 3) and a["opacities"].shape == (1000, 1)
    assert np.allclose(np.linalg.norm(a["rotations"], axis=1), 1.0, atol=1e-5)


def test_sh_color_oracle_matches_reference_get_points_rgb(ref_fix):
    """oracle/sh_color_ref.py (the checker of the fused SH->RGB op) against outputs AND gradients of the reference's
    own eval_sh composed as in tetgs_model.py:413-442."""
    from oracle import sh_color_ref
    for levels in (1, 2, 3, 4):
        sh = torch.tensor(ref_fix["rgb_sh"], requires_grad=True)
        pos = torch.tensor(ref_fix["rgb_pos"], requires_grad=True)
        col = sh_color_ref.points_rgb(sh, levels, positions=pos, camera_centers=torch.tensor(ref_fix["rgb_cam"]))
        col.backward(torch.tensor(ref_fix["rgb_gcol"]))
        assert np.allclose(col.detach().numpy(), ref_fix[f"rgb_colors_l{levels}"], rtol=1e-6, atol=1e-7)
        assert np.allclose(sh.grad.numpy(), ref_fix[f"rgb_dsh_l{levels}"], rtol=1e-6, atol=1e-7)
        if levels > 1:
            assert np.allclose(pos.grad.numpy(), ref_fix[f"rgb_dpos_l{levels}"], rtol=1e-5, atol=1e-6)


def test_sh_color_requires_gpu_and_a_mode():
    from youreditableavatar_amd import sh_color
    with pytest.raises(ValueError, match="Either camera_centers or directions must be provided."):
        sh_color.points_rgb(torch.zeros(4, 16, 3), 4, positions=torch.zeros(4, 3))
    with pytest.raises(RuntimeError, match="no CPU path"):
        sh_color.points_rgb(torch.zeros(4, 16, 3), 4, directions=torch.zeros(4, 3))


def test_sync_free_batch_capacity_policy():
    """Host logic of multiview.SyncFreeBatch: no bound -> synchronous; headroom and granule rounding; slow decay; cooldown."""
    from youreditableavatar_amd.multiview import SyncFreeBatch
    b = SyncFreeBatch(headroom=1.25, granule=1 << 16)
    assert b.capacity() is None and b.streams == 4 and b.deferred
    b.bound = 960_000
    cap = b.capacity()
    assert cap % (1 << 16) == 0 and 960_000 * 1.25 < cap <= 960_000 * 1.25 + (1 << 16)
    b.bound = 1 << 40
    assert b.capacity() == 0x7fffffff                       # the ABI's limit on tile instances
    b.bound = 1000
    b._cooldown = 2                                         # after a list outgrew the LDS sort: synchronous for a while
    assert b.capacity() is None
    b._cooldown = 0
    assert b.capacity() == 1 << 16


def test_tile_bound_policy(monkeypatch):
    """Host logic of the bounds on the tiles with instances that size the sync-free grids: none until a step has reported counts, 10 %
    headroom rounded up to 64 tiles, TGS_TILE_BOUND=0 switches them off; the drop-in API's speculation keeps a decaying maximum of the
    tile counts beside the instance counts and counts an exceeded tile guess as a miss."""
    from youreditableavatar_amd.multiview import SyncFreeBatch
    b = SyncFreeBatch()
    assert b.tile_bound is None and b.tile_capacity() == 0 and b.class_bound == [0, 0]
    b.tile_bound = 2721
    assert b.tile_capacity() == 3008 and b.tile_capacity() % 64 == 0 and b.tile_capacity() >= 2721 * 1.1
    monkeypatch.setenv("TGS_TILE_BOUND", "0")
    assert b.tile_capacity() == 0
    monkeypatch.delenv("TGS_TILE_BOUND")
    from youreditableavatar_amd.diff_gaussian_rasterization import _Speculation
    sp = _Speculation()
    key = (500_000, 1080, 1920, "cuda:0")
    assert sp.guess(key) is None and sp.tile_guess(key) == 0
    sp.update(key, 750_000, None, tiles=2700, tile_guess=0)
    g, tg = sp.guess(key), sp.tile_guess(key)
    assert g >= 750_000 * 1.15 and tg % 64 == 0 and 2700 * 1.15 <= tg <= 2700 * 1.15 + 64
    sp.update(key, 700_000, g, tiles=2500, tile_guess=tg)            # inside both guesses: no miss, the maxima decay slowly
    assert sp.state[key][1] == 0 and sp.state[key][0] >= 700_000 and sp.state[key][3] >= 2500
    for _ in range(sp.MAX_MISSES):                                   # the instance guess holds, the tile guess does not: misses all the same
        sp.update(key, 700_000, sp.guess(key) or 10 ** 9, tiles=10 * tg, tile_guess=tg)
    assert sp.guess(key) is None                                      # cooling down
    # light-tile groups for a frame that has the GPU to itself: only when the last frame had thousands of light tiles (2048 x 2048), not at 1080p
    sp2 = _Speculation()
    assert sp2.light_tiles(key) is False
    sp2.update(key, 750_000, None, tiles=2700, tile_guess=0, mid_tiles=890)
    assert sp2.light_tiles(key) is False
    sp2.update(key, 1_500_000, None, tiles=9000, tile_guess=0, mid_tiles=2500)
    assert sp2.light_tiles(key) is True
    # one outlier view must not hold the binning buffer at its size for thousands of calls (ADVICE of round 5): after FAR_CALLS calls below
    # half of the bound it falls to twice the largest of them; a camera set whose counts stay within a factor of two of the bound keeps it
    sp3 = _Speculation()
    sp3.update(key, 3_000_000, None)
    for _ in range(sp3.FAR_CALLS - 1):
        sp3.update(key, 700_000, sp3.guess(key))
    assert sp3.state[key][0] > 2_500_000
    sp3.update(key, 700_000, sp3.guess(key))
    assert 1_390_000 <= sp3.state[key][0] <= 1_400_000 and sp3.guess(key) >= 1_390_000 * 1.15
    for _ in range(3 * sp3.FAR_CALLS):
        sp3.update(key, 800_000, sp3.guess(key))
    assert sp3.state[key][0] > 1_100_000                 # 800 k is not below half of the bound: only the slow decay applies


def test_deferred_backward_rejects_mixed_batches():
    import torch
    from youreditableavatar_amd.multiview import DeferredBackward
    from youreditableavatar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings
    z = torch.zeros
    rs = lambda deg: GaussianRasterizationSettings(image_height=8, image_width=8, tanfovx=1.0, tanfovy=1.0, bg=z(3), scale_modifier=1.0, viewmatrix=z(4, 4),
                                                    projmatrix=z(4, 4), sh_degree=deg, campos=z(3), prefiltered=False, debug=False)
    leaves = dict(means3D=z(4, 3), sh=z(4, 16, 3), colors_precomp=z(0), opacities=z(4, 1), scales=z(4, 3), rotations=z(4, 4), cov3D_precomp=z(0))
    d = DeferredBackward()
    d.add(rs(3), leaves, z(4), z(1), z(1), z(1), 0, None)
    with pytest.raises(RuntimeError):
        d.add(rs(2), leaves, z(4), z(1), z(1), z(1), 0, None)                      # another SH degree
    with pytest.raises(RuntimeError):
        d.add(rs(3), dict(leaves, means3D=z(4, 3)), z(4), z(1), z(1), z(1), 0, None)   # another parameter tensor


def test_bench_contract_is_parseable_without_a_gpu():
    """bench.py: the driver's flags exist, the defaults are N=1 and a run of minutes, and the JSON line carries the contract's keys
    (checked on the source: running it needs the MI355X)."""
    import ast
    import subprocess
    import sys
    src = open(os.path.join(ROOT, "bench.py")).read()
    ast.parse(src)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--views-per-gpu", "--streams", "--sync-per-frame", "--no-cpu"):
        assert flag in out.stdout, flag
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"steps"', '"warmup"', '"ms_per_step"', '"higher_is_better"', '"scaling"', '"vs_baseline"', '"dtype"',
                '"data"', '"config"', '"workload"', '"roofline"', '"bound"', '"achieved"', '"peak"', '"frac"', '"traffic"', '"cpu_baseline"', '"cores"', '"kind"',
                '"sample"'):
        assert key in src, key
    assert "from oracle" in src and "cpu_baseline" in src          # the oracle is used for the reported CPU baseline only
