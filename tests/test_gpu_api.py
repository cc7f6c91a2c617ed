"""GPU: the public drop-in API (GaussianRasterizer + autograd) the way tetgs_scene calls it, edge cases,
and the BASELINE.json configurations at full size against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _settings(cam, deg, dev, debug=False, campos_2d=False, cpu_settings=False):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    d = torch.device("cpu") if cpu_settings else dev
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(d)
    campos = t(cam.campos)
    if campos_2d:
        campos = campos.reshape(1, 3)          # p3d get_camera_center() returns [1,3] (tetgs_model.py:502)
    return GaussianRasterizationSettings(image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx,
                                         tanfovy=cam.tanfovy, bg=t(cam.bg), scale_modifier=1.0, viewmatrix=t(cam.viewmatrix),
                                         projmatrix=t(cam.projmatrix), sh_degree=deg, campos=campos, prefiltered=False, debug=debug)


def _leaves(cloud, dev, colors=None):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev).requires_grad_(True)
    L = dict(means3D=t(cloud["means3D"]), opacities=t(cloud["opacities"]), scales=t(cloud["scales"]), rotations=t(cloud["rotations"]))
    if colors is None:
        L["shs"] = t(cloud["shs"])
    else:
        L["colors_precomp"] = t(colors)
    return L


def test_training_style_call_matches_oracle(gpu_device):
    """colors_precomp + scales/rotations, means2D gradient carrier, campos [1,3] -- tetgs_model.py:524-614."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from youreditableavatar_amd import scenes
    cloud = scenes.make_cloud(8000, 2, seed=31, scale_mult=3.0)
    cam = scenes.orbit_camera(240, 160, azimuth_deg=70.0)
    colors = scenes.sh_to_rgb_numpy(cloud["shs"], cloud["means3D"], cam.campos, 2)
    L = _leaves(cloud, gpu_device, colors)
    means2D = torch.zeros(8000, 3, device=gpu_device, requires_grad=True)
    rast = GaussianRasterizer(_settings(cam, 2, gpu_device, campos_2d=True))
    img, radii = rast(means3D=L["means3D"], means2D=means2D, shs=None, colors_precomp=L["colors_precomp"], opacities=L["opacities"],
                      scales=L["scales"], rotations=L["rotations"], cov3D_precomp=None)
    assert img.shape == (3, 160, 240) and radii.dtype == torch.int32 and not radii.requires_grad
    dL = scenes.upstream_gradient(240, 160, seed=8)
    img.backward(torch.from_numpy(dL).to(gpu_device))
    inp = util.scene_input(cloud, cam, mode="precomp")
    ref = util.oracle_run(inp, dL)
    assert util.rel_l2(img.detach().cpu().numpy(), ref["color"]) <= 1e-4
    assert np.array_equal(radii.cpu().numpy(), ref["radii"])
    for name, key in (("means3D", "dL_dmeans3D"), ("opacities", "dL_dopacity"), ("scales", "dL_dscales"), ("rotations", "dL_drotations"),
                      ("colors_precomp", "dL_dcolors")):
        assert util.rel_l2(L[name].grad.cpu().numpy(), ref[key]) <= 1e-4, name
    assert util.rel_l2(means2D.grad.cpu().numpy(), ref["dL_dmeans2D"]) <= 1e-4
    assert torch.all(means2D.grad[:, 2] == 0)


def test_validation_render_no_grad_and_settings_on_cpu(gpu_device):
    """shs in the rasterizer under no_grad (refine.py:413-420); settings tensors left on the CPU are moved."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from youreditableavatar_amd import scenes
    cloud = scenes.make_cloud(3000, 3, seed=32, scale_mult=3.0)
    cam = scenes.orbit_camera(128, 128)
    L = _leaves(cloud, gpu_device)
    with torch.no_grad():
        img, radii = GaussianRasterizer(_settings(cam, 3, gpu_device, cpu_settings=True))(
            means3D=L["means3D"], means2D=torch.zeros(3000, 3, device=gpu_device), opacities=L["opacities"], shs=L["shs"],
            scales=L["scales"], rotations=L["rotations"])
    ref = util.oracle_run(util.scene_input(cloud, cam))
    assert util.rel_l2(img.cpu().numpy(), ref["color"]) <= 1e-4
    assert not img.requires_grad


def test_two_graphs_alive_and_noncontiguous_inputs(gpu_device):
    """State buffers belong to their autograd graph: a second forward must not disturb the first backward
    (paint_2dgs.py:434-446 keeps `initial_outputs`); strided inputs are made contiguous like the reference."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from youreditableavatar_amd import scenes
    cloud = scenes.make_cloud(4000, 1, seed=33, scale_mult=3.0)
    cams = [scenes.orbit_camera(160, 96, azimuth_deg=a) for a in (0.0, 120.0)]
    L = _leaves(cloud, gpu_device)
    wide = torch.zeros(4000, 6, device=gpu_device)
    wide[:, ::2] = L["scales"].detach()
    scales_strided = wide[:, ::2].requires_grad_(True)           # non-contiguous view
    dL = torch.from_numpy(scenes.upstream_gradient(160, 96)).to(gpu_device)
    imgs = []
    for cam in cams:
        img, _ = GaussianRasterizer(_settings(cam, 1, gpu_device))(means3D=L["means3D"], means2D=torch.zeros(4000, 3, device=gpu_device, requires_grad=True),
                                                                    opacities=L["opacities"], shs=L["shs"], scales=scales_strided, rotations=L["rotations"])
        imgs.append(img)
    g1 = torch.autograd.grad(imgs[0], L["means3D"], dL, retain_graph=False)[0]       # backward of the FIRST graph after the second forward
    ref = util.oracle_run(util.scene_input(cloud, cams[0]), dL.cpu().numpy())
    assert util.rel_l2(g1.cpu().numpy(), ref["dL_dmeans3D"]) <= 1e-4


def test_empty_and_all_culled(gpu_device):
    from diff_gaussian_rasterization import GaussianRasterizer
    from youreditableavatar_amd import scenes
    cam = scenes.orbit_camera(64, 48, bg=(0.3, 0.6, 0.9))
    z = lambda *s: torch.zeros(*s, device=gpu_device, requires_grad=True)
    # P == 0: the reference returns an all-zero image (rasterize_points.cu:81), not the background
    img, radii = GaussianRasterizer(_settings(cam, 0, gpu_device))(means3D=z(0, 3), means2D=z(0, 3), opacities=z(0, 1), colors_precomp=z(0, 3),
                                                                     scales=z(0, 3), rotations=z(0, 4))
    assert img.shape == (3, 48, 64) and torch.all(img == 0) and radii.numel() == 0
    # everything behind the camera: background image, zero gradients
    cloud = scenes.make_cloud(500, 0, seed=34)
    cloud["means3D"] = (cloud["means3D"] * 0.1 + cam.campos * 2.0).astype(np.float32)
    L = _leaves(cloud, gpu_device)
    img, radii = GaussianRasterizer(_settings(cam, 0, gpu_device))(means3D=L["means3D"], means2D=z(500, 3), opacities=L["opacities"], shs=L["shs"],
                                                                     scales=L["scales"], rotations=L["rotations"])
    img.sum().backward()
    assert torch.all(radii == 0)
    assert torch.allclose(img, torch.tensor(cam.bg, device=gpu_device).reshape(3, 1, 1).expand_as(img))
    for k in ("means3D", "opacities", "scales", "rotations", "shs"):
        assert torch.all(L[k].grad == 0), k


def test_mark_visible(gpu_device):
    from diff_gaussian_rasterization import GaussianRasterizer
    from oracle import oracle
    from youreditableavatar_amd import scenes
    cam = scenes.orbit_camera(64, 64)
    cloud = scenes.make_cloud(5000, 0, seed=35)
    m = cloud["means3D"].copy()
    m[::3] = m[::3] * 0.1 + cam.campos * 2.0
    vis = GaussianRasterizer(_settings(cam, 0, gpu_device)).markVisible(torch.from_numpy(m).to(gpu_device))
    assert vis.dtype == torch.bool
    assert np.array_equal(vis.cpu().numpy(), oracle.mark_visible(m, cam.viewmatrix, cam.projmatrix))


def test_debug_mode_and_prefiltered_error(gpu_device, tmp_path, monkeypatch):
    """debug=True synchronises after every stage; prefiltered=True with a culled Gaussian is an error
    (auxiliary.h:156-160: printf + __trap in the reference, RuntimeError + snapshot here)."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from youreditableavatar_amd import scenes
    monkeypatch.chdir(tmp_path)
    cam = scenes.orbit_camera(64, 64)
    cloud = scenes.make_cloud(300, 0, seed=36, scale_mult=5.0)
    L = _leaves(cloud, gpu_device)
    rs = _settings(cam, 0, gpu_device, debug=True)
    img, _ = GaussianRasterizer(rs)(means3D=L["means3D"], means2D=torch.zeros(300, 3, device=gpu_device), opacities=L["opacities"], shs=L["shs"],
                                    scales=L["scales"], rotations=L["rotations"])
    img.sum().backward()
    ref = util.oracle_run(util.scene_input(cloud, cam))
    assert util.rel_l2(img.detach().cpu().numpy(), ref["color"]) <= 1e-4
    cloud["means3D"][0] = cam.campos * 2.0                       # behind the camera
    L = _leaves(cloud, gpu_device)
    rs = rs._replace(prefiltered=True)
    with pytest.raises(RuntimeError, match="filtered although prefiltered"):
        GaussianRasterizer(rs)(means3D=L["means3D"], means2D=torch.zeros(300, 3, device=gpu_device), opacities=L["opacities"], shs=L["shs"],
                               scales=L["scales"], rotations=L["rotations"])
    assert (tmp_path / "snapshot_fw.dump").exists()


@pytest.mark.parametrize("where", [0, 70_000, 199_999])
def test_prefiltered_violation_travels_through_the_block_flags(where, gpu_device):
    """Round 4: the prefiltered-violation flag is no atomic on Meta any more (the per-Gaussian stage clears Meta itself) but one word per
    256-Gaussian block, OR-ed by k_scan -- across several scan iterations at 200 k Gaussians (782 blocks).  One culled Gaussian anywhere
    must raise, in the synchronous forward, in the speculative one (second call on the same sizes) and in the whole-batch path; the same
    cloud without `prefiltered` renders, and twice in a row (Meta really is cleared per frame: no stale flag)."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from youreditableavatar_amd import scenes
    from youreditableavatar_amd.multiview import FlatGradients, SyncFreeBatch
    P = 200_000
    cam = scenes.orbit_camera(320, 200, azimuth_deg=12.0)
    cloud = scenes.make_cloud(P, 0, seed=5, scale_mult=1.0)
    behind = (np.concatenate([cloud["means3D"], np.ones((P, 1), np.float32)], 1) @ cam.viewmatrix)[:, 2] <= 0.2
    cloud["means3D"][behind] *= 0.2                                  # every Gaussian in front of the near plane ...
    assert not ((np.concatenate([cloud["means3D"], np.ones((P, 1), np.float32)], 1) @ cam.viewmatrix)[:, 2] <= 0.2).any()
    L = _leaves(cloud, gpu_device)
    rs_ok = _settings(cam, 0, gpu_device)._replace(prefiltered=True)
    kw = lambda LL: dict(means3D=LL["means3D"], means2D=torch.zeros(P, 3, device=gpu_device), opacities=LL["opacities"], shs=LL["shs"], scales=LL["scales"], rotations=LL["rotations"])
    for _ in range(3):                                              # synchronous, then speculative: no violation, no stale flag
        img, _r = GaussianRasterizer(rs_ok)(**kw(L))
    bad = dict(cloud); bad["means3D"] = cloud["means3D"].copy(); bad["means3D"][where] = cam.campos * 2.0      # ... except this one
    Lb = _leaves(bad, gpu_device)
    for _ in range(2):                                              # (the first call of a size is synchronous, the second speculative)
        with pytest.raises(RuntimeError, match="filtered although prefiltered"):
            GaussianRasterizer(rs_ok)(**kw(Lb))
    img2, _r = GaussianRasterizer(rs_ok)(**kw(L))                    # and the flag does not stick
    assert torch.equal(img, img2)
    names = ("means3D", "opacities", "scales", "rotations", "shs")
    FlatGradients([Lb[n] for n in names])
    batch = SyncFreeBatch(granule=256, streams=2)
    dL = torch.from_numpy(scenes.upstream_gradient(320, 200, seed=1)).to(gpu_device)
    with pytest.raises(RuntimeError, match="filtered although prefiltered"):
        for _ in range(3):                                          # (the first batch takes the per-view fallback, the later ones the sync-free path)
            batch.run_views([rs_ok, rs_ok], Lb["means3D"], Lb["opacities"], Lb["shs"], Lb["scales"], Lb["rotations"], lambda im: dL, accumulate=False)


@pytest.mark.parametrize("cfg", [2, 3])
def test_baseline_config_full_size_vs_oracle(cfg, gpu_device):
    """BASELINE.json configs 2 (100k / 800x800 / SH3) and 3 (500k / 1920x1080 / SH3) at FULL size against the CPU oracle."""
    from youreditableavatar_amd import scenes
    cloud, cams, dL = scenes.config_scene(cfg)
    inp = util.scene_input(cloud, cams[0])
    mine = util.hip_run(inp, dL)
    ref = util.oracle_run(inp, dL)
    rep = util.compare(mine, ref)
    util.record_parity(f"cfg{cfg}", rep, extra=dict(num_rendered=int(mine["num_rendered"]), num_rendered_reference=int(ref["num_rendered"])))
    print(cfg, {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in rep.items()})


@pytest.mark.parametrize("scale_mult", [4.0, 6.0])
def test_large_splats_full_size_vs_oracle(scale_mult, gpu_device):
    """Config 2's cloud (100k Gaussians, 800x800, SH3) with the scales multiplied: most splats now cover 5..64 tiles (the wave-cooperative
    k_scatter path, instance pruning through the 64-bit live-tile mask, slab rows by popcount) and some more than 64 (workgroup
    path) -- the full parity bar against the CPU oracle, lists included."""
    from youreditableavatar_amd import scenes
    cfg = scenes.CONFIGS[2]
    cloud = scenes.make_cloud(cfg["P"], cfg["sh_degree"], cfg["seed"], scale_mult=scale_mult)
    cam = scenes.orbit_camera(cfg["width"], cfg["height"], azimuth_deg=25.0)
    inp = util.scene_input(cloud, cam)
    dL = scenes.upstream_gradient(cfg["width"], cfg["height"], seed=9)
    mine = util.hip_run(inp, dL)
    ref = util.oracle_run(inp, dL)
    tt = ref["tiles_touched"]
    assert ((tt > 4) & (tt <= 64)).mean() > 0.2 and (scale_mult < 6 or (tt > 64).sum() > 100)      # the paths are really taken
    rep = util.compare(mine, ref)
    print(scale_mult, mine["num_rendered"], ref["num_rendered"], {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in rep.items() if k in ("color", "instances_dropped", "n_contrib_equal", "dL_dmeans2D")})


def test_config5_overflow_stress_vs_oracle(gpu_device):
    """Config 5 (2M Gaussians, 2048x2048, 1 % flat 1e-8 splats, 1000 oversized splats, 20 000 splats piled into one tile) at FULL size
    against the CPU oracle -- and what the config is for: a tile list beyond the LDS sort (n_overflow > 0, longest list > 8192),
    sorted on the device by k_tile_sort's overflow workers.  Plus size-independent properties: sorted lists, range bookkeeping,
    n_contrib bounds, finite outputs, linearity of the backward in dL."""
    from diff_gaussian_rasterization import _C
    from youreditableavatar_amd import scenes
    cloud, cams, dL = scenes.config_scene(5)
    inp = util.scene_input(cloud, cams[0])
    a = util.hip_run(inp, dL)
    P, H, W = 2_000_000, 2048, 2048
    rg = a["ranges"].astype(np.int64)
    lens = rg[:, 1] - rg[:, 0]
    assert lens.max() > 8192 and int((lens > 8192).sum()) >= 1                              # the overflow sort really ran
    assert lens.sum() == a["num_rendered"] == int(a["tiles_touched"].astype(np.int64).sum())
    assert np.all(a["n_contrib"].reshape(H // 16, 16, W // 16, 16).max(axis=(1, 3)).reshape(-1) <= lens)
    # every list is sorted by (depth, index)
    keys = (a["depths"].view(np.uint32).astype(np.uint64)[a["point_list"]] << np.uint64(32)) | a["point_list"].astype(np.uint64)
    starts = rg[lens > 0, 0]
    brk = np.zeros(len(keys), bool); brk[starts] = True
    assert np.all((keys[1:] > keys[:-1]) | brk[1:])
    assert np.isfinite(a["color"]).all() and all(np.isfinite(a[k]).all() for k in util.GRAD_KEYS)
    assert (a["radii"] > 0).sum() > 1_900_000 and a["tiles_touched"].max() >= 64          # oversized splats are there
    b = util.hip_run(inp, 2.0 * dL, introspect=False)                                      # backward is linear in the upstream gradient
    for k in ("dL_dmeans3D", "dL_dopacity", "dL_dsh"):
        assert util.rel_l2(b[k], 2.0 * a[k]) <= 1e-5, k
    del b
    ref = util.oracle_run(inp, dL)                                                         # the full parity bar, lists included
    rlens = ref["ranges"][:, 1].astype(np.int64) - ref["ranges"][:, 0]
    assert rlens.max() >= 20_000
    rep = util.compare(a, ref)
    util.record_parity("cfg5", rep, extra=dict(longest_list=int(lens.max()), lists_beyond_lds_sort=int((lens > 8192).sum()), num_rendered=int(a["num_rendered"]),
                                               num_rendered_reference=int(ref["num_rendered"])))
    print(5, {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in rep.items()})


def test_fused_accumulate_equals_autograd_sum(gpu_device):
    """multiview.rasterize_accumulate: gradients of several views added in place == autograd's sum of per-view gradients."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from youreditableavatar_amd import scenes
    from youreditableavatar_amd.multiview import FlatGradients, rasterize_accumulate
    cloud = scenes.make_cloud(6000, 3, seed=41, scale_mult=3.0)
    cams = [scenes.orbit_camera(176, 112, azimuth_deg=a) for a in (0.0, 90.0, 200.0)]
    dL = torch.from_numpy(scenes.upstream_gradient(176, 112)).to(gpu_device)
    names = ("means3D", "opacities", "scales", "rotations", "shs")

    def run(fused):
        L = _leaves(cloud, gpu_device)
        flat = FlatGradients([L[n] for n in names])
        for cam in cams:
            rs = _settings(cam, 3, gpu_device)
            kw = dict(means3D=L["means3D"], means2D=torch.zeros(6000, 3, device=gpu_device, requires_grad=True), opacities=L["opacities"],
                      shs=L["shs"], scales=L["scales"], rotations=L["rotations"])
            img, _ = rasterize_accumulate(rs, **kw) if fused else GaussianRasterizer(rs)(**kw)
            img.backward(dL)
        return flat.flat.clone()

    from diff_gaussian_rasterization import _C
    _C.set_deterministic(True)          # fixed summation order, so the two paths can be compared tightly
    try:
        a, b = run(True), run(False)
    finally:
        _C.set_deterministic(False)
    assert a.abs().max() > 0
    assert util.rel_l2(a.cpu().numpy(), b.cpu().numpy()) <= 1e-6


def test_sync_free_forward_matches_and_rejects(gpu_device):
    """tgs_forward_async: with enough binning capacity the frame is the synchronous one bit for bit; with too little it
    is rejected on the device -- background image, zero dL_dmeans2D, nothing accumulated -- and the status says so."""
    from diff_gaussian_rasterization import _C
    from youreditableavatar_amd import scenes
    cloud = scenes.make_cloud(6000, 3, seed=43, scale_mult=3.0)
    cam = scenes.orbit_camera(176, 112, azimuth_deg=30.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(gpu_device)
    e = torch.Tensor([])
    args = (t(cam.bg), t(cloud["means3D"]), e, t(cloud["opacities"]), t(cloud["scales"]), t(cloud["rotations"]), 1.0, e, t(cam.viewmatrix),
            t(cam.projmatrix), cam.tanfovx, cam.tanfovy, 112, 176, t(cloud["shs"]), 3, t(cam.campos), False, False)
    R, color, radii, geom, binning, img = _C.rasterize_gaussians(*args)
    assert R > 1000 and _C.frame_status(img) == (R, 0)
    dL = t(scenes.upstream_gradient(176, 112))

    def backward(R_, radii_, geom_, binning_, img_, into):
        return _C.rasterize_gaussians_backward_accumulate(args[0], args[1], radii_, e, args[4], args[5], 1.0, e, args[8], args[9], cam.tanfovx,
                                                          cam.tanfovy, dL, args[14], 3, args[16], geom_, R_, binning_, img_, False, into)

    def zeros():
        return dict(means3D=torch.zeros(6000, 3, device=gpu_device), opacities=torch.zeros(6000, 1, device=gpu_device),
                    sh=torch.zeros(6000, 16, 3, device=gpu_device), scales=torch.zeros(6000, 3, device=gpu_device),
                    rotations=torch.zeros(6000, 4, device=gpu_device))

    _C.set_deterministic(True)
    try:
        ref = zeros()
        g2d_ref = backward(R, radii, geom, binning, img, ref)
        # roomy capacity: identical frame, true count in the Meta record
        cap = R + 5000
        R2, color2, radii2, geom2, binning2, img2 = _C.rasterize_gaussians(*args, r_capacity=cap)
        assert R2 == cap
        assert torch.equal(color2, color) and torch.equal(radii2, radii)
        assert _C.decode_meta(_C.frame_meta(img2)) == (R, 0) and _C.frame_status(img2) == (R, 0)
        got = zeros()
        g2d = backward(R2, radii2, geom2, binning2, img2, got)
        assert torch.equal(g2d, g2d_ref)
        for k in ref:
            assert torch.equal(got[k], ref[k]), k
        # exact fit is accepted, one less is not
        assert _C.frame_status(_C.rasterize_gaussians(*args, r_capacity=R)[5]) == (R, 0)
        R3, color3, radii3, geom3, binning3, img3 = _C.rasterize_gaussians(*args, r_capacity=R - 1)
        assert _C.frame_status(img3) == (R, _C.FRAME_REJECTED)
        bg = t(cam.bg).view(3, 1, 1).expand(3, 112, 176)
        assert torch.equal(color3, bg) and torch.equal(radii3, radii)
        acc = zeros()
        for v in acc.values():
            v.fill_(7.0)
        g2d3 = backward(R3, radii3, geom3, binning3, img3, acc)
        assert torch.all(g2d3 == 0) and all(torch.all(v == 7.0) for v in acc.values())
        # the plain (non-accumulating) backward of a rejected frame writes every output: zeros, never uninitialised memory
        torch.empty(1 << 22, device=gpu_device).fill_(float("nan"))     # poison what the allocator hands out next
        outs = _C.rasterize_gaussians_backward(args[0], args[1], radii3, e, args[4], args[5], 1.0, e, args[8], args[9], cam.tanfovx, cam.tanfovy, dL,
                                               args[14], 3, args[16], geom3, R3, binning3, img3, False)
        assert len(outs) == 8 and all(torch.all(o == 0) for o in outs)
    finally:
        _C.set_deterministic(False)


@pytest.mark.parametrize("deferred", [False, True])
@pytest.mark.parametrize("streams", [1, 2])
def test_sync_free_batch_rerenders_rejected_views(streams, deferred, gpu_device):
    """multiview.SyncFreeBatch: one read-back per batch; views that outgrow the bound are rendered again, and the
    accumulated gradients equal the per-frame-synchronised ones."""
    from diff_gaussian_rasterization import _C
    from youreditableavatar_amd import scenes
    from youreditableavatar_amd.multiview import FlatGradients, SyncFreeBatch, rasterize_accumulate
    cloud = scenes.make_cloud(6000, 3, seed=41, scale_mult=3.0)
    cams = [scenes.orbit_camera(176, 112, azimuth_deg=a) for a in (0.0, 90.0, 200.0, 310.0)]
    dL = torch.from_numpy(scenes.upstream_gradient(176, 112)).to(gpu_device)
    names = ("means3D", "opacities", "scales", "rotations", "shs")
    L = _leaves(cloud, gpu_device)
    flat = FlatGradients([L[n] for n in names])
    settings = [_settings(c, 3, gpu_device) for c in cams]
    caps = []

    def rasterize(v, cap):
        caps.append(cap)
        return rasterize_accumulate(settings[v], means3D=L["means3D"], means2D=torch.zeros(6000, 3, device=gpu_device, requires_grad=True),
                                    opacities=L["opacities"], shs=L["shs"], scales=L["scales"], rotations=L["rotations"], r_capacity=cap,
                                    return_meta=True)

    _C.set_deterministic(True)
    try:
        flat.zero_()
        for v in range(4):
            rasterize(v, None)[0].backward(dL)
        want, imgs_want = flat.flat.clone(), [rasterize(v, None)[0].detach().clone() for v in range(4)]
        batch = SyncFreeBatch(headroom=1.25, granule=256, streams=streams, deferred=deferred)
        flat.zero_(); caps.clear()
        imgs = batch.run(range(4), rasterize, lambda v, img: dL)                  # first batch: no bound yet -> synchronous frames
        assert caps == [None] * 4 and batch.bound is not None and torch.equal(flat.flat, want)
        flat.zero_(); caps.clear()
        imgs = batch.run(range(4), rasterize, lambda v, img: dL)                  # now sync-free
        # the batched per-Gaussian pass sums the views in registers before it touches the gradient buffers
        assert caps == [batch.capacity()] * 4 and batch.rejected == 0
        assert torch.equal(flat.flat, want) if not deferred else util.rel_l2(flat.flat.cpu().numpy(), want.cpu().numpy()) <= 5e-6
        assert all(torch.equal(a, b) for a, b in zip(imgs, imgs_want))
        batch.bound = batch.bound // 3                                            # a bound some views no longer fit
        small = batch.capacity()
        flat.zero_(); caps.clear()
        imgs = batch.run(range(4), rasterize, lambda v, img: dL)
        assert batch.rejected >= 1 and caps[:4] == [small] * 4 and caps[4:] == [None] * batch.rejected
        assert all(torch.equal(a, b) for a, b in zip(imgs, imgs_want))
        # the re-rendered views are added after the others: same terms, different order
        assert util.rel_l2(flat.flat.cpu().numpy(), want.cpu().numpy()) <= 5e-6
        assert batch.capacity() > small
    finally:
        _C.set_deterministic(False)


@pytest.mark.parametrize("levels", [1, 2, 3, 4])
@pytest.mark.parametrize("M", [16, 9])
def test_fused_sh_color_matches_reference(levels, M, gpu_device):
    """sh_color.points_rgb (tgs_sh_rgb_forward/backward) against the recorded outputs/gradients of the reference's
    get_points_rgb composition (M = 16) and against the torch restatement (other strides, direction mode)."""
    import os
    from oracle import sh_color_ref
    from youreditableavatar_amd import sh_color
    if M < levels * levels:
        pytest.skip("not enough coefficients")
    fx = np.load(os.path.join(util.GOLDEN_DIR, "ref_utils_fixture.npz"))
    sh_np = fx["rgb_sh"][:, :M].copy()
    t = lambda a, g=True: torch.tensor(a, device=gpu_device, requires_grad=g)
    sh, pos = t(sh_np), t(fx["rgb_pos"])
    col = sh_color.points_rgb(sh, levels, positions=pos, camera_centers=t(fx["rgb_cam"], False))
    col.backward(t(fx["rgb_gcol"], False))
    assert util.rel_l2(col.detach().cpu().numpy(), fx[f"rgb_colors_l{levels}"]) <= 1e-6
    assert util.rel_l2(sh.grad.cpu().numpy(), fx[f"rgb_dsh_l{levels}"][:, :M]) <= 1e-6
    if levels > 1:
        assert util.rel_l2(pos.grad.cpu().numpy(), fx[f"rgb_dpos_l{levels}"]) <= 1e-5
    else:
        assert torch.all(pos.grad == 0)
    # direction mode against the CPU restatement
    d_np = fx["sh_dirs"][:, :].repeat(4, 0)[:sh_np.shape[0]]
    shd, dirs = t(sh_np), t(d_np)
    cd = sh_color.points_rgb(shd, levels, directions=dirs)
    cd.backward(t(fx["rgb_gcol"], False))
    sh_c, dirs_c = torch.tensor(sh_np, requires_grad=True), torch.tensor(d_np, requires_grad=True)
    cr = sh_color_ref.points_rgb(sh_c, levels, directions=dirs_c)
    cr.backward(torch.tensor(fx["rgb_gcol"]))
    assert util.rel_l2(cd.detach().cpu().numpy(), cr.detach().numpy()) <= 1e-6
    assert util.rel_l2(shd.grad.cpu().numpy(), sh_c.grad.numpy()) <= 1e-6
    if levels > 1:
        assert util.rel_l2(dirs.grad.cpu().numpy(), dirs_c.grad.numpy()) <= 1e-5


@pytest.mark.parametrize("levels,M", [(1, 1), (1, 16), (2, 4), (2, 16), (3, 9), (3, 16), (4, 16)])
def test_fused_sh_color_from_dc_and_rest(levels, M, gpu_device):
    """sh_color.points_rgb_dc_rest (tgs_sh_rgb_dcrest_*): colours straight from the model's two parameters (_sh_coordinates_dc [P,1,3],
    _sh_coordinates_rest [P,M-1,3]; tetgs_model.py:234-239) -- the same numbers and gradients as the reference's get_points_rgb on their
    torch.cat (recorded in ref_utils_fixture.npz), with the gradients arriving on the two parameters; at one level the rest parameter is
    not touched (None gradient where the reference hands out zeros).  M - 1 == levels^2 - 1 takes the block-staged path, M = 16 with
    fewer levels the row-wise one."""
    import os
    from oracle import sh_color_ref
    from youreditableavatar_amd import sh_color
    fx = np.load(os.path.join(util.GOLDEN_DIR, "ref_utils_fixture.npz"))
    sh_np = fx["rgb_sh"][:, :M].copy()
    t = lambda a, g=True: torch.tensor(np.ascontiguousarray(a), device=gpu_device, requires_grad=g)
    dc, pos = t(sh_np[:, :1]), t(fx["rgb_pos"])
    rest = t(sh_np[:, 1:]) if M > 1 else None
    col = sh_color.points_rgb_dc_rest(dc, rest, levels, positions=pos, camera_centers=t(fx["rgb_cam"], False))
    col.backward(t(fx["rgb_gcol"], False))
    want_dsh = fx[f"rgb_dsh_l{levels}"][:, :M]
    assert util.rel_l2(col.detach().cpu().numpy(), fx[f"rgb_colors_l{levels}"]) <= 1e-6
    assert util.rel_l2(dc.grad.cpu().numpy(), want_dsh[:, :1]) <= 1e-6
    if levels == 1:
        assert rest is None or rest.grad is None                 # exactly zero in the reference: not materialised here
        assert np.all(want_dsh[:, 1:] == 0)
        assert torch.all(pos.grad == 0)
    else:
        assert util.rel_l2(rest.grad.cpu().numpy(), want_dsh[:, 1:]) <= 1e-6
        assert np.all(rest.grad.cpu().numpy()[:, levels * levels - 1:] == 0)
        assert util.rel_l2(pos.grad.cpu().numpy(), fx[f"rgb_dpos_l{levels}"]) <= 1e-5
    # direction mode, and the same result as the [P,M,3] entry point bit for bit
    d_np = fx["sh_dirs"][:, :].repeat(4, 0)[:sh_np.shape[0]]
    dc2, dirs = t(sh_np[:, :1]), t(d_np)
    rest2 = t(sh_np[:, 1:]) if M > 1 else None
    cd = sh_color.points_rgb_dc_rest(dc2, rest2, levels, directions=dirs)
    cd.backward(t(fx["rgb_gcol"], False))
    shd, dirs_b = t(sh_np), t(d_np)
    cb = sh_color.points_rgb(shd, levels, directions=dirs_b)
    cb.backward(t(fx["rgb_gcol"], False))
    assert torch.equal(cd, cb) and torch.equal(dc2.grad, shd.grad[:, :1])
    if levels > 1:
        assert torch.equal(rest2.grad, shd.grad[:, 1:])
        assert util.rel_l2(dirs.grad.cpu().numpy(), dirs_b.grad.cpu().numpy()) <= 1e-6      # (two kernels: the compiler contracts the polynomials differently)
    # a point count that is not a multiple of the block or of four floats (direct path of the last block)
    n = 1001
    dc3 = t(sh_np[:n, :1]); rest3 = t(sh_np[:n, 1:]) if M > 1 else None
    c3 = sh_color.points_rgb_dc_rest(dc3, rest3, levels, directions=t(d_np[:n], False))
    assert torch.equal(c3, cb[:n])
    with pytest.raises(ValueError):
        sh_color.points_rgb_dc_rest(dc, rest, levels)
    if levels > 1:
        with pytest.raises(ValueError):
            sh_color.points_rgb_dc_rest(dc, None, levels, directions=dirs)


def test_fused_sh_color_feeds_the_rasterizer(gpu_device):
    """The training-step composition: fused SH->RGB -> colors_precomp rasterizer -> gradients reach the SH tensor."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from youreditableavatar_amd import scenes, sh_color
    cloud = scenes.make_cloud(5000, 3, seed=51, scale_mult=3.0)
    cam = scenes.orbit_camera(160, 120)
    L = _leaves(cloud, gpu_device)
    colors = sh_color.points_rgb(L["shs"], 4, positions=L["means3D"], camera_centers=torch.tensor(cam.campos, device=gpu_device).reshape(1, 3))
    img, _ = GaussianRasterizer(_settings(cam, 3, gpu_device))(means3D=L["means3D"], means2D=torch.zeros(5000, 3, device=gpu_device, requires_grad=True),
                                                               opacities=L["opacities"], colors_precomp=colors, scales=L["scales"], rotations=L["rotations"])
    dL = scenes.upstream_gradient(160, 120)
    img.backward(torch.from_numpy(dL).to(gpu_device))
    # same scene with SH evaluated inside the rasterizer: identical maths, so image and SH gradients agree
    ref = util.oracle_run(util.scene_input(cloud, cam), dL)
    assert util.rel_l2(img.detach().cpu().numpy(), ref["color"]) <= 1e-4
    assert util.rel_l2(L["shs"].grad.cpu().numpy(), ref["dL_dsh"]) <= 1e-4
    assert util.rel_l2(L["means3D"].grad.cpu().numpy(), ref["dL_dmeans3D"]) <= 2e-4


@pytest.mark.parametrize("P", [5000, 129, 1])
def test_batched_backward_equals_per_view_backward(P, gpu_device):
    """tgs_backward_render + tgs_backward_batch (one per-Gaussian pass for all views) == tgs_backward per view, summed;
    per-view dL_dmeans2D identical; more views than one launch holds (BATCH_VIEWS = 8).  P = 129 / 1: a workgroup of the batch pass with one
    Gaussian in range (every prologue load of the others is clamped to a valid element, their values unused)."""
    from diff_gaussian_rasterization import _C
    from youreditableavatar_amd import scenes
    cloud = scenes.make_cloud(P, 3, seed=47, scale_mult=3.0 if P > 1000 else 12.0)
    cams = [scenes.orbit_camera(176, 112, azimuth_deg=a) for a in np.linspace(0, 330, 11)]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(gpu_device)
    e = torch.Tensor([])
    means, opac, scales, rots, shs = t(cloud["means3D"]), t(cloud["opacities"]), t(cloud["scales"]), t(cloud["rotations"]), t(cloud["shs"])
    dL = t(scenes.upstream_gradient(176, 112))
    _C.set_deterministic(True)
    try:
        want = dict(means3D=torch.zeros(P, 3, device=gpu_device), opacities=torch.zeros(P, 1, device=gpu_device), sh=torch.zeros(P, 16, 3, device=gpu_device),
                    scales=torch.zeros(P, 3, device=gpu_device), rotations=torch.zeros(P, 4, device=gpu_device))
        views, want2d = [], []
        for cam in cams:
            bg, vm, pm, cp = t(cam.bg), t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos)
            R, color, radii, geom, binning, img = _C.rasterize_gaussians(bg, means, e, opac, scales, rots, 1.0, e, vm, pm, cam.tanfovx, cam.tanfovy, 112, 176, shs,
                                                                       3, cp, False, False)
            g = _C.rasterize_gaussians_backward(bg, means, radii, e, scales, rots, 1.0, e, vm, pm, cam.tanfovx, cam.tanfovy, dL, shs, 3, cp, geom, R, binning,
                                                img, False)
            d2d, _dcol, dop, dm3, _dcov, dsh, dsc, drot = g
            want2d.append(d2d)
            for k, v in (("means3D", dm3), ("opacities", dop), ("sh", dsh), ("scales", dsc), ("rotations", drot)):
                want[k] += v
            # fresh state for the split path (the one-view backward above consumed nothing, but keep the two paths apart)
            R, color, radii, geom, binning, img = _C.rasterize_gaussians(bg, means, e, opac, scales, rots, 1.0, e, vm, pm, cam.tanfovx, cam.tanfovy, 112, 176, shs,
                                                                       3, cp, False, False)
            _C.rasterize_gaussians_backward_render(bg, dL, R, binning, img, P)
            views.append(dict(viewmatrix=vm, projmatrix=pm, campos=cp, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, image_height=112, image_width=176, radii=radii,
                              geom=geom, binning=binning, img=img, R=R))
        got = {k: torch.full_like(v, 3.0) for k, v in want.items()}      # accumulate=False must overwrite
        outs = _C.rasterize_gaussians_backward_batch(views, means, shs, 3, scales, rots, 1.0, e, got, accumulate=False)
        for (g2d, _), w in zip(outs, want2d):
            assert torch.equal(g2d, w)
        for k in want:     # same terms, other association: views 0-7 are summed in registers, 8-10 added as a second partial sum
            assert util.rel_l2(got[k].cpu().numpy(), want[k].cpu().numpy()) <= 1e-5, k
        again = {k: v.clone() for k, v in got.items()}
        _C.rasterize_gaussians_backward_batch(views[:3], means, shs, 3, scales, rots, 1.0, e, again, accumulate=True)
        three = {k: torch.zeros_like(v) for k, v in got.items()}
        _C.rasterize_gaussians_backward_batch(views[:3], means, shs, 3, scales, rots, 1.0, e, three, accumulate=False)
        for k in want:
            assert util.rel_l2(again[k].cpu().numpy(), (got[k] + three[k]).cpu().numpy()) <= 1e-6, k
    finally:
        _C.set_deterministic(False)


@pytest.mark.parametrize("streams,group", [(1, 1), (3, 1), (2, 4)])
def test_run_views_whole_batch_path(streams, group, gpu_device):
    """SyncFreeBatch.run_views (tgs_forward_views / tgs_backward_render_views / tgs_backward_batch: three native calls per
    batch) == the per-view path: images, per-view dL/d means2D, accumulated parameter gradients; rejected views are redone."""
    from diff_gaussian_rasterization import _C
    from youreditableavatar_amd import scenes
    from youreditableavatar_amd.multiview import FlatGradients, SyncFreeBatch, rasterize_accumulate
    P = 6000
    cloud = scenes.make_cloud(P, 3, seed=41, scale_mult=3.0)
    cams = [scenes.orbit_camera(176, 112, azimuth_deg=a) for a in (0.0, 70.0, 140.0, 210.0, 300.0)]
    dLs = torch.stack([torch.from_numpy(scenes.upstream_gradient(176, 112, seed=50 + i)) for i in range(5)]).to(gpu_device)
    names = ("means3D", "opacities", "scales", "rotations", "shs")
    L = _leaves(cloud, gpu_device)
    flat = FlatGradients([L[n] for n in names])
    settings = [_settings(c, 3, gpu_device) for c in cams]
    _C.set_deterministic(True)
    _C.set_forward_group(group)             # > 1: k_preprocess_fwd_batch serves several views per launch
    try:
        flat.zero_()
        want_img, want_2d = [], []
        for v in range(5):
            m2 = torch.zeros(P, 3, device=gpu_device, requires_grad=True)
            img, _ = rasterize_accumulate(settings[v], means3D=L["means3D"], means2D=m2, opacities=L["opacities"], shs=L["shs"], scales=L["scales"],
                                          rotations=L["rotations"])
            img.backward(dLs[v])
            want_img.append(img.detach().clone()); want_2d.append(m2.grad.clone())
        want = flat.flat.clone()
        batch = SyncFreeBatch(granule=256, streams=streams)
        calls = []

        def upstream(images):
            calls.append(images.shape)
            return dLs

        run = lambda: batch.run_views(settings, L["means3D"], L["opacities"], L["shs"], L["scales"], L["rotations"], upstream)
        # group == 1: the very same kernels as the per-view path -> bitwise.  group > 1: the per-Gaussian forward comes from the all-views
        # kernel, a second compilation of the same fp32 math (fma contraction differs: conics agree to the last bit or two)
        same = torch.equal if group == 1 else (lambda a, b: util.rel_l2(a.cpu().numpy(), b.cpu().numpy()) <= 2e-6)
        for rep in range(3):                                # first batch: synchronous (learns the bound); then the whole-batch path, twice (pool reuse)
            flat.zero_()
            imgs = run()
            assert tuple(imgs.shape) == (5, 3, 112, 176)
            assert all(same(imgs[v], want_img[v]) for v in range(5)), rep
            assert all(same(batch.viewspace_grads[v], want_2d[v]) for v in range(5)), rep
            assert util.rel_l2(flat.flat.cpu().numpy(), want.cpu().numpy()) <= 2e-5, rep     # two compilations of the same fp32 math (fma contraction): scales / rotations differ by ~5e-6
        assert batch.rejected == 0 and len(calls) == 3
        batch.bound = batch.bound // 3                      # some views no longer fit: rejected on the device, rendered again
        flat.zero_()
        imgs = run()
        assert batch.rejected >= 1
        assert all(same(imgs[v], want_img[v]) for v in range(5))
        assert all(same(batch.viewspace_grads[v], want_2d[v]) for v in range(5))
        assert util.rel_l2(flat.flat.cpu().numpy(), want.cpu().numpy()) <= 2e-5
        # a bound on the tiles with instances that every view exceeds (the sync-free grids are sized by it): rejected by k_scan, rendered again
        assert batch.tile_bound is not None and 8 < batch.tile_bound <= 77 and batch.tile_capacity() >= batch.tile_bound
        rejected_before, learned = batch.rejected, batch.tile_bound
        batch.tile_capacity = lambda: 8
        flat.zero_()
        imgs = run()
        del batch.tile_capacity
        assert batch.rejected == rejected_before + 5 and batch.tile_bound == learned
        assert all(same(imgs[v], want_img[v]) for v in range(5))
        assert all(same(batch.viewspace_grads[v], want_2d[v]) for v in range(5))
        assert util.rel_l2(flat.flat.cpu().numpy(), want.cpu().numpy()) <= 2e-5
        # accumulate=False stores: whatever the buffers held is gone
        flat.flat.fill_(5.0)
        batch.run_views(settings, L["means3D"], L["opacities"], L["shs"], L["scales"], L["rotations"], upstream, accumulate=False)
        assert util.rel_l2(flat.flat.cpu().numpy(), want.cpu().numpy()) <= 2e-5
        # one gradient image for all views
        flat.zero_()
        run2 = batch.run_views(settings, L["means3D"], L["opacities"], L["shs"], L["scales"], L["rotations"], lambda images: dLs[0])
        assert torch.isfinite(flat.flat).all() and flat.flat.abs().max() > 0
    finally:
        _C.set_deterministic(False)
        _C.set_forward_group(2)                 # the default


@pytest.mark.parametrize("mode,cov_mode", [("sh", "scale_rot"), ("precomp", "scale_rot"), ("sh", "cov3d")])
def test_backward_outputs_the_caller_discards_are_not_written(mode, cov_mode, gpu_device):
    """Round 6: dL_dcolors on the SH path and dL_dcov3D on the scale / rotation path go to inputs that are None (__init__.py:137-152) -- the
    drop-in backward passes need_colors / need_cov3D = False and the per-Gaussian kernel does not write them (NULL through tgs_backward_opt).
    Every other output is bit for bit what the call that writes everything returns; a required one (colours with colors_precomp, cov3D with
    cov3D_precomp) is written whatever the flags say."""
    from diff_gaussian_rasterization import _C, GaussianRasterizer
    from youreditableavatar_amd import scenes
    cloud = scenes.make_cloud(5000, 2, seed=9, scale_mult=2.0)
    if cov_mode == "cov3d":
        from oracle.emu_crosscheck_cov import cov3d_from
        cloud["cov3D_precomp"] = cov3d_from(cloud["scales"], cloud["rotations"])
    cam = scenes.orbit_camera(200, 120, azimuth_deg=30.0)
    inp = util.scene_input(cloud, cam, mode, cov_mode)
    dL = torch.from_numpy(scenes.upstream_gradient(200, 120, seed=4)).to(gpu_device)
    t = lambda k: (torch.from_numpy(np.ascontiguousarray(inp[k], np.float32)).to(gpu_device) if inp.get(k) is not None else torch.Tensor([]))
    bg, means3D, opac, view, proj, campos = t("bg"), t("means3D"), t("opacities"), t("viewmatrix"), t("projmatrix"), t("campos")
    sh, colors, scales, rots, cov = t("shs"), t("colors_precomp"), t("scales"), t("rotations"), t("cov3D_precomp")
    R, color, radii, geom, binning, img = _C.rasterize_gaussians(bg, means3D, colors, opac, scales, rots, 1.0, cov, view, proj, float(inp["tanfovx"]), float(inp["tanfovy"]), 120, 200,
                                                                 sh, 2, campos, False, False)
    bw = lambda **kw: _C.rasterize_gaussians_backward(bg, means3D, radii, colors, scales, rots, 1.0, cov, view, proj, float(inp["tanfovx"]), float(inp["tanfovy"]), dL, sh, 2, campos,
                                                      geom, R, binning, img, False, deterministic=True, **kw)
    full = bw()
    lean = bw(need_colors=False, need_cov3D=False)
    names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")
    for n, a, b in zip(names, full, lean):
        dropped = (n == "dL_dcolors" and mode == "sh") or (n == "dL_dcov3D" and cov_mode == "scale_rot")
        if dropped:
            assert b.numel() == 0 and a.shape[0] == 5000, n
        else:
            assert torch.equal(a, b), n
    # ... and through the drop-in API: the same parameter gradients as before
    Lp = {k: t(k).requires_grad_(True) for k in ("means3D", "opacities") + (("shs",) if mode == "sh" else ("colors_precomp",)) +
          (("scales", "rotations") if cov_mode == "scale_rot" else ("cov3D_precomp",))}
    img2, _ = GaussianRasterizer(_settings(cam, 2, gpu_device))(means3D=Lp["means3D"], means2D=torch.zeros(5000, 3, device=gpu_device, requires_grad=True), opacities=Lp["opacities"],
                                                               shs=Lp.get("shs"), colors_precomp=Lp.get("colors_precomp"), scales=Lp.get("scales"), rotations=Lp.get("rotations"),
                                                               cov3D_precomp=Lp.get("cov3D_precomp"))
    img2.backward(dL)
    want = dict(means3D=full[3], opacities=full[2], shs=full[5], colors_precomp=full[1], scales=full[6], rotations=full[7], cov3D_precomp=full[4])
    for k, p in Lp.items():
        assert p.grad is not None and util.rel_l2(p.grad.cpu().numpy(), want[k].cpu().numpy().reshape(p.grad.shape)) <= 2e-6, k


@pytest.mark.parametrize("P", [6000, 6037])
def test_run_views_with_level_major_sh_gradients(P, gpu_device):
    """FlatGradients(level_major=True): the batch's per-Gaussian pass writes dL_dsh coefficient plane by coefficient plane
    (tgs_backward_batch_range_planes) -- the SAME numbers as the row-major pass, bit for bit, at another address; stored and accumulated,
    range by range, with a ragged last workgroup (P not a multiple of 128), and through the rare paths that go through the one-view kernel
    (first batch, a view rendered again)."""
    from diff_gaussian_rasterization import _C
    from youreditableavatar_amd import scenes
    from youreditableavatar_amd.multiview import FlatGradients, SyncFreeBatch
    cloud = scenes.make_cloud(P, 3, seed=43, scale_mult=3.0)
    cams = [scenes.orbit_camera(176, 112, azimuth_deg=a) for a in (0.0, 100.0, 250.0)]
    dLs = torch.stack([torch.from_numpy(scenes.upstream_gradient(176, 112, seed=60 + i)) for i in range(3)]).to(gpu_device)
    names = ("means3D", "opacities", "scales", "rotations", "shs")
    settings = [_settings(c, 3, gpu_device) for c in cams]
    _C.set_deterministic(True)
    try:
        res = {}
        for lm in (False, True):
            L = _leaves(cloud, gpu_device)
            flat = FlatGradients([L[n] for n in names], sh_params={4: 0}, level_major=lm)
            assert L["shs"].grad.is_contiguous() != lm
            batch = SyncFreeBatch(granule=256, streams=2)
            run = lambda **kw: batch.run_views(settings, L["means3D"], L["opacities"], L["shs"], L["scales"], L["rotations"], lambda images: dLs, **kw)
            out = []
            flat.zero_(); run(); out.append([L[n].grad.clone().contiguous() for n in names])                      # first batch: synchronous frames (one-view kernel)
            flat.zero_(); run(); out.append([L[n].grad.clone().contiguous() for n in names])                      # whole-batch path, accumulate into zeros
            flat.flat.fill_(7.0); run(accumulate=False, grad_chunks=3, on_chunk=lambda f, c: None)                # stores, in three ranges
            out.append([L[n].grad.clone().contiguous() for n in names])
            run(grad_chunks=2, on_chunk=lambda f, c: None); out.append([L[n].grad.clone().contiguous() for n in names])   # accumulates on top: twice the step
            batch.bound = batch.bound // 3; flat.zero_(); run()                                                     # some views rendered again
            assert batch.rejected >= 1
            out.append([L[n].grad.clone().contiguous() for n in names])
            res[lm] = out
        for step, (a, b) in enumerate(zip(res[False], res[True])):
            for n, x, y in zip(names, a, b):
                assert torch.equal(x, y), (step, n, util.rel_l2(y.cpu().numpy(), x.cpu().numpy()))
        assert float(res[True][1][4].abs().max()) > 0 and util.rel_l2(res[True][3][4].cpu().numpy(), 2.0 * res[True][2][4].cpu().numpy()) <= 1e-6
    finally:
        _C.set_deterministic(False)


def test_run_views_with_a_list_beyond_the_lds_sort(gpu_device):
    """The sync-free batch path on a scene whose longest tile list (> 16k entries) does not fit the LDS sort: the frame is NOT rejected
    (the overflow is sorted on the device by k_tile_sort's workers, no host-sized launch), nothing is rendered again, no cooldown,
    and images / accumulated gradients equal the synchronous per-view path -- which in turn is checked against the oracle in
    tests/test_gpu_parity.py::test_real_overflow_lists_vs_oracle."""
    from diff_gaussian_rasterization import _C
    from youreditableavatar_amd import scenes
    from youreditableavatar_amd.multiview import FlatGradients, SyncFreeBatch, rasterize_accumulate
    n_blob, P = 20_000, 23_000
    cloud = scenes.concentrate(scenes.make_cloud(P, 1, seed=91, scale_mult=2.0), n_blob, centre=(0.01, -0.01, 0.0), sigma=0.003)
    W, H = 208, 144
    cams = [scenes.orbit_camera(W, H, azimuth_deg=a) for a in (0.0, 95.0, 200.0)]
    dLs = torch.stack([torch.from_numpy(scenes.upstream_gradient(W, H, seed=60 + i)) for i in range(3)]).to(gpu_device)
    names = ("means3D", "opacities", "scales", "rotations", "shs")
    L = _leaves(cloud, gpu_device)
    flat = FlatGradients([L[n] for n in names])
    settings = [_settings(c, 1, gpu_device) for c in cams]
    _C.set_deterministic(True)
    try:
        flat.zero_()
        want_img = []
        for v in range(3):
            m2 = torch.zeros(P, 3, device=gpu_device, requires_grad=True)
            img, _ = rasterize_accumulate(settings[v], means3D=L["means3D"], means2D=m2, opacities=L["opacities"], shs=L["shs"], scales=L["scales"],
                                          rotations=L["rotations"])
            img.backward(dLs[v])
            want_img.append(img.detach().clone())
        want = flat.flat.clone()
        batch = SyncFreeBatch(granule=256, streams=2)
        for rep in range(3):                                # first batch synchronous (learns the bound), then sync-free twice
            imgs = batch.run_views(settings, L["means3D"], L["opacities"], L["shs"], L["scales"], L["rotations"], lambda images: dLs, accumulate=False)
            assert all(torch.equal(imgs[v], want_img[v]) for v in range(3)), rep
            assert util.rel_l2(flat.flat.cpu().numpy(), want.cpu().numpy()) <= 2e-5, rep
        assert batch.rejected == 0 and batch._cooldown == 0 and batch.capacity() is not None
        longest = max(_C.decode_meta_full(batch._pool["host"][v])[2] for v in range(3))
        n_ovf = max(_C.decode_meta_full(batch._pool["host"][v])[3] for v in range(3))
        assert longest > 16384 and n_ovf >= 1              # the sync-free frames really held an overflow list
    finally:
        _C.set_deterministic(False)


@pytest.mark.parametrize("streams,split", [(1, False), (3, False), (4, True)])
def test_run_views_per_view_upstream(streams, split, gpu_device):
    """run_views(upstream_view=...): the loss of each view is taken on the view's own stream and its backward follows without the
    streams meeting in between -- same images, per-view dL/d means2D and parameter gradients as with one upstream_batch call;
    a rejected view is redone with its own upstream call."""
    from diff_gaussian_rasterization import _C
    from youreditableavatar_amd import scenes
    from youreditableavatar_amd.multiview import FlatGradients, SyncFreeBatch
    P = 5000
    cloud = scenes.make_cloud(P, 2, seed=43, scale_mult=2.0)
    cams = [scenes.orbit_camera(160, 96, azimuth_deg=a) for a in (10.0, 80.0, 150.0, 220.0, 290.0)]
    targets = torch.rand(5, 3, 96, 160, generator=torch.Generator().manual_seed(7)).to(gpu_device)
    names = ("means3D", "opacities", "scales", "rotations", "shs")
    L = _leaves(cloud, gpu_device)
    flat = FlatGradients([L[n] for n in names])
    settings = [_settings(c, 2, gpu_device) for c in cams]
    grad_view = lambda v, image: 2.0 * (image - targets[v]) / image.numel()              # d/d image of the mean squared error of one view
    grad_batch = lambda images: 2.0 * (images - targets) / images[0].numel()
    _C.set_deterministic(True)
    try:
        ref = SyncFreeBatch(granule=256, streams=streams)
        for rep in range(2):
            flat.zero_()
            want_img = ref.run_views(settings, L["means3D"], L["opacities"], L["shs"], L["scales"], L["rotations"], grad_batch).clone()
        want, want_2d = flat.flat.clone(), ref.viewspace_grads.clone()
        batch = SyncFreeBatch(granule=256, streams=streams, split=split)     # split: two streams bin, two composite
        seen = []
        def upstream(v, image):
            seen.append(v)
            return grad_view(v, image)
        for rep in range(3):                                # synchronous first batch, then the pipelined path twice
            flat.zero_()
            del seen[:]
            imgs = batch.run_views(settings, L["means3D"], L["opacities"], L["shs"], L["scales"], L["rotations"], None, upstream_view=upstream)
            assert sorted(seen) == [0, 1, 2, 3, 4]
            assert torch.equal(imgs, want_img), rep
            assert torch.equal(batch.viewspace_grads, want_2d), rep
            assert util.rel_l2(flat.flat.cpu().numpy(), want.cpu().numpy()) <= 5e-6, rep
        batch.bound = batch.bound // 3                      # rejected on the device, rendered again
        flat.zero_()
        imgs = batch.run_views(settings, L["means3D"], L["opacities"], L["shs"], L["scales"], L["rotations"], None, upstream_view=upstream)
        assert batch.rejected >= 1
        assert torch.equal(imgs, want_img) and torch.equal(batch.viewspace_grads, want_2d)
        assert util.rel_l2(flat.flat.cpu().numpy(), want.cpu().numpy()) <= 2e-5
        with pytest.raises(RuntimeError):
            batch.run_views(settings, L["means3D"], L["opacities"], L["shs"], L["scales"], L["rotations"], grad_batch, upstream_view=upstream)
    finally:
        _C.set_deterministic(False)


def test_run_views_with_per_view_colors(gpu_device):
    """run_views(colors_precomp=[V,P,3]) -- the reference's training mode, colours evaluated by the caller per view -- against the
    per-view path: images, dL/d colours per view, accumulated parameter gradients; fused SH->RGB in front of it reproduces the
    SH-in-rasterizer gradients end to end."""
    from diff_gaussian_rasterization import _C
    from youreditableavatar_amd import scenes, sh_color
    from youreditableavatar_amd.multiview import FlatGradients, SyncFreeBatch, rasterize_accumulate
    P = 5000
    cloud = scenes.make_cloud(P, 3, seed=43, scale_mult=3.0)
    cams = [scenes.orbit_camera(176, 112, azimuth_deg=a) for a in (10.0, 100.0, 190.0, 280.0)]
    V = len(cams)
    dLs = torch.stack([torch.from_numpy(scenes.upstream_gradient(176, 112, seed=60 + i)) for i in range(V)]).to(gpu_device)
    L = _leaves(cloud, gpu_device)
    names = ("means3D", "opacities", "scales", "rotations", "shs")
    flat = FlatGradients([L[n] for n in names])
    settings = [_settings(c, 3, gpu_device) for c in cams]
    campos = [torch.from_numpy(np.ascontiguousarray(c.campos, np.float32)).to(gpu_device).view(1, 3) for c in cams]
    _C.set_deterministic(True)
    try:
        # (a) SH inside the rasterizer: the reference result for all parameter gradients
        batch_sh = SyncFreeBatch(granule=256)
        for _ in range(2):
            flat.zero_()
            imgs_sh = batch_sh.run_views(settings, L["means3D"], L["opacities"], L["shs"], L["scales"], L["rotations"], lambda im: dLs).clone()
        want = flat.flat.clone()
        # (b) colours by the fused SH->RGB op per view, rasterizer in colors_precomp mode, colour gradients handed back to the op
        batch = SyncFreeBatch(granule=256)
        for rep in range(3):                                # synchronous first, then the whole-batch path (twice: pool reuse)
            flat.zero_()
            cols = [sh_color.points_rgb(L["shs"], 4, positions=L["means3D"], camera_centers=campos[v]) for v in range(V)]
            colors = torch.stack([c.detach() for c in cols])
            imgs = batch.run_views(settings, L["means3D"], L["opacities"], None, L["scales"], L["rotations"], lambda im: dLs, colors_precomp=colors)
            assert tuple(batch.color_grads.shape) == (V, P, 3)
            for v in range(V):
                cols[v].backward(batch.color_grads[v])      # adds dL/d sh and the view-direction part of dL/d means3D
            assert util.rel_l2(imgs.cpu().numpy(), imgs_sh.cpu().numpy()) <= 1e-6, rep
            assert util.rel_l2(flat.flat.cpu().numpy(), want.cpu().numpy()) <= 2e-5, rep
        assert batch.rejected == 0
        with pytest.raises(RuntimeError):
            batch.run_views(settings, L["means3D"], L["opacities"], L["shs"], L["scales"], L["rotations"], lambda im: dLs, colors_precomp=colors)
    finally:
        _C.set_deterministic(False)


def test_run_views_at_the_benchmark_configuration(gpu_device):
    """BASELINE config 3/4 at full size (500k Gaussians, 1920x1080, SH 3, 8 views on 4 streams): the whole-batch path bench.py
    times against the per-view synchronous path (reference protocol) -- images bit-equal, per-view dL/d means2D and the
    accumulated parameter gradients within the stated fp32 tolerance."""
    from diff_gaussian_rasterization import _C
    from youreditableavatar_amd import scenes
    from youreditableavatar_amd.multiview import FlatGradients, SyncFreeBatch, rasterize_accumulate
    cfg = scenes.CONFIGS[3]
    P, W, H, D = cfg["P"], cfg["width"], cfg["height"], cfg["sh_degree"]
    cloud = scenes.make_cloud(P, D, cfg["seed"])
    V = 8
    cams = [scenes.orbit_camera(W, H, azimuth_deg=k * 360.0 / 64) for k in range(V)]
    dL = torch.from_numpy(scenes.upstream_gradient(W, H, seed=cfg["seed"] + 1000)).to(gpu_device)
    L = _leaves(cloud, gpu_device)
    names = ("means3D", "opacities", "scales", "rotations", "shs")
    flat = FlatGradients([L[n] for n in names])
    settings = [_settings(c, D, gpu_device) for c in cams]
    flat.zero_()
    want_img, want_2d = [], []
    for v in range(V):
        m2 = torch.zeros(P, 3, device=gpu_device, requires_grad=True)
        img, _ = rasterize_accumulate(settings[v], means3D=L["means3D"], means2D=m2, opacities=L["opacities"], shs=L["shs"], scales=L["scales"],
                                      rotations=L["rotations"])
        img.backward(dL)
        want_img.append(img.detach().clone()); want_2d.append(m2.grad.clone())
    want = flat.flat.clone()
    batch = SyncFreeBatch()
    for rep in range(3):
        imgs = batch.run_views(settings, L["means3D"], L["opacities"], L["shs"], L["scales"], L["rotations"], lambda im: dL, accumulate=False)
    assert batch.rejected == 0 and batch.capacity() is not None
    for v in range(V):
        assert torch.equal(imgs[v], want_img[v]), v
        assert util.rel_l2(batch.viewspace_grads[v].cpu().numpy(), want_2d[v].cpu().numpy()) <= 1e-5, v   # in-tile summation order (LDS atomics)
    off = 0
    for n in names:
        k = L[n].numel()
        e = util.rel_l2(flat.flat[off:off + k].cpu().numpy(), want[off:off + k].cpu().numpy())
        assert e <= 1e-4, (n, e)
        off += k
    # ... and against the ORACLE, not only against this library's own per-view path: a two-view batch (views 0 and 5 of the orbit) through
    # the same whole-batch entry points, images and per-view dL/d means2D per view, the accumulated parameter gradients against the SUM of
    # the oracle's per-view gradients (summed in float64).  The bar is util.compare's, on the sum: max(1e-4, 2 x the distance of the
    # reference's fp32 arithmetic from the same function in double -- here of the summed fp32 results from the summed double results).
    pick = [0, 5]
    refs = [util.oracle_run(util.scene_input(cloud, cams[v]), dL.cpu().numpy()) for v in pick]
    batch2 = SyncFreeBatch()
    for rep in range(2):
        imgs2 = batch2.run_views([settings[v] for v in pick], L["means3D"], L["opacities"], L["shs"], L["scales"], L["rotations"], lambda im: dL, accumulate=False)
    assert batch2.rejected == 0 and batch2.capacity() is not None
    rep = {}
    for i, r in enumerate(refs):
        rep[f"color_view{pick[i]}"] = util.rel_l2(imgs2[i].cpu().numpy(), r["color"])
        rep[f"dL_dmeans2D_view{pick[i]}"] = util.rel_l2(batch2.viewspace_grads[i].cpu().numpy(), r["dL_dmeans2D"])
    key = dict(means3D="dL_dmeans3D", opacities="dL_dopacity", scales="dL_dscales", rotations="dL_drotations", shs="dL_dsh")
    off = 0
    tol = {}
    for n in names:
        k = L[n].numel()
        ref_sum = sum(np.asarray(r[key[n]], np.float64).reshape(-1) for r in refs)
        rep[key[n] + "_sum"] = util.rel_l2(flat.flat[off:off + k].cpu().numpy(), ref_sum)
        tol[key[n] + "_sum"] = util.REL_TOL
        if rep[key[n] + "_sum"] > util.REL_TOL:
            for r in refs:
                util.reference_noise_of(r)
            f64_sum = sum(np.asarray(r["_f64"][key[n]], np.float64).reshape(-1) for r in refs)
            tol[key[n] + "_sum"] = max(util.REL_TOL, 2.0 * util.rel_l2(ref_sum, f64_sum))
        off += k
    util.record_parity("config4_two_view_batch_vs_oracle_sum", rep)
    print({k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in rep.items()})
    for k, v in rep.items():
        assert v <= tol.get(k, util.REL_TOL), (k, v, tol.get(k))


def test_speculative_forward_is_the_complete_frame(gpu_device):
    """tgs_forward_speculative: with a fitting guess and with a guess that is far too small (retry with exact sizes) the image,
    radii, true count and all gradients equal the plain synchronous call."""
    from diff_gaussian_rasterization import _C
    from youreditableavatar_amd import scenes
    cloud = scenes.make_cloud(6000, 3, seed=45, scale_mult=3.0)
    cam = scenes.orbit_camera(176, 112, azimuth_deg=60.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(gpu_device)
    e = torch.Tensor([])
    args = (t(cam.bg), t(cloud["means3D"]), e, t(cloud["opacities"]), t(cloud["scales"]), t(cloud["rotations"]), 1.0, e, t(cam.viewmatrix),
            t(cam.projmatrix), cam.tanfovx, cam.tanfovy, 112, 176, t(cloud["shs"]), 3, t(cam.campos), False, False)
    dL = t(scenes.upstream_gradient(176, 112))
    # (every knob travels as an explicit option of the call -- tgs_options_t -- nothing process- or thread-wide is set)
    R, color, radii, geom, binning, img = _C.rasterize_gaussians(*args)
    bw = lambda R_, radii_, geom_, binning_, img_, **kw: _C.rasterize_gaussians_backward(args[0], args[1], radii_, e, args[4], args[5], 1.0, e, args[8], args[9],
                                                                                       cam.tanfovx, cam.tanfovy, dL, args[14], 3, args[16], geom_, R_, binning_, img_, False,
                                                                                       deterministic=True, **kw)
    want = bw(R, radii, geom, binning, img)
    tiles = -1
    for guess in (R + 7000, R, max(R // 10, 1), 0):
        carve, color2, radii2, geom2, binning2, img2, true_R, (tiles, mid_tiles) = _C.rasterize_gaussians(*args, r_guess=guess)
        assert true_R == R and carve == (guess if guess >= R else R), guess
        assert torch.equal(color2, color) and torch.equal(radii2, radii), guess
        assert _C.frame_status(img2) == (R, 0)
        got = bw(carve, radii2, geom2, binning2, img2)
        for a, b in zip(got, want):
            assert torch.equal(a, b), guess
    # the synchronous forward reports the same frame facts on request
    out = _C.rasterize_gaussians(*args, info=True)
    assert len(out) == 8 and out[6] == R and out[7] == (tiles, mid_tiles) and 0 <= mid_tiles <= tiles
    # The grids of the stages enqueued ahead of the read-back can be sized by a guessed bound on the tiles with instances
    # (tgs_options_t::tile_bound): a fitting bound and one that is far too small (retry with exact sizes) give the same frame, and the
    # backward may visit the frame's exact number of non-empty tiles only.
    assert 8 < tiles <= 77
    for bound in (tiles + 3, tiles, 4):
        carve, color2, radii2, geom2, binning2, img2, true_R, (tiles2, mid2) = _C.rasterize_gaussians(*args, r_guess=R + 1000, tile_bound=bound)
        assert true_R == R and tiles2 == tiles and mid2 == mid_tiles, bound
        assert torch.equal(color2, color) and torch.equal(radii2, radii), bound
        assert _C.frame_status(img2) == (R, 0)
        got = bw(carve, radii2, geom2, binning2, img2, tile_bound=tiles)
        for a, b in zip(got, want):
            assert torch.equal(a, b), bound
        # the default (LDS-atomic) kernels with exact bounds for both tile classes, and with the light-tile kernels switched off: same gradients
        # up to the in-tile summation order
        for kw in (dict(tile_bound=tiles, mid_bound=max(mid_tiles, 1)), dict(tile_bound=tiles, light_tiles=False), dict()):
            got = _C.rasterize_gaussians_backward(args[0], args[1], radii2, e, args[4], args[5], 1.0, e, args[8], args[9], cam.tanfovx, cam.tanfovy, dL, args[14], 3,
                                                  args[16], geom2, carve, binning2, img2, False, **kw)
            for a, b in zip(got, want):
                assert util.rel_l2(a.cpu().numpy(), b.cpu().numpy()) <= 1e-5, (bound, kw)
        assert _C.frame_status(img2) == (R, 0)
    # a backward with a bound BELOW the frame's tiles with instances would drop gradients silently: the frame's flags say so instead
    bw(carve, radii2, geom2, binning2, img2, tile_bound=tiles - 3)
    with pytest.raises(RuntimeError, match="tile bound"):
        _C.frame_status(img2)


def test_two_threads_with_different_options(gpu_device):
    """Re-entrancy (rasterizer.h:20-85: the reference's statics keep no state): two host threads render different scenes through the one
    library at the same time, one with the reference's instance lists + a small LDS sort budget + the fixed-order backward, the other with
    the defaults.  Each must get exactly what it gets alone -- no option of one call leaks into the other's."""
    import threading
    from diff_gaussian_rasterization import _C
    from youreditableavatar_amd import scenes
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(gpu_device)
    e = torch.Tensor([])

    def make(seed, W, H, P):
        cloud = scenes.make_cloud(P, 2, seed=seed, scale_mult=3.0)
        cam = scenes.orbit_camera(W, H, azimuth_deg=10.0 * seed)
        args = (t(cam.bg), t(cloud["means3D"]), e, t(cloud["opacities"]), t(cloud["scales"]), t(cloud["rotations"]), 1.0, e, t(cam.viewmatrix),
                t(cam.projmatrix), cam.tanfovx, cam.tanfovy, H, W, t(cloud["shs"]), 2, t(cam.campos), False, False)
        return args, cam, t(scenes.upstream_gradient(W, H, seed=seed))

    jobs = [dict(scene=make(3, 208, 144, 9000), fw=dict(pruning=False, sort_lds_cap=64), bw=dict(deterministic=True)),
            dict(scene=make(4, 176, 160, 7000), fw=dict(), bw=dict())]

    def run(job, stream=None):
        args, cam, dL = job["scene"]
        with torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream()):
            R, color, radii, geom, binning, img, true_R, _tiles = _C.rasterize_gaussians(*args, info=True, **job["fw"])
            g = _C.rasterize_gaussians_backward(args[0], args[1], radii, e, args[4], args[5], 1.0, e, args[8], args[9], cam.tanfovx, cam.tanfovy, dL, args[14], 2,
                                                args[16], geom, R, binning, img, False, **job["bw"])
            n_contrib = _C.state_field("n_contrib", args[1].shape[0], args[13], args[12], R, True, True, geom, binning, img)
        torch.cuda.current_stream().synchronize() if stream is None else stream.synchronize()
        return R, color.clone(), n_contrib.clone(), [x.clone() for x in g]

    alone = [run(j) for j in jobs]
    assert alone[0][0] != alone[1][0]
    # pruning off really is what job 0 got (more instances than with the default) -- otherwise the test could not see a leak
    R_pruned = _C.rasterize_gaussians(*jobs[0]["scene"][0])[0]
    assert alone[0][0] > R_pruned
    results, errors = [None, None], []

    def worker(i, reps=12):
        try:
            st = torch.cuda.Stream(device=gpu_device)
            for _ in range(reps):
                results[i] = run(jobs[i], st)
                R, color, nc, g = results[i]
                assert R == alone[i][0]
                assert torch.equal(color, alone[i][1]) and torch.equal(nc, alone[i][2])
                if i == 0:          # fixed-order backward: bitwise
                    assert all(torch.equal(a, b) for a, b in zip(g, alone[i][3]))
                else:
                    assert all(util.rel_l2(a.cpu().numpy(), b.cpu().numpy()) <= 1e-5 for a, b in zip(g, alone[i][3]))
        except Exception as ex:      # noqa: BLE001 (reported below)
            errors.append((i, repr(ex)))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors


def test_example_training_loop_converges(gpu_device):
    """examples/train_views.py: cameras once -> run_views -> fused loss for all views -> batched backward -> Adam; the loss falls."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("train_views", os.path.join(util.ROOT, "examples", "train_views.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    vals = mod.run(steps=25, P=8000, W=192, H=128, V=6, log=lambda *_: None)
    assert len(vals) == 25 and all(np.isfinite(vals))
    assert vals[-1] < 0.6 * vals[0], (vals[0], vals[-1])
