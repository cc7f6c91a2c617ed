"""Camera set-up computed once ("next" row 3): host arithmetic pinned to the reference's helpers, device views on the GPU."""
import os

import numpy as np
import pytest
import torch

from tests import util

FX = os.path.join(util.GOLDEN_DIR, "ref_camera_fixture.npz")


def _build(device):
    from youreditableavatar_amd.cameras import RasterCameras
    fx = np.load(FX)
    cams = RasterCameras.from_camera_to_worlds(fx["c2w"], float(fx["znear"]), float(fx["zfar"]), float(fx["fov_x"]), float(fx["fov_y"]), 1080, 1920,
                                               principal_ndc=fx["principal"], device=device)
    return fx, cams


def test_matches_reference_camera_setup():
    fx, cams = _build("cpu")
    assert len(cams) == 5
    np.testing.assert_allclose(cams.viewmatrix.numpy(), fx["view"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(cams.projmatrix.numpy(), fx["full"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(cams.campos.numpy(), fx["c2w"][:, :, 3], rtol=0, atol=0)
    # the camera centre is where the view transform maps to the origin
    for i in range(5):
        c = np.append(cams.campos[i].numpy(), 1.0) @ cams.viewmatrix[i].numpy()
        np.testing.assert_allclose(c[:3], 0.0, atol=2e-5)
    assert abs(cams.tanfovx - 1920 / 2800.0) < 1e-6 and abs(cams.tanfovy - 1080 / 2800.0) < 1e-6


def test_rejects_bad_shapes():
    from youreditableavatar_amd.cameras import RasterCameras
    with pytest.raises(ValueError):
        RasterCameras.from_camera_to_worlds(np.zeros((2, 4, 4), np.float32), 0.01, 100.0, 1.0, 1.0, 8, 8, device="cpu")


@pytest.mark.gpu
def test_settings_are_device_views_and_render(gpu_device):
    from diff_gaussian_rasterization import GaussianRasterizer
    from youreditableavatar_amd import scenes
    from youreditableavatar_amd.cameras import RasterCameras
    cam = scenes.orbit_camera(176, 112, azimuth_deg=40.0)
    # rebuild the orbit camera from its camera-to-world transform (OpenGL axes) and compare renders
    w2c = cam.viewmatrix.T.astype(np.float64)                    # row-vector convention -> column convention
    c2w = np.linalg.inv(w2c)
    c2w[:3, 1:3] *= -1
    import math
    cams = RasterCameras.from_camera_to_worlds(c2w[None, :3, :].astype(np.float32), 0.01, 100.0, 2 * math.atan(cam.tanfovx), 2 * math.atan(cam.tanfovy),
                                               112, 176, device=gpu_device)
    bg = torch.tensor(cam.bg, device=gpu_device)
    rs = cams.settings(0, bg, 3)
    assert rs.viewmatrix.data_ptr() == cams.viewmatrix[0].data_ptr() and rs.viewmatrix.is_cuda
    cloud = scenes.make_cloud(3000, 3, seed=3, scale_mult=3.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(gpu_device)
    kw = dict(means3D=t(cloud["means3D"]), means2D=torch.zeros(3000, 3, device=gpu_device), opacities=t(cloud["opacities"]), shs=t(cloud["shs"]),
              scales=t(cloud["scales"]), rotations=t(cloud["rotations"]))
    img, _ = GaussianRasterizer(rs)(**kw)
    from tests.test_gpu_api import _settings
    ref, _ = GaussianRasterizer(_settings(cam, 3, gpu_device))(**kw)
    assert util.rel_l2(img.cpu().numpy(), ref.cpu().numpy()) <= 1e-4
