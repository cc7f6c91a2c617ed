#!/usr/bin/env python3
"""BUILD CONTAINER ONLY.  Drives the reference's own getWorld2View / getProjectionMatrix (Edit_core/utils/graphics_utils.py,
imported from /root/reference) through the per-render camera set-up of TetGS.render_image_gaussian_rasterizer
(tetgs_scene/tetgs_model.py:478-502: axis flip, inverse, world-view transform, projection with the principal-point
entries, their product) in torch float32 on the CPU, and records inputs and results into
tests/golden/ref_camera_fixture.npz for youreditableavatar_amd.cameras.  A fixture is data."""
import math
import os
import sys
import types

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_camera_fixture.npz")


def main():
    p3d = types.ModuleType("pytorch3d"); tr = types.ModuleType("pytorch3d.transforms")
    tr.matrix_to_quaternion = lambda m: None          # imported by graphics_utils, not used here
    sys.modules["pytorch3d"] = p3d; sys.modules["pytorch3d.transforms"] = tr
    sys.path.insert(0, "/root/reference/Edit_core")
    from utils import graphics_utils as gu
    rng = np.random.Generator(np.random.PCG64(21))
    n = 5
    c2ws, Ks, views, fulls = [], [], [], []
    znear, zfar = 0.01, 100.0
    fov_x, fov_y = 2 * math.atan(1920 / (2 * 1400.0)), 2 * math.atan(1080 / (2 * 1400.0))
    for i in range(n):
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] *= -1
        c2w34 = np.concatenate([q, rng.standard_normal((3, 1)) * 3], 1).astype(np.float32)
        K02, K12 = (0.0, 0.0) if i == 0 else rng.uniform(-0.05, 0.05, 2)
        m = torch.cat([torch.tensor(c2w34), torch.tensor([[0.0, 0.0, 0.0, 1.0]])], 0)
        m[:3, 1:3] *= -1
        w2c = torch.inverse(m)
        view = torch.Tensor(gu.getWorld2View(R=w2c[:3, :3].T, t=w2c[:3, 3], tensor=True)).transpose(0, 1)
        proj = gu.getProjectionMatrix(znear, zfar, fov_x, fov_y).transpose(0, 1).clone()
        proj[2, 0] = -K02
        proj[2, 1] = -K12
        full = view.unsqueeze(0).bmm(proj.unsqueeze(0)).squeeze(0)
        c2ws.append(c2w34); Ks.append([K02, K12]); views.append(view.numpy()); fulls.append(full.numpy())
    np.savez_compressed(OUT, c2w=np.stack(c2ws), principal=np.array(Ks, np.float64), znear=znear, zfar=zfar, fov_x=fov_x, fov_y=fov_y,
                        view=np.stack(views), full=np.stack(fulls))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
