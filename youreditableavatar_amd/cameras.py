"""Per-camera rasterizer inputs, computed once ("next" row 3 of SURVEY.md 8f).

``TetGS.render_image_gaussian_rasterizer`` rebuilds the camera on every call (tetgs_scene/tetgs_model.py:478-502,
same body in tetgs_edit_2d.py:472-515): a 4x4 inverse, ``getWorld2View`` / ``getProjectionMatrix`` on the host,
``.item()`` reads of znear / zfar (two synchronisations), ``.cuda()`` uploads and a ``bmm`` -- about ten tiny kernels,
six host-to-device copies and two syncs per render for data that never changes during training.  ``RasterCameras``
does that arithmetic once for all cameras (float32, the reference's order of operations) and keeps the results on the
device; ``settings(i, ...)`` hands out views of those tensors, so a render starts with no copy and no sync.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
import torch

from .scenes import projection_matrix, world_to_view


@dataclass
class RasterCameras:
    viewmatrix: torch.Tensor        # [n,4,4]  world_view_transform (transposed, as the rasterizer expects)
    projmatrix: torch.Tensor        # [n,4,4]  full_proj_transform
    campos: torch.Tensor            # [n,3]
    tanfovx: float
    tanfovy: float
    image_height: int
    image_width: int

    def __len__(self) -> int:
        return int(self.viewmatrix.shape[0])

    @staticmethod
    def from_camera_to_worlds(camera_to_worlds, znear: float, zfar: float, fov_x: float, fov_y: float, image_height: int, image_width: int,
                              principal_ndc: Optional[Sequence[Sequence[float]]] = None, device="cuda") -> "RasterCameras":
        """``camera_to_worlds``: [n,3,4] NeRF/OpenGL camera-to-world transforms (Y up, Z back), numpy or tensor -- what
        ``nerf_cameras.camera_to_worlds`` holds.  ``principal_ndc``: per camera (K[0,0,2], K[0,1,2]) of the pytorch3d
        camera (tetgs_model.py:499-500), zeros when absent."""
        c2w_all = np.asarray(camera_to_worlds.detach().cpu() if isinstance(camera_to_worlds, torch.Tensor) else camera_to_worlds, dtype=np.float32)
        if c2w_all.ndim != 3 or c2w_all.shape[1:] != (3, 4):
            raise ValueError(f"camera_to_worlds must be [n,3,4], got {c2w_all.shape}")
        n = c2w_all.shape[0]
        views, projs, centers = [], [], []
        P = projection_matrix(float(znear), float(zfar), float(fov_x), float(fov_y)).T.copy()        # :494-498
        for i in range(n):
            c2w = np.concatenate([c2w_all[i], np.array([[0, 0, 0, 1]], np.float32)], 0)              # :480-481
            c2w[:3, 1:3] *= -1                                                                      # OpenGL -> COLMAP axes, :483
            w2c = np.linalg.inv(c2w).astype(np.float32)                                             # :487
            R, T = w2c[:3, :3].T, w2c[:3, 3]                                                        # :488-489
            wv = world_to_view(R, T).T                                                              # :490-491
            proj = P.copy()
            if principal_ndc is not None:
                proj[2, 0] = -float(principal_ndc[i][0])                                            # :499-500
                proj[2, 1] = -float(principal_ndc[i][1])
            views.append(wv)
            projs.append((wv @ proj).astype(np.float32))                                            # :502
            centers.append(c2w_all[i][:, 3])                                                        # camera centre = translation of c2w (:504-506)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(np.stack(a), np.float32)).to(device)
        import math
        return RasterCameras(t(views), t(projs), t(centers), math.tan(fov_x * 0.5), math.tan(fov_y * 0.5), int(image_height), int(image_width))

    def settings(self, index: int, bg: torch.Tensor, sh_degree: int, scale_modifier: float = 1.0, prefiltered: bool = False, debug: bool = False):
        """``GaussianRasterizationSettings`` of camera ``index`` (tetgs_model.py:508-521): device views, nothing is copied."""
        from .diff_gaussian_rasterization import GaussianRasterizationSettings
        return GaussianRasterizationSettings(image_height=self.image_height, image_width=self.image_width, tanfovx=self.tanfovx, tanfovy=self.tanfovy,
                                             bg=bg, scale_modifier=scale_modifier, viewmatrix=self.viewmatrix[index], projmatrix=self.projmatrix[index],
                                             sh_degree=sh_degree, campos=self.campos[index], prefiltered=prefiltered, debug=debug)
