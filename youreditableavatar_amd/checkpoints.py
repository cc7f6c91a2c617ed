"""The on-disk format either side of the path (last paragraph): what the reference's model classes save and load.

``TetGS.save_model`` (Edit_core/tetgs_scene/tetgs_model.py:635-640) writes ``torch.save({'state_dict': self.state_dict(), **kwargs})``; the
Gaussian part of that state dict is the raw parameter set the rasterizer call is assembled from (``:196-239``, ``:158-170``):

    all_densities [P,1]   densities before the sigmoid                 _scales [P,3]       log-scales
    _quaternions [P,4]    unnormalised (r, x, y, z)                     _sh_coordinates_dc [P,1,3] / _sh_coordinates_rest [P,M-1,3]
    _points               [P,3] positions, or [P,1] offsets along ``normals`` from ``ori_points`` (mesh-bound models with update_normal)

``convert_refined_tetgs_into_masked_gaussians`` (``:678-736``) hands the same quantities to the editing stages as a dict of ``keep_*``
tensors.  ``GaussianState`` reads either and turns it into rasterizer inputs with the fused ops of this package -- without the model
classes themselves, which need pytorch3d / open3d.  Host logic: loading runs anywhere; ``rasterizer_inputs`` needs the HIP device.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch

_KEEP = {"keep_xyz": "points", "keep_opacities": "all_densities", "keep_scales": "_scales", "keep_rots": "_quaternions",
         "keep_sh_coordinates_dc": "_sh_coordinates_dc", "keep_sh_coordinates_rest": "_sh_coordinates_rest"}


@dataclass
class GaussianState:
    all_densities: torch.Tensor                  # [P,1]
    scales_raw: torch.Tensor                     # [P,3]  log-scales
    quaternions_raw: torch.Tensor                # [P,4]
    sh_dc: torch.Tensor                          # [P,1,3]
    sh_rest: Optional[torch.Tensor]              # [P,M-1,3] or None (one-level models)
    points: Optional[torch.Tensor] = None        # [P,3] positions (models that learn positions, or a keep_* dict)
    ori_points: Optional[torch.Tensor] = None    # [P,3]  } mesh-bound models with update_normal:
    normals: Optional[torch.Tensor] = None       # [P,3]  }   points = ori_points + normals * offsets
    offsets: Optional[torch.Tensor] = None       # [P,1]  }

    @property
    def n_points(self) -> int:
        return int(self.scales_raw.shape[0])

    @property
    def sh_levels(self) -> int:
        m = 1 + (int(self.sh_rest.shape[1]) if self.sh_rest is not None else 0)
        return int(round(m ** 0.5))

    def to(self, device) -> "GaussianState":
        mv = lambda t: None if t is None else t.to(device=device, dtype=torch.float32).contiguous()
        return GaussianState(*(mv(getattr(self, f)) for f in self.__dataclass_fields__))

    def rasterizer_inputs(self, camera_center: torch.Tensor, sh_levels: Optional[int] = None) -> Dict[str, torch.Tensor]:
        """means3D / opacities / scales / rotations / colors_precomp for ``GaussianRasterizer`` in the reference's training mode
        (tetgs_model.py:524-551: colours outside the rasterizer), through ``gaussian_bind`` and ``points_rgb_dc_rest``."""
        from .bindings import gaussian_bind
        from .sh_color import points_rgb_dc_rest
        op, sc, qu, pts = gaussian_bind(self.all_densities, self.scales_raw, self.quaternions_raw, self.ori_points, self.normals, self.offsets)
        means = pts if pts is not None else self.points
        levels = self.sh_levels if sh_levels is None else int(sh_levels)
        colors = points_rgb_dc_rest(self.sh_dc, self.sh_rest if levels > 1 else None, levels, positions=means, camera_centers=camera_center)
        return dict(means3D=means, opacities=op, scales=sc, rotations=qu, colors_precomp=colors)


def from_state_dict(sd: Dict[str, torch.Tensor]) -> GaussianState:
    """A ``state_dict`` of TetGS / EditTetGS / Edit3DTetGS (or the ``keep_*`` dict of the editing stages) -> GaussianState."""
    if any(k in sd for k in _KEEP):
        sd = {_KEEP.get(k, k): v for k, v in sd.items()}
    need = ("all_densities", "_scales", "_quaternions", "_sh_coordinates_dc")
    missing = [k for k in need if k not in sd]
    if missing:
        raise KeyError(f"not a TetGS Gaussian state: missing {missing}")
    P = int(sd["_scales"].shape[0])
    f = lambda k: None if sd.get(k) is None else torch.as_tensor(sd[k]).detach().float()
    pts, ori, nrm, off = f("points"), f("ori_points"), f("normals"), None
    raw = f("_points")
    if raw is not None:
        if raw.dim() == 2 and raw.shape[1] == 1 and ori is not None and nrm is not None:
            off = raw                                        # offsets along the face normals (tetgs_model.py:168-170)
        elif raw.dim() == 2 and raw.shape[1] == 3:
            pts = raw
        else:
            raise ValueError(f"_points of shape {tuple(raw.shape)} needs ori_points and normals beside it")
    if off is None and pts is None:
        raise KeyError("no positions: expected _points [P,3], or _points [P,1] + ori_points + normals, or keep_xyz")
    rest = f("_sh_coordinates_rest")
    st = GaussianState(f("all_densities").reshape(P, 1), f("_scales"), f("_quaternions"), f("_sh_coordinates_dc").reshape(P, 1, 3),
                       rest if (rest is not None and rest.numel()) else None, pts, ori if off is not None else None, nrm if off is not None else None, off)
    for name in ("scales_raw", "quaternions_raw", "sh_dc"):
        if getattr(st, name).shape[0] != P:
            raise ValueError(f"{name}: {getattr(st, name).shape[0]} rows, expected {P}")
    return st


def load(path: str, map_location="cpu") -> GaussianState:
    """``torch.load`` of a ``save_model`` checkpoint (tetgs_model.py:635-640) -> GaussianState."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    return from_state_dict(ck["state_dict"] if isinstance(ck, dict) and "state_dict" in ck else ck)
