"""The on-disk format either side of the path (last paragraph): what the reference's model classes save and load.

``TetGS.save_model`` (Edit_core/tetgs_scene/tetgs_model.py:635-640) writes ``torch.save({'state_dict': self.state_dict(), **kwargs})``; the
Gaussian part of that state dict is the raw parameter set the rasterizer call is assembled from (``:196-239``, ``:158-170``):

    all_densities [P,1]   densities before the sigmoid                 _scales [P,3]       log-scales
    _quaternions [P,4]    unnormalised (r, x, y, z)                     _sh_coordinates_dc [P,1,3] / _sh_coordinates_rest [P,M-1,3]
    _points               [P,3] positions, or [P,1] offsets along ``normals`` from ``ori_points`` (mesh-bound models with update_normal)

``convert_refined_tetgs_into_masked_gaussians`` (``:678-736``) hands the same quantities to the editing stages as a dict of ``keep_*``
tensors.  ``GaussianState`` reads either and turns it into rasterizer inputs with the fused ops of this package -- without the model
classes themselves, which need pytorch3d / open3d.  Host logic: loading runs anywhere; ``rasterizer_inputs`` needs the HIP device.

The editing stages' own checkpoints hold TWO groups and none of the keys above (``EditTetGS`` tetgs_edit_2d.py:118-262, ``Edit3DTetGS``
tetgs_edit_3d.py:105-258; read back at tetgs_edit_2d.py:685-693):

    _keep_points [Pk,3]  all_keep_densities  _keep_scales  _keep_quaternions  _keep_sh_coordinates_dc / _rest        frozen
    _edit_points [Pe,3] (2D)  or  [Pe,1] offsets + ori_edit_points [Pe,3] + _edit_normals [Pe,3] (3D)
    all_edit_densities  _edit_scales  _edit_quaternions  _edit_sh_coordinates_dc ( / _rest)                          learnable

``from_state_dict`` returns a ``GroupedGaussianState`` for those (keep rows first, like the classes' ``torch.cat([keep, edit])``).
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


@dataclass
class GroupedGaussianState:
    """keep + edit group of an EditTetGS / Edit3DTetGS checkpoint.  ``edit.points`` [Pe,3] (2D stage) or ``edit.offsets`` [Pe,1] +
    ``edit.ori_points`` + ``edit.normals`` (3D stage)."""
    keep: GaussianState
    edit: GaussianState

    @property
    def n_points(self) -> int:
        return self.keep.n_points + self.edit.n_points

    def to(self, device) -> "GroupedGaussianState":
        return GroupedGaussianState(self.keep.to(device), self.edit.to(device))

    def rasterizer_inputs(self, camera_center: torch.Tensor, edit_sh_levels: Optional[int] = None) -> Dict[str, torch.Tensor]:
        """means3D / opacities / scales / rotations / colors_precomp as ``render_image_gaussian_rasterizer`` of the two classes assembles them
        (tetgs_edit_2d.py:536-575, tetgs_edit_3d.py:554-592), through ``gaussian_bind_groups`` and ``points_rgb_groups``.  ``edit_sh_levels``:
        the 3D stage's ``sh_deg + 1`` schedule (tetgs_edit_3d.py:577); default: all levels the edit group holds."""
        from .bindings import gaussian_bind_groups
        from .sh_color import points_rgb_groups
        k, e = self.keep, self.edit
        pos = dict(edit_points=e.points) if e.offsets is None else dict(ori_edit_points=e.ori_points, edit_normals=e.normals, edit_offsets=e.offsets)
        op, sc, qu, pts = gaussian_bind_groups(keep_points=k.points, keep_densities=k.all_densities, keep_scales=k.scales_raw, keep_quaternions=k.quaternions_raw,
                                               edit_densities=e.all_densities, edit_scales=e.scales_raw, edit_quaternions=e.quaternions_raw, **pos)
        elev = e.sh_levels if edit_sh_levels is None else int(edit_sh_levels)
        # the classes hand get_points_rgb the RAW parameter as the edit positions: [Pe,3] positions (2D) or the [Pe,1] offsets (3D, tetgs_edit_3d.py:556)
        colors = points_rgb_groups(keep_sh_dc=k.sh_dc, keep_sh_rest=k.sh_rest, keep_sh_levels=k.sh_levels, keep_positions=k.points,
                                   edit_sh_dc=e.sh_dc, edit_sh_rest=e.sh_rest if elev > 1 else None, edit_sh_levels=elev,
                                   edit_positions=e.points if e.offsets is None else e.offsets, camera_centers=camera_center)
        return dict(means3D=pts, opacities=op, scales=sc, rotations=qu, colors_precomp=colors)


_GROUP_KEYS = {"keep": {"_keep_points": "_points", "all_keep_densities": "all_densities", "_keep_scales": "_scales", "_keep_quaternions": "_quaternions",
                        "_keep_sh_coordinates_dc": "_sh_coordinates_dc", "_keep_sh_coordinates_rest": "_sh_coordinates_rest"},
               "edit": {"_edit_points": "_points", "all_edit_densities": "all_densities", "_edit_scales": "_scales", "_edit_quaternions": "_quaternions",
                        "_edit_sh_coordinates_dc": "_sh_coordinates_dc", "_edit_sh_coordinates_rest": "_sh_coordinates_rest",
                        "ori_edit_points": "ori_points", "_edit_normals": "normals"}}


def from_state_dict(sd: Dict[str, torch.Tensor]):
    """A ``state_dict`` of TetGS, or the ``keep_*`` dict of the editing stages -> GaussianState; a ``state_dict`` of EditTetGS /
    Edit3DTetGS (``_keep_*`` + ``_edit_*`` keys) -> GroupedGaussianState."""
    if "_keep_points" in sd or "_edit_points" in sd:
        parts = {}
        for grp, names in _GROUP_KEYS.items():
            sub = {new: sd[old] for old, new in names.items() if old in sd}
            if "_points" not in sub:
                raise KeyError(f"an editing-stage checkpoint needs both groups: no _{grp}_points")
            parts[grp] = from_state_dict(sub)
        if parts["keep"].points is None:
            raise ValueError("_keep_points must be [Pk,3] positions")
        return GroupedGaussianState(parts["keep"], parts["edit"])
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
