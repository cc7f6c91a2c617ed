"""Torch restatement of the binding-side properties of the reference's model classes (TEST INFRASTRUCTURE ONLY).

Edit_core/tetgs_scene/tetgs_model.py: strengths :261-265, scaling :279-281 (scale_activation = torch.exp, :16), quaternions :283-286,
points :252-258.  **Parity unpinned**: the model classes import pytorch3d / open3d, which are absent here, so they cannot be imported to
generate a fixture; the four properties are single calls into torch (sigmoid, exp, F.normalize, a fused multiply-add), restated verbatim."""
import torch


def bind(all_densities=None, raw_scales=None, raw_quaternions=None, ori_points=None, normals=None, offsets=None):
    strengths = torch.sigmoid(all_densities.view(-1, 1)) if all_densities is not None else None                  # :265
    scaling = torch.exp(raw_scales) if raw_scales is not None else None                                          # :280
    quaternions = torch.nn.functional.normalize(raw_quaternions, dim=-1) if raw_quaternions is not None else None   # :286
    points = ori_points + normals * offsets if ori_points is not None else None                                  # :257
    return strengths, scaling, quaternions, points
