"""Torch restatement of the binding-side properties of the reference's model classes (TEST INFRASTRUCTURE ONLY).

Edit_core/tetgs_scene/tetgs_model.py: strengths :261-265, scaling :279-281 (scale_activation = torch.exp, :16), quaternions :283-286,
points :252-258; the two-group forms of tetgs_edit_2d.py:280-318 and tetgs_edit_3d.py:272-331 (torch.cat([keep, edit]) first).
Pinned by tests/golden/ref_bind_fixture.npz: outputs and autograd gradients of the reference's OWN classes, imported in the build
container with stubs for the packages it cannot load (tests/make_ref_bind_fixture.py); tests/test_bind.py checks this restatement
against that fixture on the CPU and the HIP ops against both."""
import torch


def bind(all_densities=None, raw_scales=None, raw_quaternions=None, ori_points=None, normals=None, offsets=None):
    strengths = torch.sigmoid(all_densities.view(-1, 1)) if all_densities is not None else None                  # :265
    scaling = torch.exp(raw_scales) if raw_scales is not None else None                                          # :280
    quaternions = torch.nn.functional.normalize(raw_quaternions, dim=-1) if raw_quaternions is not None else None   # :286
    points = ori_points + normals * offsets if ori_points is not None else None                                  # :257
    return strengths, scaling, quaternions, points


def bind_groups(keep_points, keep_densities, keep_scales, keep_quaternions, edit_densities, edit_scales, edit_quaternions,
                edit_points=None, ori_edit_points=None, edit_normals=None, edit_offsets=None):
    """EditTetGS (edit_points [Pe,3], tetgs_edit_2d.py:281-318) / Edit3DTetGS (ori + normals * offsets, tetgs_edit_3d.py:273-331)"""
    new_edit = edit_points if edit_points is not None else ori_edit_points + edit_normals * edit_offsets
    points = torch.cat([keep_points, new_edit], dim=0)
    strengths = torch.sigmoid(torch.cat([keep_densities, edit_densities], dim=0).view(-1, 1))
    scaling = torch.exp(torch.cat([keep_scales, edit_scales], dim=0))
    quaternions = torch.nn.functional.normalize(torch.cat([keep_quaternions, edit_quaternions], dim=0), dim=-1)
    return strengths, scaling, quaternions, points
