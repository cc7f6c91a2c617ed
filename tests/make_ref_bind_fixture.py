#!/usr/bin/env python3
"""BUILD CONTAINER ONLY.  Imports the reference's three model classes (Edit_core/tetgs_scene/tetgs_model.py: TetGS,
tetgs_edit_2d.py: EditTetGS, tetgs_edit_3d.py: Edit3DTetGS) from /root/reference with stub modules in place of the packages this
image lacks (open3d, pytorch3d, PIL-based loaders -- none of them is touched by the properties exercised here), builds instances
with ``object.__new__`` carrying only the parameters those properties read, and records what the classes' OWN code returns for

    points / strengths / scaling / quaternions / sh_coordinates (keep_ / edit_)      the per-step binding of the rasterizer's inputs
    get_points_rgb(...) exactly as render_image_gaussian_rasterizer calls it         (tetgs_model.py:524-531, tetgs_edit_2d.py:548-565,
                                                                                       tetgs_edit_3d.py:566-582 incl. the [N,1] positions)

together with the autograd gradients of those outputs for seeded upstream gradients, in float64 (the reference code is dtype-agnostic)
-- into tests/golden/ref_bind_fixture.npz.  A fixture is data: inputs and expected outputs."""
import os
import sys
import types

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_bind_fixture.npz")


class _Anything:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return None


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything


def _stub_modules():
    for name in ("open3d", "pytorch3d", "pytorch3d.renderer", "pytorch3d.renderer.cameras", "pytorch3d.structures", "pytorch3d.transforms", "pytorch3d.ops",
                 "diff_gaussian_rasterization", "tetgs_scene.gs_model", "tetgs_scene.cameras", "PIL"):
        sys.modules[name] = _Stub(name)
    sys.path.insert(0, "/root/reference/Edit_core")


def _bare(cls, **attrs):
    obj = object.__new__(cls)
    torch.nn.Module.__init__(obj)
    obj.nerfmodel = types.SimpleNamespace(device=torch.device("cpu"))
    for k, v in attrs.items():
        object.__setattr__(obj, k, v) if not isinstance(v, torch.nn.Parameter) else setattr(obj, k, v)
    return obj


def main():
    _stub_modules()
    from tetgs_scene.tetgs_model import TetGS, scale_activation
    from tetgs_scene.tetgs_edit_2d import EditTetGS
    from tetgs_scene.tetgs_edit_3d import Edit3DTetGS
    rng = np.random.Generator(np.random.PCG64(404))
    f = lambda *s: rng.standard_normal(s)
    par = lambda a, grad=True: torch.nn.Parameter(torch.tensor(a, dtype=torch.float64), requires_grad=grad)
    unit = lambda a: a / np.linalg.norm(a, axis=-1, keepdims=True)
    rec = {}

    def logscales(P):
        s = np.log(np.abs(f(P, 3)) * 0.02 + 1e-3)
        s[::5, 0] = np.log(1e-8)                           # the flat axis of mesh-bound Gaussians (tetgs_edit_2d.py:203)
        return s

    def record(prefix, inputs, outputs, upstream):
        """inputs: {name: Parameter}; outputs: {name: tensor}; upstream: {name: ndarray} -> gradients of sum(out * upstream) on every learnable input"""
        loss = sum((outputs[k] * torch.tensor(upstream[k], dtype=torch.float64)).sum() for k in outputs)
        learn = {k: v for k, v in inputs.items() if v.requires_grad}
        grads = torch.autograd.grad(loss, list(learn.values()), allow_unused=True)
        for k, v in inputs.items():
            rec[f"{prefix}.in.{k}"] = v.detach().numpy()
        for k, v in outputs.items():
            rec[f"{prefix}.out.{k}"] = v.detach().numpy()
            rec[f"{prefix}.up.{k}"] = upstream[k]
        for (k, _), g in zip(learn.items(), grads):
            rec[f"{prefix}.grad.{k}"] = np.zeros(tuple(inputs[k].shape)) if g is None else g.numpy()

    cam = np.array([[0.4, -2.9, 0.7]])

    # ---- TetGS (tetgs_model.py:252-286, 413-442), mesh-bound with update_normal: _points is [P,1] ----
    P = 301
    base_in = dict(_points=par(f(P, 1) * 0.01), ori_points=par(f(P, 3), False), normals=par(unit(f(P, 3)), False), all_densities=par(f(P, 1) * 2),
                   _scales=par(logscales(P)), _quaternions=par(f(P, 4)), _sh_coordinates_dc=par(f(P, 1, 3)), _sh_coordinates_rest=par(f(P, 15, 3) * 0.2))
    m = _bare(TetGS, update_normal=True, return_one_densities=False, sh_levels=4, scale_activation=scale_activation, **base_in)
    for levels in (1, 3, 4):
        outs = dict(points=m.points, strengths=m.strengths, scaling=m.scaling, quaternions=m.quaternions,
                    colors=m.get_points_rgb(positions=m.points, camera_centers=torch.tensor(cam), sh_levels=levels))
        record(f"tetgs.L{levels}", base_in, outs, {k: f(*v.shape) for k, v in outs.items()})
    rec["tetgs.sh_coordinates"] = m.sh_coordinates.detach().numpy()

    # ---- EditTetGS (tetgs_edit_2d.py:280-318, 419-449, 548-565): keep group frozen, edit group learnable, positions [Pe,3] ----
    Pk, Pe = 157, 211
    keep = dict(_keep_points=par(f(Pk, 3), False), all_keep_densities=par(f(Pk, 1) * 2, False), _keep_scales=par(logscales(Pk), False),
                _keep_quaternions=par(f(Pk, 4), False), _keep_sh_coordinates_dc=par(f(Pk, 1, 3), False), _keep_sh_coordinates_rest=par(f(Pk, 15, 3) * 0.2, False))
    edit2 = dict(_edit_points=par(f(Pe, 3)), all_edit_densities=par(f(Pe, 1) * 2), _edit_scales=par(logscales(Pe)), _edit_quaternions=par(f(Pe, 4)),
                 _edit_sh_coordinates_dc=par(f(Pe, 1, 3)))
    e2 = _bare(EditTetGS, return_one_densities=False, keep_sh_levels=4, edit_sh_levels=1, scale_activation=scale_activation, **keep, **edit2)

    def edit_outputs(model, edit_levels):
        keep_c = model.get_points_rgb(positions=model._keep_points, camera_centers=torch.tensor(cam), sh_levels=model.keep_sh_levels, sh_coordinates=model.keep_sh_coordinates)
        edit_c = model.get_points_rgb(positions=model._edit_points, camera_centers=torch.tensor(cam), sh_levels=edit_levels, sh_coordinates=model.edit_sh_coordinates)
        return dict(points=model.points, strengths=model.strengths, scaling=model.scaling, quaternions=model.quaternions, colors=torch.cat([keep_c, edit_c], dim=0))

    outs = edit_outputs(e2, e2.edit_sh_levels)
    record("edit2d", {**keep, **edit2}, outs, {k: f(*v.shape) for k, v in outs.items()})
    rec["edit2d.keep_sh_coordinates"] = e2.keep_sh_coordinates.detach().numpy()

    # ---- Edit3DTetGS (tetgs_edit_3d.py:272-331, 566-582): edit positions = ori + normal * offset; the colour call gets the [Pe,1] offsets ----
    edit3 = dict(_edit_points=par(f(Pe, 1) * 0.01), ori_edit_points=par(f(Pe, 3), False), _edit_normals=par(unit(f(Pe, 3)), False), all_edit_densities=par(f(Pe, 1) * 2),
                 _edit_scales=par(logscales(Pe)), _edit_quaternions=par(f(Pe, 4)), _edit_sh_coordinates_dc=par(f(Pe, 1, 3)), _edit_sh_coordinates_rest=par(f(Pe, 15, 3) * 0.2))
    e3 = _bare(Edit3DTetGS, update_normal=True, bind_3dgs=True, return_one_densities=False, keep_sh_levels=4, edit_sh_levels=4, scale_activation=scale_activation,
               **keep, **edit3)
    for levels in (1, 4):                                  # sh_deg + 1 of the refinement schedule (tetgs_edit_3d.py:577)
        outs = edit_outputs(e3, levels)
        record(f"edit3d.L{levels}", {**keep, **edit3}, outs, {k: f(*v.shape) for k, v in outs.items()})
    rec["camera_center"] = cam
    np.savez_compressed(OUT, **rec)
    print("wrote", OUT, len(rec), "arrays,", os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
