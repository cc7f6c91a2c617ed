#!/usr/bin/env python3
"""BUILD CONTAINER ONLY.  Imports the reference's importable Python helpers from /root/reference
(Edit_core/utils/graphics_utils.py with a stub for pytorch3d.transforms, spherical_harmonics.py) and
records input/output pairs into tests/golden/ref_utils_fixture.npz.  The fixture pins
youreditableavatar_amd.scenes (camera assembly, caller-side SH->RGB) to what the reference's callers
feed the rasterizer (tetgs_model.py:480-537).  A fixture is data; no reference source is copied."""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/Edit_core"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_utils_fixture.npz")


def main():
    p3d = types.ModuleType("pytorch3d"); tr = types.ModuleType("pytorch3d.transforms")
    tr.matrix_to_quaternion = lambda m: None      # only imported, never called by the helpers used here
    sys.modules["pytorch3d"] = p3d; sys.modules["pytorch3d.transforms"] = tr
    sys.path.insert(0, REF)
    from utils import graphics_utils as gu
    from utils import spherical_harmonics as shu
    rng = np.random.Generator(np.random.PCG64(5))
    out = {}
    # projection matrices
    cases = [(1e-4, 100.0, 0.7, 0.5), (0.01, 50.0, 1.2, 0.9), (1e-4, 100.0, 2 * np.arctan(1920 / (2 * 1303.6)), np.pi / 4)]
    out["proj_args"] = np.array(cases, np.float64)
    out["proj_out"] = np.stack([gu.getProjectionMatrix(*c).numpy() for c in cases])
    # world-to-view
    Rs, ts, w2v = [], [], []
    for _ in range(3):
        q, _r = np.linalg.qr(rng.standard_normal((3, 3)))
        t = rng.standard_normal(3)
        Rs.append(q); ts.append(t); w2v.append(gu.getWorld2View(q, t))
    out["w2v_R"] = np.stack(Rs); out["w2v_t"] = np.stack(ts); out["w2v_out"] = np.stack(w2v)
    out["focal2fov_args"] = np.array([[1303.6, 1080.0], [800.0, 800.0]]); out["focal2fov_out"] = np.array([gu.focal2fov(a, b) for a, b in out["focal2fov_args"]])
    # eval_sh per degree: sh laid out [..., C, (deg+1)^2] in the reference helper
    P = 64
    sh = rng.standard_normal((P, 16, 3)).astype(np.float32)
    d = rng.standard_normal((P, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    out["sh_coeffs"] = sh; out["sh_dirs"] = d
    for deg in range(4):
        n = (deg + 1) ** 2
        res = shu.eval_sh(deg, torch.from_numpy(sh[:, :n, :]).transpose(1, 2), torch.from_numpy(d))
        out[f"sh_out_deg{deg}"] = res.numpy()
    # get_points_rgb (tetgs_model.py:413-442) composed from the reference's own eval_sh, with autograd gradients
    Pn = 200
    shc = torch.tensor(rng.standard_normal((Pn, 16, 3)) * 0.6, dtype=torch.float32, requires_grad=True)
    pos = torch.tensor(rng.standard_normal((Pn, 3)), dtype=torch.float32, requires_grad=True)
    cam = torch.tensor([[0.3, -2.0, 1.5]], dtype=torch.float32)
    gcol = torch.tensor(rng.standard_normal((Pn, 3)), dtype=torch.float32)
    out["rgb_sh"] = shc.detach().numpy(); out["rgb_pos"] = pos.detach().numpy(); out["rgb_cam"] = cam.numpy(); out["rgb_gcol"] = gcol.numpy()
    for levels in (1, 2, 3, 4):
        for t in (shc, pos):
            t.grad = None
        dirs = torch.nn.functional.normalize(pos - cam, dim=-1)
        view = shc[:, :levels ** 2].transpose(-1, -2).reshape(-1, 3, levels ** 2)
        colors = torch.clamp_min(shu.eval_sh(levels - 1, view, dirs) + 0.5, 0.0).view(-1, 3)
        colors.backward(gcol)
        out[f"rgb_colors_l{levels}"] = colors.detach().numpy()
        out[f"rgb_dsh_l{levels}"] = shc.grad.numpy().copy()
        out[f"rgb_dpos_l{levels}"] = pos.grad.numpy().copy() if pos.grad is not None else np.zeros((Pn, 3), np.float32)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
