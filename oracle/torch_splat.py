"""Independent PyTorch-autograd point-splat (TEST INFRASTRUCTURE ONLY).

Purpose: (1) an *independent* check of the analytic backward restated in tgs_oracle.c -- the
forward below is written from the textbook formulas (Sigma = R S^2 R^T, EWA cov2D = J W Sigma W^T J^T
+ 0.3 I, front-to-back compositing) and differentiated by autograd in fp64, so a derivative that
was mis-transcribed from the reference cannot hide; (2) the "PyTorch autograd point-splat" CPU
baseline named by BASELINE.json's north_star (timed by bench.py next to the C oracle).

Conventions taken from the reference so that the two are comparable (file:line under
Edit_core/thirdparties/diff-gaussian-rasterization/cuda_rasterizer):
 * list membership = the 3-sigma radius tile rectangle (forward.cu:229-237, auxiliary.h:46-56), order =
   (depth, index) (rasterizer_impl.cu:98-109,303-308);
 * skip rules and termination of forward.cu:325-362; the min(0.99, .) clamp is straight-through
   (backward.cu:503-541 does not gate the gradient by it);
 * the x/z, y/z clamp of forward.cu:82-87 is treated as a constant when active (backward.cu:175-176);
 * the gradient w.r.t. ``means2D`` is dL/d(NDC xy) (backward.cu:460-461,545-546).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]
TILE = 16


def _eval_sh(deg, sh, d):
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    r = SH_C0 * sh[:, 0]
    if deg > 0:
        r = r - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = (r + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6]
             + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        r = (r + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
             + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
             + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
             + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return r


def splat(*, means3D, means2D, opacities, viewmatrix, projmatrix, campos, bg, tanfovx, tanfovy, image_height,
          image_width, sh_degree=0, shs=None, colors_precomp=None, scales=None, rotations=None,
          cov3D_precomp=None, scale_modifier=1.0, return_aux=False):
    """All tensor arguments are torch tensors of one dtype (float64 for checks, float32 for timing)."""
    dt = means3D.dtype
    H, W = int(image_height), int(image_width)
    P = means3D.shape[0]
    V = viewmatrix.to(dt)           # transposed w2c: p_view = [p,1] @ V
    PM = projmatrix.to(dt)
    ones = torch.ones(P, 1, dtype=dt)
    ph = torch.cat([means3D, ones], 1)
    p_view = ph @ V
    p_hom = ph @ PM
    p_w = 1.0 / (p_hom[:, 3:4] + 1e-7)
    ndc = p_hom[:, :2] * p_w + means2D[:, :2]
    tz = p_view[:, 2]
    valid = tz > 0.2

    if cov3D_precomp is not None:
        c = cov3D_precomp
        Sigma = torch.stack([torch.stack([c[:, 0], c[:, 1], c[:, 2]], -1),
                             torch.stack([c[:, 1], c[:, 3], c[:, 4]], -1),
                             torch.stack([c[:, 2], c[:, 4], c[:, 5]], -1)], -2)
    else:
        q = rotations
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        Rm = torch.stack([
            torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
            torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
            torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], -2)
        S = scale_modifier * scales
        RS = Rm * S[:, None, :]
        Sigma = RS @ RS.transpose(1, 2)

    fx = W / (2.0 * tanfovx)
    fy = H / (2.0 * tanfovy)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tzs = torch.where(valid, tz, torch.ones_like(tz))
    txtz = p_view[:, 0] / tzs
    tytz = p_view[:, 1] / tzs
    cx = (txtz < -limx) | (txtz > limx)
    cy = (tytz < -limy) | (tytz > limy)
    tx = torch.where(cx, (txtz.clamp(-limx, limx) * tzs).detach(), p_view[:, 0])
    ty = torch.where(cy, (tytz.clamp(-limy, limy) * tzs).detach(), p_view[:, 1])
    zero = torch.zeros_like(tzs)
    J = torch.stack([torch.stack([fx / tzs, zero, -fx * tx / (tzs * tzs)], -1),
                     torch.stack([zero, fy / tzs, -fy * ty / (tzs * tzs)], -1)], -2)      # [P,2,3]
    Wr = V[:3, :3].T                                                                      # w2c rotation
    A = J @ Wr                                                                            # [P,2,3]
    cov = A @ Sigma @ A.transpose(1, 2)
    a = cov[:, 0, 0] + 0.3
    b = cov[:, 0, 1]
    c2 = cov[:, 1, 1] + 0.3
    det = a * c2 - b * b
    valid = valid & (det != 0)
    dets = torch.where(valid, det, torch.ones_like(det))
    conic_a, conic_b, conic_c = c2 / dets, -b / dets, a / dets
    mid = 0.5 * (a + c2)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    # radius / rect decisions in float32 exactly like the reference would take them
    radius = torch.ceil(3.0 * torch.sqrt(lam.detach().float())).to(torch.int64)
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    pxf, pyf, rf = px.detach().float(), py.detach().float(), radius.float()
    rminx = ((pxf - rf) / TILE).to(torch.int64).clamp(0, gx)
    rminy = ((pyf - rf) / TILE).to(torch.int64).clamp(0, gy)
    rmaxx = ((pxf + rf + TILE - 1) / TILE).to(torch.int64).clamp(0, gx)
    rmaxy = ((pyf + rf + TILE - 1) / TILE).to(torch.int64).clamp(0, gy)
    valid = valid & ((rmaxx - rminx) * (rmaxy - rminy) > 0)
    radii = torch.where(valid, radius, torch.zeros_like(radius)).to(torch.int32)

    if colors_precomp is not None:
        rgb = colors_precomp
    else:
        d = means3D - campos.reshape(1, 3).to(dt)
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(_eval_sh(sh_degree, shs, d) + 0.5, 0.0)

    depth32 = tz.detach().float().numpy()
    order = np.lexsort((np.arange(P), depth32))           # (depth, idx) ascending
    order = order[valid.numpy()[order]]
    o_t = torch.from_numpy(order)
    rminx_o, rmaxx_o, rminy_o, rmaxy_o = rminx[o_t], rmaxx[o_t], rminy[o_t], rmaxy[o_t]

    out = torch.zeros(3, H, W, dtype=dt)
    n_contrib = np.zeros((H, W), np.int64)
    final_T = np.ones((H, W), np.float64)
    num_rendered = int(((rmaxx - rminx) * (rmaxy - rminy))[valid].sum())
    opac = opacities.reshape(-1)
    bgt = bg.to(dt).reshape(3, 1)
    for tyi in range(gy):
        for txi in range(gx):
            sel = (rminx_o <= txi) & (txi < rmaxx_o) & (rminy_o <= tyi) & (tyi < rmaxy_o)
            ids = o_t[sel]
            K = ids.numel()
            y0, x0 = tyi * TILE, txi * TILE
            hh, ww = min(TILE, H - y0), min(TILE, W - x0)
            ys, xs = torch.meshgrid(torch.arange(y0, y0 + hh), torch.arange(x0, x0 + ww), indexing="ij")
            pixx = xs.reshape(-1, 1).to(dt)
            pixy = ys.reshape(-1, 1).to(dt)
            if K == 0:
                out[:, y0:y0 + hh, x0:x0 + ww] = bgt.reshape(3, 1, 1).expand(3, hh, ww)
                continue
            dx = px[ids].reshape(1, K) - pixx
            dy = py[ids].reshape(1, K) - pixy
            power = -0.5 * (conic_a[ids] * dx * dx + conic_c[ids] * dy * dy) - conic_b[ids] * dx * dy
            G = torch.exp(torch.clamp(power, max=0.0))
            araw = opac[ids] * G
            alpha = araw + (torch.clamp(araw, max=0.99) - araw).detach()
            active = (power <= 0) & (alpha >= 1.0 / 255.0)
            al = torch.where(active, alpha, torch.zeros_like(alpha))
            T_after = torch.cumprod(1 - al, dim=1)
            T_before = torch.cat([torch.ones_like(T_after[:, :1]), T_after[:, :-1]], 1)
            term = (T_after < 1e-4) & active
            dead = torch.cumsum(term.to(torch.int64), 1) > 0          # entry at/after the first terminator
            contrib = active & ~dead
            w = torch.where(contrib, al * T_before, torch.zeros_like(al))
            Cpix = w @ rgb[ids]                                         # [npix,3]
            Tfin = torch.where(dead.any(1), (T_before * (term & (torch.cumsum(term.to(torch.int64), 1) == 1)).to(dt)).sum(1), T_after[:, -1])
            col = Cpix.T + Tfin.reshape(1, -1) * bgt
            out[:, y0:y0 + hh, x0:x0 + ww] = col.reshape(3, hh, ww)
            if return_aux:
                pos = torch.arange(1, K + 1).reshape(1, K) * contrib.to(torch.int64)
                n_contrib[y0:y0 + hh, x0:x0 + ww] = pos.max(1).values.reshape(hh, ww).numpy()
                final_T[y0:y0 + hh, x0:x0 + ww] = Tfin.detach().reshape(hh, ww).double().numpy()
    if return_aux:
        return out, radii, dict(n_contrib=n_contrib, final_T=final_T, num_rendered=num_rendered)
    return out, radii


def run_scene(cloud: dict, cam, dL: Optional[np.ndarray] = None, mode: str = "sh", cov_mode: str = "scale_rot",
              dtype=torch.float64) -> Dict[str, np.ndarray]:
    """Forward (+ autograd backward when dL is given) on a scenes.make_cloud / orbit_camera pair."""
    t = lambda a, g=False: torch.tensor(np.asarray(a), dtype=dtype, requires_grad=g)
    P = cloud["means3D"].shape[0]
    leaves = {"means3D": t(cloud["means3D"], True), "means2D": torch.zeros(P, 3, dtype=dtype, requires_grad=True),
              "opacities": t(cloud["opacities"], True)}
    kw = {}
    if mode == "sh":
        leaves["shs"] = t(cloud["shs"], True)
    else:
        from youreditableavatar_amd import scenes
        cp = cloud.get("colors_precomp")
        if cp is None:
            cp = scenes.sh_to_rgb_numpy(cloud["shs"], cloud["means3D"], cam.campos, cloud["sh_degree"])
        leaves["colors_precomp"] = t(cp, True)
    if cov_mode == "scale_rot":
        leaves["scales"], leaves["rotations"] = t(cloud["scales"], True), t(cloud["rotations"], True)
    else:
        leaves["cov3D_precomp"] = t(cloud["cov3D_precomp"], True)
    color, radii, aux = splat(viewmatrix=t(cam.viewmatrix), projmatrix=t(cam.projmatrix), campos=t(cam.campos),
                              bg=t(cam.bg), tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, image_height=cam.image_height,
                              image_width=cam.image_width, sh_degree=cloud["sh_degree"],
                              scale_modifier=cam.scale_modifier, return_aux=True, **leaves, **kw)
    res = {"color": color.detach().numpy(), "radii": radii.numpy(), **aux}
    if dL is not None:
        color.backward(torch.tensor(dL, dtype=dtype))
        for k, v in leaves.items():
            res["grad_" + k] = v.grad.numpy() if v.grad is not None else np.zeros(tuple(v.shape))
    return res
