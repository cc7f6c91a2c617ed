#!/usr/bin/env python3
"""A whole training step on the MI355X-native path, end to end, on synthetic data:

  cameras computed once (RasterCameras)  ->  8 views rendered without a per-frame host sync (SyncFreeBatch.run_views)
  ->  the trainers' L1 + SSIM loss and its gradient for all views in one call (loss.l1_ssim_value_and_grad)
  ->  per-pixel backward per view, ONE per-Gaussian backward for the batch  ->  Adam on the Gaussian parameters.

It fits a perturbed copy of a Gaussian cloud back to images rendered from the original one and prints the loss per step.
Usage:  python examples/train_views.py [--steps 30] [--gaussians 20000] [--size 320 200] [--views 8]
"""
import argparse
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def orbit_c2w(n, radius=3.0, elevation_deg=10.0):
    """n camera-to-world transforms [n,3,4] in OpenGL axes (Y up, Z back) on an orbit looking at the origin"""
    out = []
    for k in range(n):
        az, el = 2 * math.pi * k / n, math.radians(elevation_deg)
        eye = np.array([radius * math.cos(el) * math.sin(az), radius * math.sin(el), radius * math.cos(el) * math.cos(az)])
        z = eye / np.linalg.norm(eye)                       # camera looks along -z
        x = np.cross([0.0, 1.0, 0.0], z); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        out.append(np.concatenate([np.stack([x, y, z], 1), eye[:, None]], 1))
    return np.stack(out).astype(np.float32)


def run(steps=30, P=20000, W=320, H=200, V=8, seed=0, device="cuda", log=print):
    from youreditableavatar_amd import scenes
    from youreditableavatar_amd.cameras import RasterCameras
    from youreditableavatar_amd.loss import l1_ssim_value_and_grad
    from youreditableavatar_amd.multiview import FlatGradients, SyncFreeBatch
    dev = torch.device(device)
    cloud = scenes.make_cloud(P, 3, seed=seed, scale_mult=2.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    fov_x = 2 * math.atan(W / (2 * 1.1 * W)); fov_y = 2 * math.atan(H / (2 * 1.1 * W))
    cams = RasterCameras.from_camera_to_worlds(orbit_c2w(V), 0.01, 100.0, fov_x, fov_y, H, W, device=dev)
    bg = torch.zeros(3, device=dev)
    settings = [cams.settings(i, bg, 3) for i in range(V)]
    truth = {k: t(cloud[k]) for k in ("means3D", "opacities", "scales", "rotations", "shs")}
    params = {k: v.clone() for k, v in truth.items()}
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    params["shs"] = params["shs"] + 0.3 * torch.randn(params["shs"].shape, generator=g).to(dev)       # what the optimiser has to undo
    params["opacities"] = (params["opacities"] * 0.6).clamp(0.02, 0.99)
    for v in params.values():
        v.requires_grad_(True)
    order = ("means3D", "opacities", "scales", "rotations", "shs")
    grads = FlatGradients([params[k] for k in order])
    batch = SyncFreeBatch()
    with torch.no_grad():                                   # target images: the unperturbed cloud through the same path
        tg = {k: v.clone().requires_grad_(True) for k, v in truth.items()}
        FlatGradients([tg[k] for k in order])
        targets = SyncFreeBatch().run_views(settings, tg["means3D"], tg["opacities"], tg["shs"], tg["scales"], tg["rotations"],
                                            lambda im: torch.zeros_like(im)).clone()
    opt = torch.optim.Adam([{"params": [params["shs"]], "lr": 2e-2}, {"params": [params["opacities"]], "lr": 1e-2}])
    losses = []

    def upstream(images):
        out3, grad = l1_ssim_value_and_grad(images, targets, 0.2)
        losses.append(out3)                                 # device tensor: no host sync inside the step
        return grad

    for step in range(steps):
        batch.run_views(settings, params["means3D"], params["opacities"], params["shs"], params["scales"], params["rotations"], upstream, accumulate=False)
        grads.all_reduce()                                  # no-op on one GPU
        opt.step()
        with torch.no_grad():
            params["opacities"].clamp_(0.01, 0.99)
    vals = [float(x[0]) for x in losses]
    for i in range(0, len(vals), max(1, len(vals) // 10)):
        log(f"step {i:3d}  loss {vals[i]:.5f}")
    log(f"final loss {vals[-1]:.5f}  (views re-rendered after a capacity miss: {batch.rejected})")
    return vals


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--gaussians", type=int, default=20000)
    ap.add_argument("--size", type=int, nargs=2, default=[320, 200])
    ap.add_argument("--views", type=int, default=8)
    a = ap.parse_args()
    run(a.steps, a.gaussians, a.size[0], a.size[1], a.views)
