#!/usr/bin/env python3
"""BUILD CONTAINER ONLY.  Imports the reference's Edit_core/utils/loss_utils.py from /root/reference and records
inputs, outputs and autograd gradients of its l1_loss / ssim and of the 'l1+dssim' loss its trainers build from them
(refine.py:245-247) into tests/golden/ref_loss_fixture.npz.  A fixture is data; no reference source is copied."""
import os
import sys

import numpy as np
import torch

REF = "/root/reference/Edit_core"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_loss_fixture.npz")


def main():
    sys.path.insert(0, REF)
    from utils import loss_utils as lu
    rng = np.random.Generator(np.random.PCG64(11))
    out = {}
    cases = {"a": (3, 37, 53), "b": (3, 16, 16), "c": (1, 9, 70), "d": (3, 64, 96)}
    for name, (C, H, W) in cases.items():
        gt = rng.uniform(0, 1, (C, H, W)).astype(np.float32)
        # smooth-ish prediction near the target plus noise, some exact matches (sign(0) in the L1 gradient)
        pred = np.clip(gt + rng.normal(0, 0.15, (C, H, W)), 0, 1).astype(np.float32)
        pred[:, ::5, ::7] = gt[:, ::5, ::7]
        p = torch.tensor(pred, requires_grad=True)
        g = torch.tensor(gt)
        s = lu.ssim(p, g)
        (gs,) = torch.autograd.grad(s, p)
        l1 = lu.l1_loss(p, g)
        (gl1,) = torch.autograd.grad(l1, p)
        loss = (1.0 - 0.2) * lu.l1_loss(p, g) + 0.2 * (1.0 - lu.ssim(p, g))
        (gl,) = torch.autograd.grad(loss, p)
        out[f"{name}_pred"], out[f"{name}_gt"] = pred, gt
        out[f"{name}_ssim"], out[f"{name}_dssim"] = s.detach().numpy(), gs.numpy()
        out[f"{name}_l1"], out[f"{name}_dl1"] = l1.detach().numpy(), gl1.numpy()
        out[f"{name}_loss"], out[f"{name}_dloss"] = loss.detach().numpy(), gl.numpy()
    out["window"] = lu.gaussian(11, 1.5).numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items() if k.endswith("_pred")})


if __name__ == "__main__":
    main()
