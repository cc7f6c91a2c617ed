"""The trainers' photometric loss ("next" row 2): oracle pinned to the reference's own outputs on CPU, HIP parity on the GPU."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_ref
from tests import util

FX = os.path.join(util.GOLDEN_DIR, "ref_loss_fixture.npz")
CASES = "abcd"


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_reference(name):
    """oracle/loss_ref.py against outputs + autograd gradients recorded from the reference's loss_utils.py."""
    fx = np.load(FX)
    p, g = torch.tensor(fx[f"{name}_pred"], requires_grad=True), torch.tensor(fx[f"{name}_gt"])
    for fn, key in ((loss_ref.ssim, "ssim"), (loss_ref.l1_loss, "l1"), (loss_ref.l1_ssim_loss, "loss")):
        v = fn(p, g)
        (dv,) = torch.autograd.grad(v, p)
        np.testing.assert_allclose(v.detach().numpy(), fx[f"{name}_{key}"], rtol=1e-6)
        assert util.rel_l2(dv.numpy(), fx[f"{name}_d{key}"]) <= 1e-6
    np.testing.assert_allclose(loss_ref.gaussian_window().numpy(), fx["window"], rtol=0, atol=0)


def test_cpu_tensor_is_refused():
    from youreditableavatar_amd import loss
    with pytest.raises(RuntimeError):
        loss.l1_ssim_loss(torch.zeros(3, 8, 8), torch.zeros(3, 8, 8))


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_matches_reference_fixture(name):
    from youreditableavatar_amd import loss
    fx = np.load(FX)
    g = torch.tensor(fx[f"{name}_gt"]).cuda()
    for fn, key in ((loss.ssim, "ssim"), (loss.l1_loss, "l1"), (loss.l1_ssim_loss, "loss")):
        p = torch.tensor(fx[f"{name}_pred"]).cuda().requires_grad_(True)
        v = fn(p, g)
        v.backward()
        np.testing.assert_allclose(v.item(), fx[f"{name}_{key}"], rtol=2e-6)
        assert util.rel_l2(p.grad.cpu().numpy(), fx[f"{name}_d{key}"]) <= 1e-5, key     # fp32 tolerance, stated


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 1, 1), (3, 5, 200), (1, 33, 17), (2, 3, 40, 56), (3, 1080, 1920)])
def test_gpu_matches_oracle_fp64(shape):
    """other sizes (ragged tiles, batch, the full 1080p frame) against the oracle evaluated in float64"""
    from youreditableavatar_amd import loss
    rng = np.random.default_rng(sum(shape))
    gt = rng.uniform(0, 1, shape).astype(np.float32)
    pred = np.clip(gt + rng.normal(0, 0.1, shape), 0, 1).astype(np.float32)
    p64 = torch.tensor(pred, dtype=torch.float64, requires_grad=True)
    want = loss_ref.l1_ssim_loss(p64, torch.tensor(gt, dtype=torch.float64), 0.2)
    (dwant,) = torch.autograd.grad(want, p64)
    p = torch.tensor(pred).cuda().requires_grad_(True)
    got = loss.l1_ssim_loss(p, torch.tensor(gt).cuda(), 0.2)
    (3.0 * got).backward()
    np.testing.assert_allclose(got.item(), want.item(), rtol=1e-5)
    assert util.rel_l2(p.grad.cpu().numpy() / 3.0, dwant.numpy()) <= 1e-5


@pytest.mark.gpu
def test_gpu_value_and_grad_is_reproducible():
    from youreditableavatar_amd import loss
    rng = np.random.default_rng(5)
    a, b = torch.tensor(rng.uniform(0, 1, (3, 200, 300)).astype(np.float32)).cuda(), torch.tensor(rng.uniform(0, 1, (3, 200, 300)).astype(np.float32)).cuda()
    o1, g1 = loss.l1_ssim_value_and_grad(a, b)
    o2, g2 = loss.l1_ssim_value_and_grad(a, b)
    assert torch.equal(o1, o2) and torch.equal(g1, g2)
    o3, g3 = loss.l1_ssim_value_and_grad(a, a)
    assert abs(o3[1].item() - 1.0) < 1e-6 and o3[2].item() == 0.0 and abs(o3[0].item()) < 1e-6
