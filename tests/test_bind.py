"""The fused binding-side op (youreditableavatar_amd/bindings.gaussian_bind, csrc/tgs_bind.hip) against the torch restatement of the
reference's model properties (oracle/bind_ref.py; tetgs_model.py:252-286)."""
import numpy as np
import pytest
import torch

from tests import util


def _inputs(P, seed=3):
    rng = np.random.default_rng(seed)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)
    d = f(P, 1) * 3.0
    s = np.log(np.abs(f(P, 3)) * 0.01 + 1e-8).astype(np.float32)     # log of small positive scales; tetgs_edit_2d.py:203 stores log(1e-8) for the flat axis
    s[::7, 0] = np.log(1e-8)
    q = f(P, 4)
    if P > 5:
        q[5] *= 1e-20                                                # a (nearly) degenerate quaternion: the max(|x|, 1e-12) branch
    o, n = f(P, 3), f(P, 3)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    dl = f(P, 1) * 0.01
    return d, s, q, o, n, dl


def test_bind_oracle_matches_its_definition():
    """CPU: the restatement is the four torch calls of the reference's properties."""
    from oracle import bind_ref
    d, s, q, o, n, dl = (torch.tensor(a, dtype=torch.float64) for a in _inputs(257))
    op, sc, qu, pt = bind_ref.bind(d, s, q, o, n, dl)
    assert torch.allclose(op, 1 / (1 + torch.exp(-d))) and torch.allclose(sc, torch.exp(s))
    assert torch.allclose(qu.norm(dim=1)[torch.arange(257) != 5], torch.ones(256, dtype=torch.float64))
    assert torch.allclose(pt, o + n * dl)
    assert bind_ref.bind(d)[1:] == (None, None, None)


@pytest.mark.gpu
@pytest.mark.parametrize("P", [1, 255, 10_000])
def test_gaussian_bind_matches_the_reference_properties(P, gpu_device):
    from oracle import bind_ref
    from youreditableavatar_amd.bindings import gaussian_bind
    arrs = _inputs(P)
    dev = [torch.tensor(a, device=gpu_device, requires_grad=(i in (0, 1, 2, 5))) for i, a in enumerate(arrs)]
    ref = [torch.tensor(a, dtype=torch.float64, requires_grad=(i in (0, 1, 2, 5))) for i, a in enumerate(arrs)]
    outs = gaussian_bind(*dev)
    want = bind_ref.bind(*ref)
    rng = np.random.default_rng(9)
    gs = [rng.standard_normal(tuple(w.shape)).astype(np.float32) for w in want]
    torch.autograd.backward(list(outs), [torch.tensor(g, device=gpu_device) for g in gs])
    torch.autograd.backward(list(want), [torch.tensor(g, dtype=torch.float64) for g in gs])
    for name, a, b in zip(("strengths", "scaling", "quaternions", "points"), outs, want):
        assert tuple(a.shape) == tuple(b.shape)
        assert util.rel_l2(a.detach().cpu().numpy(), b.detach().numpy()) <= 2e-7, name
    for i, name in ((0, "all_densities"), (1, "_scales"), (2, "_quaternions"), (5, "_points")):
        if P > 5 or i != 2:
            keep = np.ones(P, bool)
            if i == 2 and P > 5:
                keep[5] = False      # the degenerate quaternion: 1/|x| ~ 1e20 amplifies fp32 rounding of the projection; checked for finiteness only
            assert util.rel_l2(dev[i].grad.cpu().numpy()[keep], ref[i].grad.numpy()[keep]) <= 2e-6, name
            assert np.isfinite(dev[i].grad.cpu().numpy()).all(), name
    assert dev[3].grad is None and dev[4].grad is None               # buffers of the reference's models
    # a subset of the groups, and no CPU path
    op, sc, qu, pt = gaussian_bind(dev[0].detach(), None, dev[2].detach())
    assert sc is None and pt is None and torch.equal(op, outs[0].detach()) and torch.equal(qu, outs[2].detach())
    with pytest.raises(RuntimeError, match="no CPU path"):
        gaussian_bind(torch.zeros(4, 1))
    with pytest.raises(ValueError):
        gaussian_bind()
