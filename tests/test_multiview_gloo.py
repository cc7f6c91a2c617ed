"""CPU, world size 2 over gloo: the view-sharded data-parallel step (youreditableavatar_amd/multiview.py)
gives every rank the gradient of the whole batch, equal to the unsharded sum of per-view gradients.
The per-view renderer is the CPU oracle wrapped in an autograd.Function (test-only)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class OracleRasterize(torch.autograd.Function):
    """Test-only: the oracle as a differentiable per-view renderer on CPU tensors."""

    @staticmethod
    def forward(ctx, means3D, opacities, scales, rotations, shs, cam, deg):
        from oracle import oracle
        kw = dict(bg=cam.bg, means3D=means3D.detach().numpy(), viewmatrix=cam.viewmatrix, projmatrix=cam.projmatrix, campos=cam.campos,
                  tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, shs=shs.detach().numpy(), scales=scales.detach().numpy(),
                  rotations=rotations.detach().numpy())
        color, radii, st = oracle.forward(opacities=opacities.detach().numpy(), image_height=cam.image_height, image_width=cam.image_width,
                                          sh_degree=deg, **kw)
        ctx.st, ctx.kw = st, kw
        return torch.from_numpy(color)

    @staticmethod
    def backward(ctx, g):
        from oracle import oracle
        r = oracle.backward(ctx.st, g.contiguous().numpy(), **ctx.kw)
        t = torch.from_numpy
        return t(r["dL_dmeans3D"]), t(r["dL_dopacity"]), t(r["dL_dscales"]), t(r["dL_drotations"]), t(r["dL_dsh"]), None, None


def _scene(stored_degree=1):
    from youreditableavatar_amd import scenes
    cloud = scenes.make_cloud(300, stored_degree, seed=4, scale_mult=6.0)
    cams = [scenes.orbit_camera(48, 32, azimuth_deg=k * 60.0) for k in range(5)]     # 5 views: uneven shards on 2 ranks
    dLs = [scenes.upstream_gradient(48, 32, seed=100 + k) for k in range(5)]
    return cloud, cams, dLs


def _params(cloud):
    return [torch.tensor(cloud[k], requires_grad=True) for k in ("means3D", "opacities", "scales", "rotations", "shs")]


def _step(rank, world, port, out_q, stored_degree=1, active_degree=None):
    from youreditableavatar_amd import multiview
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cloud, cams, dLs = _scene(stored_degree)
    params = _params(cloud)
    grads = multiview.FlatGradients(params, sh_params={4: 0})
    deg = cloud["sh_degree"] if active_degree is None else active_degree
    mine = multiview.render_batch_sharded(lambda v: OracleRasterize.apply(*params, cams[v], deg),
                                          lambda v, img: torch.from_numpy(dLs[v]), len(cams), grads, sh_degree=active_degree)
    out_q.put((rank, mine, grads.flat.clone().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.timeout(300)
def test_sharded_step_equals_unsharded_sum():
    from youreditableavatar_amd import multiview
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_step, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference: sum over all views
    cloud, cams, dLs = _scene()
    params = _params(cloud)
    grads = multiview.FlatGradients(params)
    multiview.render_batch_sharded(lambda v: OracleRasterize.apply(*params, cams[v], cloud["sh_degree"]),
                                   lambda v, img: torch.from_numpy(dLs[v]), len(cams), grads, rank=0, world_size=1)
    ref = grads.flat.numpy()
    assert res[0][1] == [0, 1, 2] and res[1][1] == [3, 4]                   # contiguous shards, sizes differ by at most one
    for _rank, _mine, flat in res:
        assert np.allclose(flat, ref, rtol=1e-5, atol=1e-9)
    assert np.array_equal(res[0][2], res[1][2])                            # every rank holds the same reduced buffer
    assert np.abs(ref).max() > 0


@pytest.mark.timeout(300)
@pytest.mark.parametrize("active", [0, 1])
def test_sharded_step_with_live_sh_rows_only_equals_unsharded_sum(active):
    """SH stored for degree 3, rendered at degree 0 / 1 (the reference's sh_levels schedule: refine_3dgs.py:165-166, paint_2dgs.py:61-63):
    the step reduces the (D + 1)^2 live coefficients only and every rank still ends with the whole batch's gradient."""
    from youreditableavatar_amd import multiview
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_step, args=(r, 2, port, q, 3, active)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    cloud, cams, dLs = _scene(3)
    params = _params(cloud)
    grads = multiview.FlatGradients(params, sh_params={4: 0})
    multiview.render_batch_sharded(lambda v: OracleRasterize.apply(*params, cams[v], active),
                                   lambda v, img: torch.from_numpy(dLs[v]), len(cams), grads, rank=0, world_size=1)
    ref = grads.flat.numpy()
    live = (active + 1) ** 2
    gsh = params[4].grad.numpy()
    assert np.abs(gsh[:, :live]).max() > 0 and np.all(gsh[:, live:] == 0)  # the premise: dead coefficients have exactly zero gradient
    for _rank, _mine, flat in res:
        assert np.allclose(flat, ref, rtol=1e-5, atol=1e-9)
    assert np.array_equal(res[0][2], res[1][2])
    assert grads.reduced_bytes(sh_degree=active) == 300 * 4 * (3 + 1 + 3 + 4 + 3 * live)
    assert grads.reduced_bytes() == 300 * 4 * (3 + 1 + 3 + 4 + 48)


def test_shard_views_partitions():
    from youreditableavatar_amd.multiview import shard_views
    for V, Wd in [(64, 8), (5, 2), (7, 8), (0, 4), (8, 8)]:
        got = [v for r in range(Wd) for v in shard_views(V, r, Wd)]
        assert got == list(range(V))
        sizes = [len(shard_views(V, r, Wd)) for r in range(Wd)]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_views(4, 4, 4)


def test_flat_gradients_are_views():
    from youreditableavatar_amd.multiview import FlatGradients
    a, b = torch.zeros(3, 2, requires_grad=True), torch.zeros(4, requires_grad=True)
    fg = FlatGradients([a, b])
    (a.sum() * 2 + (b * torch.arange(4.0)).sum()).backward()
    assert fg.flat.tolist() == [2.0] * 6 + [0.0, 1.0, 2.0, 3.0]
    assert a.grad.data_ptr() == fg.flat.data_ptr()
    assert fg.all_reduce() is None          # no process group: no-op


def _rows_step(rank, world, port, out_q, sh_degree=None, level_major=False):
    """every rank fills its flat buffer with rank-dependent values and reduces it range by range (FlatGradients.all_reduce_rows)"""
    from youreditableavatar_amd import multiview
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P = 1000
    params = [torch.zeros(P, 3, requires_grad=True), torch.zeros(P, 1, requires_grad=True), torch.zeros(P, 16, 3, requires_grad=True)]
    if sh_degree is not None:                            # the model's own two SH parameters as well (tetgs_model.py:234-239)
        params += [torch.zeros(P, 1, 3, requires_grad=True), torch.zeros(P, 15, 3, requires_grad=True)]
        params += [torch.zeros(P, 4, 3, requires_grad=True)]   # looks like SH by its shape, is not named as SH: reduced whole (ADVICE of round 5)
    fg = multiview.FlatGradients(params, sh_params=None if sh_degree is None else {2: 0, 3: 0, 4: 1}, level_major=level_major)
    g = torch.Generator().manual_seed(7 + rank)
    for p in params:                                     # (through the .grad views: a level-major buffer has padding between its planes)
        p.grad.copy_(torch.randn(p.shape, generator=g))
    if sh_degree is not None:                            # coefficients above the active degree have zero gradient on every rank
        live = (sh_degree + 1) ** 2
        params[2].grad[:, live:] = 0
        params[4].grad[:, live - 1:] = 0
    cat = lambda: torch.cat([p.grad.contiguous().flatten() for p in params])
    mine = cat()
    works = []
    if level_major == "whole":                           # one call for all Gaussians: whole planes, padding included
        works += fg.all_reduce_rows(0, P, sh_degree=sh_degree)
    else:
        for first in range(0, P, 256):                   # ranges of 256 Gaussians, the last one ragged
            works += fg.all_reduce_rows(first, min(256, P - first), sh_degree=sh_degree)
    for w in works:
        w.wait()
    if level_major:                                      # a level-major buffer hands slices of `flat` to the collective: nothing is staged
        assert not fg.__dict__.get("_stage") and all(not isinstance(w, multiview._PackedReduce) for w in works)
        assert params[2].grad.stride() == (3, fg.regions[2][2], 1) and fg.regions[2][2] % 64 == 0 and fg.regions[2][0] % 64 == 0
    out_q.put((rank, mine.numpy(), cat().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_all_reduce_by_gaussian_ranges_equals_one_all_reduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rows_step, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    total = res[0][1] + res[1][1]
    for _rank, _mine, reduced in res:
        assert np.array_equal(reduced, total)           # every element reduced exactly once


@pytest.mark.timeout(300)
@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_all_reduce_of_live_sh_rows_by_ranges(deg):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rows_step, args=(r, 2, port, q, deg)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    total = res[0][1] + res[1][1]
    for _rank, _mine, reduced in res:
        assert np.array_equal(reduced, total)           # live coefficients summed exactly once, dead ones still zero


@pytest.mark.timeout(300)
@pytest.mark.parametrize("deg,mode", [(0, True), (1, True), (2, "whole"), (3, True), (0, "whole")])
def test_all_reduce_of_live_sh_rows_level_major(deg, mode):
    """FlatGradients(level_major=True): the SH gradients are stored coefficient plane by coefficient plane, the live coefficients of a step are
    the leading planes, and all_reduce_rows hands slices of the flat buffer to the collective -- no staging copy (round 6)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rows_step, args=(r, 2, port, q, deg, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    total = res[0][1] + res[1][1]
    for _rank, _mine, reduced in res:
        assert np.array_equal(reduced, total)


def test_level_major_gradients_are_views_of_planes():
    from youreditableavatar_amd.multiview import FlatGradients
    P = 70
    a, sh = torch.zeros(P, 3, requires_grad=True), torch.zeros(P, 16, 3, requires_grad=True)
    fg = FlatGradients([a, sh], sh_params={1: 0}, level_major=True)
    off, n, stride = fg.regions[1]
    assert stride == 256 and off == 256 and n == 16 * 256 and fg.flat.numel() == off + n          # 3 P = 210 -> planes of 256 floats, 256-byte aligned
    w = torch.arange(P * 16 * 3, dtype=torch.float32).view(P, 16, 3)
    (sh * w).sum().backward()                            # autograd accumulates into the strided .grad in place
    assert sh.grad.data_ptr() == fg.flat.data_ptr() + 4 * off and torch.equal(sh.grad, w)
    assert torch.equal(fg.flat[off + 5 * stride: off + 5 * stride + 3 * P].view(P, 3), w[:, 5, :])     # plane 5 = coefficient 5 of every Gaussian
    sl = fg.row_slices(10, 20)
    assert len(sl) == 1 + 16 and torch.equal(sl[1 + 5].view(20, 3), w[10:30, 5, :])
    torch.optim.Adam([a, sh], lr=1e-2).step()            # an optimizer reads the strided gradient like any other
    assert float(sh.detach().abs().max()) > 0


def test_sh_live_rule():
    """SH parameters are NAMED by the caller (sh_params: index -> first coefficient); nothing is inferred from shapes"""
    from youreditableavatar_amd.multiview import FlatGradients as F
    z = lambda *sh: torch.zeros(*sh, requires_grad=True)
    fg = F([z(10, 16, 3), z(10, 1, 3), z(10, 15, 3), z(10, 3, 3), z(10, 4, 3), z(10, 8, 3), z(10, 3)], sh_params={0: 0, 1: 0, 2: 1})
    assert [fg._sh_live(0, d) for d in (0, 1, 2, 3)] == [1, 4, 9, None]
    assert [fg._sh_live(1, d) for d in (0, 1, 2, 3)] == [None, None, None, None]          # the dc tensor: its one coefficient is always live
    assert [fg._sh_live(2, d) for d in (0, 1, 2, 3)] == [0, 3, 8, None]
    for i in (3, 4, 5, 6):                               # [P,3,3] / [P,4,3] / [P,8,3] parameters that are not SH: every entry is live at every degree
        assert all(fg._sh_live(i, d) is None for d in (0, 1, 2, 3))
    assert fg._sh_live(0, None) is None
    assert fg.reduced_bytes(sh_degree=0) == 10 * 4 * (3 + 3 + 0 + 9 + 12 + 24 + 3)
    with pytest.raises(ValueError):
        F([z(10, 16, 3)])._sh_live(0, 0)                 # an active degree without named SH parameters is an error, not a guess
    with pytest.raises(ValueError):
        F([z(10, 3)], sh_params={0: 0})                  # not a [P, M, 3] tensor


def test_row_slices_cover_the_flat_buffer_once():
    from youreditableavatar_amd.multiview import FlatGradients
    P = 700
    params = [torch.zeros(P, 3, requires_grad=True), torch.zeros(P, 1, requires_grad=True), torch.zeros(P, 4, 3, requires_grad=True)]
    fg = FlatGradients(params)
    for first in range(0, P, 256):
        for t in fg.row_slices(first, min(256, P - first)):
            t += 1
    assert torch.all(fg.flat == 1)
    assert fg.all_reduce_rows(0, P) == []       # no process group: nothing to wait for
