"""simple-knn distCUDA2 ("next" row 4): oracle self-consistency on CPU, HIP parity on the GPU."""
import numpy as np
import pytest
import torch

from oracle import knn_oracle


def clouds():
    rng = np.random.default_rng(7)
    out = {}
    out["uniform_5000"] = rng.uniform(-1, 1, (5000, 3)).astype(np.float32)
    c = rng.normal(0, 1, (40, 3))
    out["clustered_6000"] = (c[rng.integers(0, 40, 6000)] + rng.normal(0, 0.01, (6000, 3))).astype(np.float32)
    d = rng.uniform(-1, 1, (3000, 3)).astype(np.float32)
    d[::3] = d[1::3]                                                      # exact duplicates
    out["duplicates_3000"] = d
    f = rng.uniform(-1, 1, (2049, 3)).astype(np.float32)
    f[:, 2] = 0.25                                                        # zero extent on one axis
    out["flat_2049"] = f
    return out


@pytest.mark.parametrize("name", list(clouds()))
def test_oracle_bruteforce_matches_kdtree(name):
    pts = clouds()[name]
    a, b = knn_oracle.dist2_bruteforce(pts), knn_oracle.dist2_kdtree(pts)
    np.testing.assert_allclose(a, b, rtol=1e-6, atol=0)


def test_oracle_tiny():
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3], [5, 5, 5]], np.float32)
    r = knn_oracle.dist2_bruteforce(pts)
    assert r[0] == np.float32((1 + 4 + 9) / 3)
    # fewer than 3 neighbours: FLT_MAX terms (simple_knn.cu:155) -- one absorbs the sum, two overflow
    assert (knn_oracle.dist2_bruteforce(pts[:3]) == knn_oracle.FLT_MAX / np.float32(3)).all()
    assert np.isinf(knn_oracle.dist2_bruteforce(pts[:2])).all()


def test_cpu_tensor_is_refused():
    from simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros(8, 3))


def _hip(pts):
    from simple_knn._C import distCUDA2
    return distCUDA2(torch.from_numpy(pts).cuda()).cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(clouds()))
def test_gpu_matches_oracle(name):
    pts = clouds()[name]
    got, want = _hip(pts), knn_oracle.dist2_bruteforce(pts)
    # identical neighbour sets; the only freedom is fma contraction inside one squared distance
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("P", [1, 2, 3, 4, 5, 7, 1023, 1024, 1025, 8191, 8192, 8193, 20000])
def test_gpu_sizes(P):
    pts = np.random.default_rng(P).normal(0, 1, (P, 3)).astype(np.float32)
    got, want = _hip(pts), knn_oracle.dist2_bruteforce(pts)
    if P < 4:
        np.testing.assert_array_equal(got, want)
    else:
        np.testing.assert_allclose(got, want, rtol=2e-6, atol=0)


@pytest.mark.gpu
def test_gpu_empty():
    from simple_knn._C import distCUDA2
    assert distCUDA2(torch.zeros(0, 3, device="cuda")).shape == (0,)


@pytest.mark.gpu
def test_gpu_full_size_against_kdtree():
    """BASELINE config-3 cloud size (500k points): exact against the k-d tree oracle."""
    from youreditableavatar_amd import scenes
    pts = scenes.make_cloud(500_000, sh_degree=0, seed=3)["means3D"]
    pts = np.ascontiguousarray(pts, np.float32)
    got, want = _hip(pts), knn_oracle.dist2_kdtree(pts)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=0)


def test_knn_oracle_self_query():
    pts = clouds()["uniform_5000"][:600]
    d, i = knn_oracle.knn_self_bruteforce(pts, 4)
    assert np.all(d[:, 0] == 0) and np.all(i[:, 0] == np.arange(600)) and np.all(np.diff(d, axis=1) >= 0)
    from scipy.spatial import cKDTree
    dk, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    np.testing.assert_allclose(d, (dk ** 2).astype(np.float32), rtol=1e-5, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("name,K", [("uniform_5000", 4), ("clustered_6000", 16), ("duplicates_3000", 4), ("flat_2049", 9), ("uniform_5000", 32)])
def test_gpu_knn_points_self(name, K):
    """knn_points(p, p, K) of the reference's scale initialisation (K = 4) and neighbour tracking (K = 16)"""
    from youreditableavatar_amd.knn import knn_points_self
    pts = clouds()[name]
    want_d, _ = knn_oracle.knn_self_bruteforce(pts, K)
    out = knn_points_self(torch.from_numpy(pts).cuda()[None], K)
    d, i = out.dists[0].cpu().numpy(), out.idx[0].cpu().numpy()
    assert out.dists.shape == (1, len(pts), K) and out.idx.dtype == torch.int64
    np.testing.assert_allclose(d, want_d, rtol=2e-6, atol=0)
    # the indices are neighbours at exactly those distances (ties may be ordered differently)
    diff = pts[i] - pts[:, None, :]
    np.testing.assert_allclose((diff * diff).sum(-1), d, rtol=2e-6, atol=1e-12)
    assert np.all(d[:, 0] == 0)
    for row in i[:: max(1, len(pts) // 50)]:
        assert len(set(row.tolist())) == K


@pytest.mark.gpu
def test_gpu_knn_fewer_points_than_k_and_scale_init():
    from youreditableavatar_amd.knn import knn_points_self
    pts = torch.tensor([[0.0, 0, 0], [1, 0, 0], [0, 3, 0]], device="cuda")
    out = knn_points_self(pts, 4)
    assert out.idx.shape == (3, 4) and torch.all(out.idx[:, 3] == -1) and torch.all(out.dists[:, 3] > 1e38)
    assert out.dists[0].tolist()[:3] == [0.0, 1.0, 9.0] and out.idx[0].tolist()[:3] == [0, 1, 2]
    # the reference's scale initialisation (tetgs_model.py:36-46) on top of it
    cloud = torch.from_numpy(clouds()["uniform_5000"]).cuda()
    knn = knn_points_self(cloud[None], K=4)
    radiuses = torch.sqrt(knn.dists[..., 1:]).mean(-1, keepdim=True).clamp_min(0.0000001)
    want_d, _ = knn_oracle.knn_self_bruteforce(cloud.cpu().numpy(), 4)
    np.testing.assert_allclose(radiuses[0, :, 0].cpu().numpy(), np.sqrt(want_d[:, 1:]).mean(-1), rtol=1e-5)
