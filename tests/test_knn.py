"""simple-knn (`distCUDA2`): oracle vs the float64 definition on CPU; HIP vs oracle bit-exact on the GPU."""
import numpy as np
import pytest
import torch

from oracle import chamfer as och


def cloud(P, seed, scale=10.0):
    return (np.random.default_rng(seed).standard_normal((P, 3)) * scale).astype(np.float32)


def definition64(p):
    d = ((p[:, None].astype(np.float64) - p[None].astype(np.float64)) ** 2).sum(-1)
    np.fill_diagonal(d, np.inf)
    return np.sort(d, 1)[:, :3].mean(1)


@pytest.mark.parametrize("P", [4, 5, 257, 1500])
def test_oracle_matches_float64_definition(P):
    p = cloud(P, P)
    np.testing.assert_allclose(och.knn_mean_dist2(p), definition64(p), rtol=3e-6)


def test_oracle_small_and_duplicate_clouds():
    """Fewer than 4 points: missing neighbours stay FLT_MAX (simple_knn.cu:154): one missing -> FLT_MAX/3, two or
    three missing -> the float32 sum overflows to inf; duplicates are neighbours at distance 0 (self-skip by position, :170)."""
    assert np.isinf(och.knn_mean_dist2(cloud(2, 1))).all() and np.isinf(och.knn_mean_dist2(cloud(1, 1))).all()
    assert np.array_equal(och.knn_mean_dist2(cloud(3, 1)), np.full(3, np.finfo(np.float32).max / np.float32(3.0), np.float32))
    p = np.repeat(cloud(50, 2), 4, 0)
    assert np.array_equal(och.knn_mean_dist2(p), np.zeros(200, np.float32))
    lattice = np.random.default_rng(0).integers(-3, 4, (400, 3)).astype(np.float32)
    d = ((lattice[:, None] - lattice[None]) ** 2).sum(-1); np.fill_diagonal(d, np.inf)
    s = np.sort(d, 1)[:, :3].astype(np.float32)
    assert np.array_equal(och.knn_mean_dist2(lattice), (s[:, 0] + s[:, 1] + s[:, 2]) / np.float32(3.0))


@pytest.mark.gpu
@pytest.mark.parametrize("P", [1, 2, 3, 4, 9, 64, 65, 1000, 30000])
def test_hip_bit_exact_random(P):
    from simple_knn._C import distCUDA2
    p = cloud(P, 10 + P)
    got = distCUDA2(torch.as_tensor(p, device="cuda:0")).cpu().numpy()
    want = och.knn_mean_dist2(p)
    assert np.array_equal(got.view(np.int32), want.view(np.int32)), f"{(got != want).sum()} of {P} differ"


@pytest.mark.gpu
def test_hip_bit_exact_lidar_init_cloud_and_duplicates():
    """A LiDAR-shaped initial cloud (what gaussian_model.py:167 feeds), with exact duplicates mixed in."""
    from lidar_rt_amd import scenes
    from simple_knn._C import distCUDA2
    r = np.random.default_rng(5)
    _, d = scenes.kitti_rays(32, 1024)
    pts = (d.reshape(-1, 3) * (5.0 + 40.0 * r.random(d.shape[0] * d.shape[1]) ** 2)[:, None]).astype(np.float32)
    pts = np.concatenate([pts, pts[r.integers(0, len(pts), 3000)]], 0)
    got = distCUDA2(torch.as_tensor(pts, device="cuda:0")).cpu().numpy()
    want = och.knn_mean_dist2(pts)
    assert np.array_equal(got.view(np.int32), want.view(np.int32)), f"{(got != want).sum()} differ"
    # gaussian_model.py:167-168 use: clamp_min(dist2, 1e-7) -> log(sqrt()) must be finite
    assert np.isfinite(np.log(np.sqrt(np.maximum(got, 1e-7)))).all()


@pytest.mark.gpu
def test_hip_non_contiguous_and_dtype():
    from simple_knn._C import distCUDA2
    p = cloud(500, 3)
    t = torch.as_tensor(np.concatenate([p, p], 1), device="cuda:0")[:, :3]          # non-contiguous view: made contiguous like the reference
    assert np.array_equal(distCUDA2(t).cpu().numpy(), och.knn_mean_dist2(p))
    with pytest.raises(RuntimeError, match="float32"):
        distCUDA2(torch.zeros(4, 3, device="cuda:0", dtype=torch.float64))
    assert distCUDA2(torch.zeros(0, 3, device="cuda:0")).shape == (0,)
