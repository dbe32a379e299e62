"""CPU checks of the Chamfer oracle (oracle/chamfer_oracle.c) against the float64 definition.

The reference holds no test or fixture for lib/utils/chamfer3D, and its kernels are CUDA-only, so
the oracle is pinned against the definition its kernels implement (chamfer3D.cu:11-133, 154-173): squared distance
to the nearest neighbour, first minimum wins, and the analytic gradient of that."""
import numpy as np
import pytest

from oracle import chamfer as och


def clouds(B, N, M, seed, scale=20.0):
    r = np.random.default_rng(seed)
    a = (r.standard_normal((B, N, 3)) * scale).astype(np.float32)
    b = (a[:, r.integers(0, N, M)] + r.standard_normal((B, M, 3)) * 0.3).astype(np.float32) if N > 0 else None
    return a, b


def brute64(a, b):
    d = ((a[:, :, None, :].astype(np.float64) - b[:, None, :, :].astype(np.float64)) ** 2).sum(-1)    # (B,N,M)
    return d.min(2), d.argmin(2), d.min(1), d.argmin(1), d


@pytest.mark.parametrize("B,N,M", [(1, 300, 257), (2, 64, 1), (1, 1, 64), (3, 33, 129)])
def test_forward_matches_float64_definition(B, N, M):
    a, b = clouds(B, N, M, 7 + N)
    d1, d2, i1, i2 = och.chamfer_forward(a, b)
    e1, j1, e2, j2, d = brute64(a, b)
    np.testing.assert_allclose(d1, e1, rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(d2, e2, rtol=2e-6, atol=1e-9)
    # the chosen neighbour attains the float64 minimum up to float32 rounding of the pair distance
    np.testing.assert_allclose(np.take_along_axis(d, i1[:, :, None].astype(np.int64), 2)[..., 0], e1, rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(np.take_along_axis(d, i2[:, None, :].astype(np.int64), 1)[:, 0], e2, rtol=2e-6, atol=1e-9)
    assert (i1 == j1).mean() > 0.99 and (i2 == j2).mean() > 0.99
    assert d1.dtype == np.float32 and i1.dtype == np.int32


def test_first_minimum_wins_on_exact_ties():
    """chamfer3D.cu:35,124 -- strict `<`: among equal distances the lowest index is kept."""
    r = np.random.default_rng(3)
    b = r.integers(-4, 5, (1, 600, 3)).astype(np.float32)          # lattice points: many exact duplicates / ties
    a = r.integers(-4, 5, (1, 500, 3)).astype(np.float32) + np.float32(0.5)
    d1, d2, i1, i2 = och.chamfer_forward(a, b)
    dd = ((a[0, :, None] - b[0, None]) ** 2).sum(-1)               # exact in float32 (small half-integers)
    assert np.array_equal(d1[0], dd.min(1)) and np.array_equal(i1[0], dd.argmin(1))       # argmin = first occurrence
    assert np.array_equal(d2[0], dd.min(0)) and np.array_equal(i2[0], dd.argmin(0))
    same = och.chamfer_forward(b, b)
    assert np.all(same[0] == 0) and np.array_equal(same[2][0], np.array([np.flatnonzero((b[0] == p).all(1))[0] for p in b[0]]))


def test_contraction_variant_moves_distances_by_an_ulp_at_most():
    a, b = clouds(1, 400, 500, 11)
    base = och.chamfer_forward(a, b, 0)
    for v in (1, 2):
        alt = och.chamfer_forward(a, b, v)
        np.testing.assert_allclose(alt[0], base[0], rtol=2.5e-7, atol=0)
        assert (alt[2] == base[2]).mean() > 0.995


def test_backward_is_the_gradient_of_the_weighted_distance_sum():
    a, b = clouds(2, 40, 50, 5, scale=3.0)
    r = np.random.default_rng(1)
    w1 = r.uniform(0.5, 1.5, (2, 40)).astype(np.float32); w2 = r.uniform(0.5, 1.5, (2, 50)).astype(np.float32)
    _, _, i1, i2 = och.chamfer_forward(a, b)
    ga, gb = och.chamfer_backward(a, b, w1, w2, i1, i2, "f64")
    ga32, gb32 = och.chamfer_backward(a, b, w1, w2, i1, i2, "f32")
    np.testing.assert_allclose(ga32, ga, rtol=1e-5, atol=1e-5); np.testing.assert_allclose(gb32, gb, rtol=1e-5, atol=1e-5)

    def loss(a_, b_):
        e1, _, e2, _, _ = brute64(a_, b_)
        return (e1 * w1).sum() + (e2 * w2).sum()

    h = 1e-4
    for (arr, g, which) in ((a, ga, 0), (b, gb, 1)):
        for (bi, pi, ax) in ((0, 3, 0), (1, 17, 2), (0, 39, 1)):
            p = arr.astype(np.float64).copy(); m = p.copy()
            p[bi, pi, ax] += h; m[bi, pi, ax] -= h
            a64, b64 = a.astype(np.float64), b.astype(np.float64)
            fd = (loss(p, b64) - loss(m, b64)) / (2 * h) if which == 0 else (loss(a64, p) - loss(a64, m)) / (2 * h)
            assert abs(fd - g[bi, pi, ax]) <= 1e-4 * max(1.0, abs(fd)), (which, bi, pi, ax, fd, g[bi, pi, ax])
