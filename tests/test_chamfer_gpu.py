"""GPU parity of the Chamfer operator (through the C ABI) against the CPU oracle: BIT-EXACT distances and indices,
brute-force and tree modes, plus the autograd contract of dist_chamfer_3D.py:31-82."""
import numpy as np
import pytest
import torch

from lidar_rt_amd import scenes
from lidar_rt_amd.chamfer3D import _C as chamfer_3D
from lidar_rt_amd.chamfer3D import chamfer_3DDist
from oracle import chamfer as och

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
from lidar_rt_amd import _capi
CH_MODES = [0, 1] + ([3] if _capi.has_legacy() else [])      # 3 = the lane-per-query kernel kc_query: cross-check library only


def lidar_clouds(H, W, seed, drop=0.15):
    """Prediction / ground-truth point pairs shaped like train.py:197-205: same rays, different ranges, masked."""
    r = np.random.default_rng(seed)
    _, d = scenes.kitti_rays(H, W)
    d = d.reshape(-1, 3)
    depth = (8.0 + 30.0 * r.random(d.shape[0]) ** 2).astype(np.float32)
    gt = d * depth[:, None]
    pred = d * (depth * (1 + 0.02 * r.standard_normal(d.shape[0])).astype(np.float32))[:, None]
    keep = r.random(d.shape[0]) > drop
    return pred[keep][None].astype(np.float32), gt[keep][None].astype(np.float32)


def run_hip(a, b, mode):
    chamfer_3D.set_option("mode", mode)
    ta, tb = torch.as_tensor(a, device=DEV), torch.as_tensor(b, device=DEV)
    B, N, M = a.shape[0], a.shape[1], b.shape[1]
    d1 = torch.full((B, N), -1.0, device=DEV); d2 = torch.full((B, M), -1.0, device=DEV)
    i1 = torch.full((B, N), -7, device=DEV, dtype=torch.int32); i2 = torch.full((B, M), -7, device=DEV, dtype=torch.int32)
    assert chamfer_3D.forward(ta, tb, d1, d2, i1, i2) == 1
    torch.cuda.synchronize()
    chamfer_3D.set_option("mode", 2)
    return d1.cpu().numpy(), d2.cpu().numpy(), i1.cpu().numpy(), i2.cpu().numpy()


def assert_bit_exact(got, want):
    for g, w, name in zip(got, want, ("dist1", "dist2", "idx1", "idx2")):
        assert g.dtype == w.dtype and g.shape == w.shape, name
        assert np.array_equal(g.view(np.int32) if g.dtype == np.float32 else g, w.view(np.int32) if w.dtype == np.float32 else w), \
            f"{name}: {(g != w).sum()} of {g.size} differ"


def rand_clouds(B, N, M, seed, scale=20.0):
    r = np.random.default_rng(seed)
    a = (r.standard_normal((B, N, 3)) * scale).astype(np.float32)
    b = (r.standard_normal((B, M, 3)) * scale).astype(np.float32)
    return a, b


@pytest.mark.parametrize("B,N,M", [(1, 1, 1), (1, 1, 700), (1, 700, 1), (1, 7, 9), (1, 64, 65), (2, 513, 4097),
                                   (3, 1000, 777), (1, 20000, 15000)])
def test_forward_bit_exact_random(B, N, M):
    a, b = rand_clouds(B, N, M, 100 + N + M)
    want = och.chamfer_forward(a, b)
    for mode in CH_MODES:
        assert_bit_exact(run_hip(a, b, mode), want)


@pytest.mark.parametrize("mode", CH_MODES)
def test_forward_bit_exact_lidar_frame(mode):
    a, b = lidar_clouds(32, 1024, 5)
    assert_bit_exact(run_hip(a, b, mode), och.chamfer_forward(a, b))


@pytest.mark.parametrize("mode", CH_MODES)
def test_exact_ties_keep_the_first_index(mode):
    r = np.random.default_rng(3)
    b = r.integers(-6, 7, (1, 5000, 3)).astype(np.float32)         # lattice: thousands of exact duplicates and ties
    a = r.integers(-6, 7, (1, 4000, 3)).astype(np.float32) + np.float32(0.5)
    assert_bit_exact(run_hip(a, b, mode), och.chamfer_forward(a, b))
    got = run_hip(b, b, mode)
    assert_bit_exact(got, och.chamfer_forward(b, b))
    assert np.all(got[0] == 0) and np.all(got[2][0] <= np.arange(b.shape[1]))


def test_degenerate_clouds_tree_mode():
    """All points identical / collinear / one far outlier: the Morton grid degenerates, results must not."""
    r = np.random.default_rng(9)
    same = np.tile(np.array([[1.5, -2.0, 0.25]], np.float32), (1, 300, 1)).reshape(1, 300, 3)
    line = np.zeros((1, 400, 3), np.float32); line[0, :, 0] = r.standard_normal(400)
    far = (r.standard_normal((1, 500, 3))).astype(np.float32); far[0, 17] = (1e6, -1e6, 1e6)
    for a, b in ((same, same), (same, line), (line, far), (far, far[:, ::-1].copy())):
        for mode in CH_MODES[1:]:
            assert_bit_exact(run_hip(a, b, mode), och.chamfer_forward(a, b))


def test_full_frame_tree_equals_brute_force_and_invariants():
    """BASELINE-size frame (64 x 2048 rays, ~111k points per cloud): tree search == brute force bit for bit; the
    oracle is checked on a sample of queries (whole-frame oracle = 2.5e10 pair evaluations)."""
    a, b = lidar_clouds(64, 2048, 1)
    t = run_hip(a, b, 1); br = run_hip(a, b, 0)
    assert_bit_exact(t, br)
    d1, d2, i1, i2 = t
    N, M = a.shape[1], b.shape[1]
    assert i1.min() >= 0 and i1.max() < M and i2.min() >= 0 and i2.max() < N
    def d2f(x, y):
        dx, dy, dz = (y[:, 0] - x[:, 0]), (y[:, 1] - x[:, 1]), (y[:, 2] - x[:, 2])
        return dx, dy, dz
    # the reported distance is the float32 pair distance of the reported neighbour (fma chain, via float64 rounding check)
    for d, i, q, c in ((d1[0], i1[0], a[0], b[0]), (d2[0], i2[0], b[0], a[0])):
        dx, dy, dz = d2f(q, c[i])
        ref = (dx.astype(np.float64) ** 2 + dy.astype(np.float64) ** 2 + dz.astype(np.float64) ** 2)
        np.testing.assert_allclose(d, ref, rtol=3e-7, atol=0)
    sel = np.random.default_rng(0).choice(N, 4096, replace=False)
    want = och.chamfer_forward(a[:, sel], b)
    assert np.array_equal(d1[0, sel], want[0][0]) and np.array_equal(i1[0, sel], want[2][0])
    # swapping the clouds swaps the outputs
    sw = run_hip(b, a, 1)
    assert_bit_exact((sw[1], sw[0], sw[3], sw[2]), t)


def test_backward_matches_oracle_and_accumulates():
    a, b = lidar_clouds(16, 512, 2)
    r = np.random.default_rng(4)
    N, M = a.shape[1], b.shape[1]
    g1 = r.standard_normal((1, N)).astype(np.float32); g2 = r.standard_normal((1, M)).astype(np.float32)
    _, _, i1, i2 = och.chamfer_forward(a, b)
    wa, wb = och.chamfer_backward(a, b, g1, g2, i1, i2, "f64")
    t = lambda x: torch.as_tensor(x, device=DEV)
    ga = torch.zeros(1, N, 3, device=DEV); gb = torch.zeros(1, M, 3, device=DEV)
    assert chamfer_3D.backward(t(a), t(b), ga, gb, t(g1), t(g2), t(i1), t(i2)) == 1
    scale = max(np.abs(wa).max(), np.abs(wb).max())
    np.testing.assert_allclose(ga.cpu().numpy(), wa, rtol=1e-5, atol=1e-5 * scale)
    np.testing.assert_allclose(gb.cpu().numpy(), wb, rtol=1e-5, atol=1e-5 * scale)
    chamfer_3D.backward(t(a), t(b), ga, gb, t(g1), t(g2), t(i1), t(i2))        # second call adds (reference binding semantics)
    np.testing.assert_allclose(ga.cpu().numpy(), 2 * wa, rtol=1e-5, atol=2e-5 * scale)


def test_autograd_module_like_train_py():
    """train.py:197-207: dist1, dist2, _, _ = chamLoss(pred[None], gt[None]); loss = (dist1 + dist2).mean() * 0.5"""
    a, b = lidar_clouds(16, 512, 6, drop=0.0)
    pa = torch.as_tensor(a, device=DEV).requires_grad_(True); pb = torch.as_tensor(b, device=DEV).requires_grad_(True)
    d1, d2, i1, i2 = chamfer_3DDist()(pa, pb)
    assert i1.dtype == torch.int32 and not i1.requires_grad
    loss = (d1 + d2).mean() * 0.5
    loss.backward()
    w = och.chamfer_forward(a, b)
    n = a.shape[1]
    ga, gb = och.chamfer_backward(a, b, np.full((1, n), 0.5 / n, np.float32), np.full((1, n), 0.5 / n, np.float32), w[2], w[3], "f64")
    assert abs(loss.item() - (w[0].astype(np.float64) + w[1]).mean() * 0.5) <= 1e-5 * abs(loss.item())
    np.testing.assert_allclose(pa.grad.cpu().numpy(), ga, rtol=1e-5, atol=1e-6 * np.abs(ga).max())
    np.testing.assert_allclose(pb.grad.cpu().numpy(), gb, rtol=1e-5, atol=1e-6 * np.abs(gb).max())


def test_non_contiguous_and_wrong_inputs_are_rejected():
    a = torch.zeros(1, 8, 3, device=DEV)
    with pytest.raises(RuntimeError, match="contiguous"):
        chamfer_3D.forward(a.transpose(1, 2).contiguous().transpose(1, 2), a, torch.zeros(1, 8, device=DEV), torch.zeros(1, 8, device=DEV),
                           torch.zeros(1, 8, device=DEV, dtype=torch.int32), torch.zeros(1, 8, device=DEV, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="idx1 must be torch.int32"):
        chamfer_3D.forward(a, a, torch.zeros(1, 8, device=DEV), torch.zeros(1, 8, device=DEV),
                           torch.zeros(1, 8, device=DEV, dtype=torch.int64), torch.zeros(1, 8, device=DEV, dtype=torch.int32))
    e = torch.zeros(1, 0, 3, device=DEV)
    with pytest.raises(Exception, match="N >= 1"):
        chamfer_3D.forward(e, a, torch.zeros(1, 0, device=DEV), torch.zeros(1, 8, device=DEV),
                           torch.zeros(1, 0, device=DEV, dtype=torch.int32), torch.zeros(1, 8, device=DEV, dtype=torch.int32))
