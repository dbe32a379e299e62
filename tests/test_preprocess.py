"""Fused Gaussian pre-processing: the CPU restatement against the golden vectors produced by the
reference's own functions, and the HIP op against both (forward and vector-Jacobian product)."""
import os

import numpy as np
import pytest
import torch

from oracle import preprocess_ref as ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "preprocess_golden.npz")
RAW = ("xyz", "log_scales", "rot_raw", "opacity_logit")
OUT = ("means", "scales", "rotations", "opacities")


def load():
    g = np.load(GOLD)
    return g, {k: torch.from_numpy(g["in_" + k]) for k in RAW}, torch.from_numpy(g["seg_start"]), torch.from_numpy(g["poses"])


def test_restatement_matches_reference_functions():
    g, raw, seg, poses = load()
    t = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
    outs = ref.preprocess(t["xyz"], t["log_scales"], t["rot_raw"], t["opacity_logit"], seg, poses)
    for o, name in zip(outs, OUT):
        np.testing.assert_allclose(o.detach().numpy(), g["out_" + name], rtol=1e-6, atol=1e-6)
    sum((o * torch.from_numpy(g["up_" + name])).sum() for o, name in zip(outs, OUT)).backward()
    for k in RAW:
        np.testing.assert_allclose(t[k].grad.numpy(), g["grad_" + k], rtol=1e-5, atol=1e-5)
    rot = outs[2].detach().numpy()
    np.testing.assert_allclose(np.linalg.norm(rot, axis=1), 1.0, atol=1e-5)          # unit quaternions, actors included


def test_pack_poses_host_logic():
    from lidar_rt_amd.preprocess import pack_poses
    seg, tab = pack_poses([None, (torch.tensor([1.0, 2, 3]), torch.tensor([[0.0, 1, 0, 0]]))], [5, 7], "cpu")
    assert seg.tolist() == [0, 5, 12] and seg.dtype == torch.int32
    assert tab.shape == (2, 8) and tab[0].tolist() == [0, 0, 0, 1, 0, 0, 0, 0] and tab[1].tolist() == [1, 2, 3, 0, 1, 0, 0, 1]


@pytest.mark.gpu
def test_hip_forward_backward_match_golden():
    from lidar_rt_amd.preprocess import fused_activations
    g, raw, seg, poses = load()
    dev = torch.device("cuda:0")
    t = {k: v.to(dev).requires_grad_(True) for k, v in raw.items()}
    outs = fused_activations(t["xyz"], t["log_scales"], t["rot_raw"], t["opacity_logit"], seg.to(dev), poses.to(dev))
    for o, name in zip(outs, OUT):
        np.testing.assert_allclose(o.detach().cpu().numpy(), g["out_" + name], rtol=2e-6, atol=2e-6)
    sum((o * torch.from_numpy(g["up_" + name]).to(dev)).sum() for o, name in zip(outs, OUT)).backward()
    for k in RAW:
        np.testing.assert_allclose(t[k].grad.cpu().numpy(), g["grad_" + k], rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("counts", [[1], [1000], [300, 1, 0, 77], [64, 64, 64]])
def test_hip_matches_restatement_on_random_assets(counts):
    from lidar_rt_amd.preprocess import fused_activations, pack_poses
    dev = torch.device("cuda:0")
    r = np.random.default_rng(sum(counts))
    P = sum(counts)
    raw = {"xyz": r.normal(size=(P, 3)) * 20, "log_scales": r.normal(-2, 1, (P, 2)), "rot_raw": r.normal(size=(P, 4)) * 2,
           "opacity_logit": r.normal(0, 3, (P, 1))}
    raw = {k: torch.from_numpy(v.astype(np.float32)) for k, v in raw.items()}
    ps = [None] + [(torch.from_numpy(r.normal(size=3).astype(np.float32)), torch.from_numpy(r.normal(size=4).astype(np.float32)))
                   for _ in counts[1:]]                       # un-normalised actor quaternions: R normalises, the product does not
    seg, tab = pack_poses(ps, counts, "cpu")
    up = [torch.from_numpy(r.normal(size=s).astype(np.float32)) for s in ((P, 3), (P, 2), (P, 4), (P, 1))]
    tc = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
    oc = ref.preprocess(tc["xyz"], tc["log_scales"], tc["rot_raw"], tc["opacity_logit"], seg, tab)
    sum((o * u).sum() for o, u in zip(oc, up)).backward()
    tg = {k: v.to(dev).requires_grad_(True) for k, v in raw.items()}
    og = fused_activations(tg["xyz"], tg["log_scales"], tg["rot_raw"], tg["opacity_logit"], seg.to(dev), tab.to(dev))
    sum((o * u.to(dev)).sum() for o, u in zip(og, up)).backward()
    for a, b, name in zip(og, oc, OUT):                       # world means: sums of products of O(20) terms -> a few ulp of 20
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), rtol=3e-6, atol=3e-5 if name == "means" else 3e-6)
    for k in RAW:
        np.testing.assert_allclose(tg[k].grad.cpu().numpy(), tc[k].grad.numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.gpu
def test_cpu_tensors_are_rejected():
    from lidar_rt_amd.preprocess import fused_activations
    z = torch.zeros
    with pytest.raises(RuntimeError, match="HIP|cuda"):
        fused_activations(z(4, 3), z(4, 2), z(4, 4), z(4, 1), z(2, dtype=torch.int32), z(1, 8))
