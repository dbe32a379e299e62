"""Quads closer than 0.2 m to the ray origin: the reference skips such a hit WITHOUT clearing its K-buffer slot
(DLT/optix_tracer/forward.cu:214 precedes :218-219), and the stale slot changes what the following 16-candidate chunks hold.
The trace kernels list such rays and k_fwd_near (csrc/lrt_near.inc) replays the K-buffer literally; checked against the oracle."""
import numpy as np
import pytest
import torch

from lidar_rt_amd import scenes
from oracle import oracle

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from tests.hip_util import run_hip, rel_l2, frac_outside
    from tests.test_hip_parity import _facing, oracle_run

from lidar_rt_amd import _capi
LEGACY = _capi.has_legacy()       # the cross-check library (tests/test_legacy_crosscheck_gpu.py): colours inside the trace kernel, the sorted backward
MODES = [{"fwd_mode": 2, "defer_colour": 1}, {"fwd_mode": 2, "defer_colour": 0}, {"fwd_mode": 0, "bwd_mode": 2 if LEGACY else 3}, {"fwd_mode": 2, "defer_colour": 1, "c4_waves": 8},
         {"fwd_mode": 2, "defer_colour": 1, "bwd_mode": 0}, {"fwd_mode": 0, "bwd_mode": 0}]      # the last two: the re-tracing backward replays the near rays itself
IDS = ["collect4-defer", "collect4", "packet-forward", "collect4-8waves", "collect4-retrace-bwd", "packet-retrace-bwd"]
_SEL_A = [i for i in range(4) if LEGACY or i != 1]
_SEL_B = [i for i in (0, 1, 2, 4, 5) if LEGACY or i != 1]


@pytest.mark.parametrize("mode", [MODES[i] for i in _SEL_A], ids=[IDS[i] for i in _SEL_A])
def test_known_answers_with_a_hit_below_the_near_threshold(mode):
    # the ray passes a little off the quads' centres: through a centre it would hit the shared edge of a quad's two triangles, and
    # the reference would then spend two K-buffer slots on one quad
    o = np.zeros((1, 1, 3), np.float32); d = np.array([[[1.0, 0.011, 0.004]]], np.float32); d /= np.linalg.norm(d)
    bg = np.zeros(3, np.float32)
    # (a) fewer than 16 hits: the near one is skipped, nothing else changes (no restart, the stale slot is never looked at again)
    xs = [0.1] + list(np.linspace(2.0, 8.5, 14))
    sc = _facing(xs, [0.05] * 15)
    fw, _ = oracle_run(sc, o, d, 0, bg)
    assert fw["n_comp"][0, 0] == 14
    h = run_hip(sc, o, d, 0, bg, opts=mode)
    np.testing.assert_allclose(h["out"][0, 0], fw["out"][0, 0], rtol=2e-5, atol=1e-7)
    # (b) 18 hits, one of them near: chunk 1 holds the near hit + 15 real ones; the stale slot sorts to the front of chunk 2, whose
    # loop stops after cnt = 2 entries: the stale one -- re-evaluated at the fictitious depth 8.1 m, on this ray close to its
    # Gaussian's axis, so it IS composited there -- and ONE of the two real ones.  The reference loses the last real hit.
    xs = [0.1] + list(np.linspace(1.0, 9.0, 17))
    sc = _facing(xs, [0.05] * 18)
    fw, _ = oracle_run(sc, o, d, 0, bg)
    assert fw["n_comp"][0, 0] == 17 and fw["accum"][-1] == 0 and fw["accum"][0] > 0        # 15 + the fictitious one + 1; the last is lost
    h = run_hip(sc, o, d, 0, bg, opts=mode)
    np.testing.assert_allclose(h["out"][0, 0], fw["out"][0, 0], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(h["accum"], fw["accum"], rtol=2e-5, atol=1e-8)
    # (c) two near hits and 40 real ones: with more new candidates than free slots the reference's `cnt` depends on the order in
    # which the any-hit program sees them (BVH order in OptiX); k_fwd_near feeds it in ascending t, and so does the oracle here
    xs = [0.05, 0.15] + list(np.linspace(1.0, 20.5, 40))
    sc = _facing(xs, [0.04] * 42)
    oracle.set_sorted_anyhit(True)
    try:
        fw, _ = oracle_run(sc, o, d, 0, bg)
    finally:
        oracle.set_sorted_anyhit(False)
    assert fw["n_comp"][0, 0] < 40                                             # hits are lost to the stale slots
    h = run_hip(sc, o, d, 0, bg, opts=mode)
    np.testing.assert_allclose(h["out"][0, 0], fw["out"][0, 0], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(h["accum"], fw["accum"], rtol=2e-5, atol=1e-8)


@pytest.mark.parametrize("mode", [MODES[i] for i in _SEL_B], ids=[IDS[i] for i in _SEL_B])
def test_sensor_inside_the_geometry_matches_the_oracle(mode):
    """A sensor in the middle of dense clutter: about a third of the rays have a quad within 0.2 m.  Forward and (replay) backward
    against the oracle whose any-hit program is fed in ascending t, the realisation of the reference's order-dependent
    behaviour that k_fwd_near replays; the rays without a near quad are unaffected and compared as usual."""
    rng = np.random.default_rng(5)
    P = 6000
    sc = scenes.make_scene(P, seed=12, radius_scale=0.3)
    sc["means"] = (rng.uniform(-2.5, 2.5, (P, 3)) * np.array([1, 1, 0.4])).astype(np.float32)      # a box of clutter around the origin
    sc["scales"] = (sc["scales"] * 0.5).astype(np.float32)
    o, d = scenes.kitti_rays(8, 96)
    dL = scenes.upstream_grad(8, 96, seed=4)
    oracle.set_sorted_anyhit(True)
    try:
        fw, bw = oracle_run(sc, o, d, 3, scenes.BG_DEFAULT, dL)
    finally:
        oracle.set_sorted_anyhit(False)
    h = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL, opts=mode)
    # how many rays are near rays?  (brute force: any quad hit with 0 < t < 0.2)
    from oracle.bruteforce import QuadScene
    qs = QuadScene(sc["means"], sc["scales"], sc["rotations"], sc["opacities"])
    near = np.array([bool((qs.candidates(oo, dd)[1] < 0.2).any()) for oo, dd in zip(o.reshape(-1, 3), d.reshape(-1, 3))])
    assert 0.1 < near.mean() < 0.9, near.mean()
    err = np.abs(h["out"] - fw["out"]).reshape(-1, 9).max(1) / np.maximum(np.abs(fw["out"]).reshape(-1, 9).max(1), 1e-3)
    assert (err[near] > 1e-3).mean() <= 0.03 and (err[~near] > 1e-3).mean() <= 0.01, ((err[near] > 1e-3).mean(), (err[~near] > 1e-3).mean())
    assert rel_l2(h["out"], fw["out"]) < 5e-3
    assert rel_l2(h["accum"], fw["accum"]) < 1e-2
    if mode.get("fwd_mode") == 2 or mode.get("bwd_mode") == 0:     # replay backward of the recorded (near-ray) hits / re-tracing backward with its own near-ray replay
        for k in ("means", "opacities", "shs"):
            assert rel_l2(h["grads"][k].reshape(bw[k].shape), bw[k]) < 2e-3, k        # measured < 5e-5 in every mode (tests/tools/near_dbg.py)
