"""Deferred hit weights (`Tracer(deferred_accum=True)`, library option deferred_accum, lrt_backward_accum).

The reference adds every composited hit's weight to `accum[gidx]` with a float atomic in the FORWARD (forward.cu:268) and its training
loop reads the sums after `loss.backward()` (train.py:156,219).  With the option a training forward returns `accum` all-zero and the
backward of that forward writes the same sums into the same tensor from its Gaussian-ordered records (k_bwd_reduce4's spare column),
from the re-tracing kernel's atomics where that runs.  Checked here: the sums equal the oracle's and the exact-at-forward path's in
every backward path, evaluation-mode forwards are untouched, a second backward does not add twice.
"""
import numpy as np
import pytest
import torch

from lidar_rt_amd import scenes
from oracle import oracle
from lidar_rt_amd.diff_lidar_tracer import Tracer

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from tests.hip_util import settings, rel_l2, frac_outside, DEV, DEFAULT_OPTS

GRADS = ("means", "scales", "rotations", "opacities", "shs")
# (options, what runs): the bucketed replay (default), the re-tracing backward asked for, the re-tracing backward behind a hit record
# that overflowed (hit_cap 8: decided from the forward's status), 16 waves per tile; the legacy modes 1 / 2 where the library has them
PATHS = [({}, "bucketed"), ({"bwd_mode": 0}, "re-trace"), ({"hit_cap": 8, "hit_cap_auto": 0}, "record overflow -> re-trace"),
         ({"c4_waves": 16}, "16 waves"), ({"fwd_mode": 0, "bwd_mode": 0}, "packet kernel both ways")]


def _oracle(sc, o, d, deg, bg, dL):
    orc = oracle.Oracle(sc["means"], sc["scales"], sc["rotations"], sc["opacities"], "f32")
    fw = orc.forward(o, d, sc["shs"], deg, bg)
    return fw, orc.backward(o, d, sc["shs"], deg, bg, fw["out"], dL)


def _run(sc, o, d, deg, bg, dL, deferred, opts=None, backward=True, training=True, twice=False):
    tr = Tracer(deferred_accum=deferred)
    if not training:
        tr.eval()
    for k, v in {**DEFAULT_OPTS, **(opts or {})}.items():
        tr.optix_context.set_option(k, v)
    tr.optix_context.set_option("deferred_accum", 1 if deferred else 0)       # the state is per device: reset what an earlier test set
    t = {k: torch.as_tensor(np.asarray(v, np.float32), device=DEV).requires_grad_(True) for k, v in sc.items()}
    ro, rd = torch.as_tensor(o, device=DEV), torch.as_tensor(d, device=DEV)
    tr.build_from_gaussians(t["means"], t["scales"], t["rotations"], t["opacities"])
    out, acc = tr(ro, rd, None, t["means"], torch.zeros_like(t["means"]), shs=t["shs"], opacities=t["opacities"], scales=t["scales"],
                  rotations=t["rotations"], tracer_settings=settings(bg, deg))
    torch.cuda.synchronize()
    acc_fwd = acc.detach().cpu().numpy().copy()
    res = {"out": out.detach().cpu().numpy(), "accum_after_forward": acc_fwd}
    if backward:
        g = torch.as_tensor(dL, device=DEV)
        out.backward(g, retain_graph=twice)
        if twice:
            for v in t.values():
                v.grad = None
            out.backward(g)
        torch.cuda.synchronize()
        res["grads"] = {k: t[k].grad.detach().cpu().numpy() for k in GRADS}
    res["accum"] = acc.detach().cpu().numpy()
    tr.optix_context.set_option("deferred_accum", 0)
    return res


@pytest.fixture(scope="module")
def s10k():
    sc, o, d = scenes.s10k()
    return sc, o, d, scenes.upstream_grad(16, 256)


@pytest.mark.parametrize("opts,what", PATHS, ids=[w for _, w in PATHS])
def test_deferred_weights_equal_the_oracle_and_the_forward_atomics_on_s10k(s10k, opts, what):
    sc, o, d, dL = s10k
    fw, bw = _oracle(sc, o, d, 3, scenes.BG_DEFAULT, dL)
    exact = _run(sc, o, d, 3, scenes.BG_DEFAULT, dL, deferred=False, opts=opts)
    h = _run(sc, o, d, 3, scenes.BG_DEFAULT, dL, deferred=True, opts=opts)
    assert not h["accum_after_forward"].any(), "a deferred training forward must leave accum all-zero"
    assert exact["accum_after_forward"].any()
    if "hit_cap" in opts:      # hits beyond the record reach the colour pass through an overflow list in arrival order: equal to rounding
        np.testing.assert_allclose(h["out"], exact["out"], rtol=2e-6, atol=1e-7)
    else:
        np.testing.assert_array_equal(h["out"], exact["out"])                     # the image does not know about the option
    assert frac_outside(h["accum"], fw["accum"], 1e-4) <= 1e-3 and rel_l2(h["accum"], fw["accum"]) < 1e-5, what
    assert rel_l2(h["accum"], exact["accum"]) < 2e-6, what                          # the same weights, added in another order
    np.testing.assert_array_equal(h["accum"] > 0, exact["accum"] > 0)               # the exact touched set (the sharded exchange's mask)
    for k in GRADS:
        assert rel_l2(h["grads"][k], exact["grads"][k]) < 2e-6, (what, k)
        assert rel_l2(h["grads"][k].reshape(bw[k].shape), bw[k]) < 1e-3, (what, k)


def test_legacy_backward_modes_fill_the_weights_too(s10k):
    """bwd_mode 1 (replay + atomics) and 2 (sorted reduction) exist in the cross-check build only (-DLRT_LEGACY)."""
    from lidar_rt_amd import _capi
    sc, o, d, dL = s10k
    fw, _ = _oracle(sc, o, d, 3, scenes.BG_DEFAULT, dL)
    ran = 0
    for mode in (1, 2):
        try:
            h = _run(sc, o, d, 3, scenes.BG_DEFAULT, dL, deferred=True, opts={"bwd_mode": mode})
        except _capi.LrtError as ex:
            assert "LRT_LEGACY" in str(ex)
            continue
        ran += 1
        assert not h["accum_after_forward"].any()
        assert rel_l2(h["accum"], fw["accum"]) < 1e-5, mode
    if ran == 0:
        pytest.skip("the shipped library has no bwd_mode 1 / 2 (built without -DLRT_LEGACY)")


def test_evaluation_forward_is_exact_at_once_and_a_second_backward_adds_nothing_twice(s10k):
    sc, o, d, dL = s10k
    fw, _ = _oracle(sc, o, d, 3, scenes.BG_DEFAULT, dL)
    ev = _run(sc, o, d, 3, scenes.BG_DEFAULT, dL, deferred=True, backward=False, training=False)
    assert rel_l2(ev["accum_after_forward"], fw["accum"]) < 1e-5, "an evaluation-mode forward keeps the reference's contract"
    for opts in ({}, {"bwd_mode": 0}):
        h2 = _run(sc, o, d, 3, scenes.BG_DEFAULT, dL, deferred=True, opts=opts, twice=True)
        assert rel_l2(h2["accum"], fw["accum"]) < 1e-5, opts


@pytest.mark.parametrize("M,deg", [(9, 2), (16, 1), (17, 3)])
def test_deferred_weights_with_other_sh_table_widths(s10k, M, deg):
    """The weight rides in the lane behind the gradient row (10 + 3 M): M = 17 is the widest table that leaves one."""
    sc, o, d, dL = s10k
    sc = dict(sc)
    if M <= 16:
        sc["shs"] = np.ascontiguousarray(sc["shs"][:, :M])
    else:
        sc["shs"] = np.ascontiguousarray(np.concatenate([sc["shs"], np.zeros((sc["shs"].shape[0], M - 16, 3), np.float32)], 1))
    fw, bw = _oracle(sc, o, d, deg, scenes.BG_DEFAULT, dL)
    h = _run(sc, o, d, deg, scenes.BG_DEFAULT, dL, deferred=True)
    assert rel_l2(h["accum"], fw["accum"]) < 1e-5
    for k in GRADS:
        assert rel_l2(h["grads"][k].reshape(bw[k].shape), bw[k]) < 1e-3, k


def test_deferred_weights_on_s200k_and_through_the_renderer():
    sc = scenes.make_scene(200_000, radius_scale=0.5); o, d = scenes.kitti_rays(32, 512)
    dL = scenes.upstream_grad(32, 512)
    fw, _ = _oracle(sc, o, d, 3, scenes.BG_DEFAULT, dL)
    exact = _run(sc, o, d, 3, scenes.BG_DEFAULT, dL, deferred=False)
    for opts in ({}, {"bwd_mode": 0}):
        h = _run(sc, o, d, 3, scenes.BG_DEFAULT, dL, deferred=True, opts=opts)
        assert not h["accum_after_forward"].any()
        # against the oracle: the large scenes' threshold events (a hit composited by one implementation and not by the other), like the
        # exact-at-forward weights; against those weights themselves: the same hits (the packet kernel of bwd_mode 0 decides a few knife edges its own way)
        assert frac_outside(h["accum"], fw["accum"], 1e-4) <= 2e-3 and rel_l2(h["accum"], fw["accum"]) < 2e-3, opts
        assert rel_l2(h["accum"], exact["accum"]) < (2e-6 if not opts else 2e-3), opts
    # renderer.raytracing with the module switch: train.py's read of accum_gaussian_weight behind loss.backward()
    import types
    from lidar_rt_amd import renderer
    from tests.test_renderer_gpu import Asset
    s10 = scenes.s10k()
    asset = Asset(s10[0], slice(None))
    args = types.SimpleNamespace(dynamic=False, opt=types.SimpleNamespace(use_rayhit=False), pipe=types.SimpleNamespace())
    sensor = (torch.as_tensor(s10[1], device=DEV), torch.as_tensor(s10[2], device=DEV), torch.zeros(3, device=DEV))
    got = {}
    old = renderer.deferred_accum, renderer.tracer_2dgs
    try:
        for flag in (False, True):
            renderer.deferred_accum, renderer.tracer_2dgs = flag, None
            for p in asset.params():
                p.grad = None
            pkg = renderer.raytracing(0, [asset], sensor, torch.tensor(scenes.BG_DEFAULT), args)
            before = pkg["accum_gaussian_weight"].detach().clone()
            (pkg["depth"].sum() + pkg["intensity"].sum()).backward()
            torch.cuda.synchronize()
            got[flag] = (before.cpu().numpy(), pkg["accum_gaussian_weight"].detach().cpu().numpy())
    finally:
        renderer.deferred_accum, renderer.tracer_2dgs = old
    assert got[False][0].any() and not got[True][0].any()
    assert rel_l2(got[True][1], got[False][1]) < 2e-6
