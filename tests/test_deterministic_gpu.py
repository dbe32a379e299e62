"""Option deterministic (`Tracer(deterministic=True)`, `ShardedTracer(deterministic=True)`, `python -m lidar_rt_amd.train --deterministic`).

The reference adds every hit's 58 gradient components to its Gaussian with float atomics (backward.cu:615,659-669): its gradients depend on the
order in which the atomics arrive.  The bucketed backward here adds a Gaussian's records up in the arrival order of integer LDS atomics (its
cursors) and the pieces of a run that crosses waves with float atomics -- equal to the reference's to rounding, different from run to run in
the last bits, which Adam (eps 1e-15) amplifies.  With the option the sums are taken in a fixed order (k_bk_sort's second pass: a run's
records by ray, long runs through a ray bitmap; k_bwd_fixup: the pieces of a run in wave order), the forward keeps no learnt state and the
hit weights come from the backward.  Checked here: independent states, and states with a different history, give the same bits; the values
are the default mode's to rounding; both ways of ordering a run (counting loop, bitmap) and runs over three and more waves are exercised.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from lidar_rt_amd import scenes
from lidar_rt_amd.parallel import ShardedTracer

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
KEYS = ("means", "scales", "rotations", "opacities", "shs", "accum")


def _scene_with_a_wall(which):
    """The scene plus one large opaque-ish surfel 6 m in front of the sensor's first columns: hundreds of rays hit it (a run of records that is
    ordered through the bitmap and spans several waves of the reduction)."""
    if which == "s200k":
        sc = scenes.make_scene(200_000, radius_scale=0.5); o, d = scenes.kitti_rays(32, 512)
    else:
        sc, o, d = scenes.s10k()
    sc = {k: np.array(v, np.float32, copy=True) for k, v in sc.items()}
    c = np.asarray(o[o.shape[0] // 2, 8], np.float32) + 6.0 * np.asarray(d[d.shape[0] // 2, 8], np.float32)
    sc["means"][0] = c; sc["scales"][0] = (2.5, 2.5); sc["opacities"][0] = 0.35
    n = -np.asarray(d[d.shape[0] // 2, 8], np.float64); n /= np.linalg.norm(n)       # the surfel faces the sensor: rotation taking z to n
    z = np.array([0.0, 0.0, 1.0]); v = np.cross(z, n); s = np.linalg.norm(v); cth = float(z @ n)
    q = np.array([1.0, 0, 0, 0]) if s < 1e-9 else np.concatenate(([np.cos(0.5 * np.arctan2(s, cth))], np.sin(0.5 * np.arctan2(s, cth)) * v / s))
    sc["rotations"][0] = q.astype(np.float32)
    return sc, o, d


def _step(tr, t, o, d, dL, n=1):
    bg = torch.as_tensor(scenes.BG_DEFAULT, device=DEV)
    for _ in range(n):
        out, _ = tr.forward(o, d, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg)
        g = tr.backward(t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg, dL)
    torch.cuda.synchronize()
    return out.clone(), {k: g[k].clone() for k in KEYS}


def _hits_per_gaussian(state, HW, P):
    idx, h = state.handle(DEV)
    cap = state.get_option("hit_cap", DEV)
    hn = np.empty(HW, np.int32); hg = np.empty((HW, cap), np.int32)
    state._lib.lrt_debug_read.restype = C.c_longlong
    for which, arr in ((5, hn), (7, hg)):
        state._lib.lrt_debug_read(h, which, arr.ctypes.data_as(C.c_void_p), C.c_longlong(arr.nbytes), None)
    used = np.arange(cap)[None, :] < hn[:, None]
    return np.bincount(hg[used], minlength=P)


@pytest.mark.parametrize("which", ["s10k", "s200k"])
def test_independent_states_and_other_histories_give_the_same_bits(which):
    sc, o, d = _scene_with_a_wall(which)
    H, W = o.shape[:2]
    t = {k: torch.as_tensor(v, device=DEV) for k, v in sc.items()}
    ro, rd = torch.as_tensor(np.asarray(o, np.float32), device=DEV), torch.as_tensor(np.asarray(d, np.float32), device=DEV)
    dL = torch.as_tensor(scenes.upstream_grad(H, W), device=DEV)
    a = ShardedTracer(deterministic=True)
    out_a, g_a = _step(a, t, ro, rd, dL)
    per_g = _hits_per_gaussian(a.backend.state, H * W, sc["means"].shape[0])
    assert per_g.max() > 200 and (per_g[per_g > 0] <= 96).any(), (int(per_g.max()),)      # both ways of ordering a run; a run over more than three waves
    # a second, fresh state
    out_b, g_b = _step(ShardedTracer(deterministic=True), t, ro, rd, dL)
    # a state that has traced another pose and other parameters before, several times (learnt tables, carried order, the last build's box)
    c = ShardedTracer(deterministic=True)
    t2 = dict(t); t2["means"] = t["means"] + 0.01
    _step(c, t2, (ro + torch.tensor([0.3, -0.2, 0.05], device=DEV)).contiguous(), rd, dL, n=3)
    out_c, g_c = _step(c, t, ro, rd, dL, n=2)
    for name, (out_x, g_x) in (("fresh", (out_b, g_b)), ("another history", (out_c, g_c))):
        assert torch.equal(out_a, out_x), name
        for k in KEYS:
            assert torch.equal(g_a[k], g_x[k]), (name, k, int((g_a[k] != g_x[k]).sum()))
    # the same numbers as the default mode, to the rounding of another order of the sums (and of other slab widths in the image)
    out_n, g_n = _step(ShardedTracer(), t, ro, rd, dL)
    assert float((out_a - out_n).abs().max()) <= 5e-5
    for k in KEYS:
        num, den = float((g_a[k].double() - g_n[k].double()).norm()), float(g_n[k].double().norm())
        assert num <= 2e-6 * den, (k, num / den)
    assert not all(torch.equal(g_a[k], g_n[k]) for k in KEYS) or which == "s10k"             # (the default's order is not this one)


def test_switching_the_option_off_restores_the_learnt_tables():
    from lidar_rt_amd.diff_lidar_tracer import Tracer
    tr = Tracer(deterministic=True)
    st = tr.optix_context
    assert tr.deferred_accum and st.get_option("deterministic", DEV) == 1 and st.get_option("carry_order", DEV) == 0
    st.set_option("deterministic", 0)
    assert st.get_option("deterministic", DEV) == 0 and st.get_option("carry_order", DEV) == 1
