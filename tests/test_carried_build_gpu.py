"""Builds on the CARRIED order (round 6).  The reference rebuilds its acceleration structure from scratch on every call
(DLT/trace_surfels.cpp:46-148); here a build of an unchanged number of Gaussians keeps the Morton order of the last full sort -- between two
training iterations the parameters move by an optimizer step -- and only re-derives records and boxes (k_pack + k_make_tree); it sorts again
when the order is `carry_max_age` builds old, when too many neighbours in it are out of Morton order (counted on the device), or when P
changes.  A ray-culled build (a rank's azimuth slab) selects RANGES of that order by a stale index of range boxes, inflated by the measured
drift of the parameters since the index's snapshot.  The traversal is exhaustive and hits are ordered by (t, index), so none of this may
change a result: checked here against builds that sort every time, and against the oracle.
"""
import numpy as np
import pytest
import torch

from lidar_rt_amd import scenes
from lidar_rt_amd.diff_lidar_tracer import Tracer, _C

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from tests.hip_util import settings, rel_l2, DEV, DEFAULT_OPTS

GRADS = ("means", "scales", "rotations", "opacities", "shs")


def _tracer(**opts):
    tr = Tracer()
    for k, v in {**DEFAULT_OPTS, **opts}.items():
        tr.optix_context.set_option(k, v)
    return tr


def _step(tr, sc, o, d, dL, cull=None):
    t = {k: torch.as_tensor(np.asarray(v, np.float32), device=DEV).requires_grad_(True) for k, v in sc.items()}
    ro, rd = torch.as_tensor(o, device=DEV), torch.as_tensor(d, device=DEV)
    _C.build_from_gaussians(tr.optix_context, t["means"], t["scales"], t["rotations"], t["opacities"], 1.0, cull_rays=(ro, rd) if cull else None)
    out, acc = tr(ro, rd, None, t["means"], torch.zeros_like(t["means"]), shs=t["shs"], opacities=t["opacities"], scales=t["scales"],
                  rotations=t["rotations"], tracer_settings=settings(scenes.BG_DEFAULT, 3))
    out.backward(torch.as_tensor(dL, device=DEV))
    torch.cuda.synchronize()
    return {"out": out.detach().cpu().numpy(), "accum": acc.detach().cpu().numpy(), "grads": {k: t[k].grad.cpu().numpy() for k in GRADS}}


def _moved(sc, rng, it, step=2e-3):
    m = dict(sc)
    m["means"] = (sc["means"] + step * it * rng.normal(size=sc["means"].shape)).astype(np.float32)
    m["scales"] = (sc["scales"] * np.exp(0.01 * it * rng.normal(size=sc["scales"].shape))).astype(np.float32)
    m["opacities"] = np.clip(sc["opacities"] + 0.01 * it * rng.normal(size=sc["opacities"].shape), 0.002, 0.99).astype(np.float32)
    return m


@pytest.mark.parametrize("P,hw,own", [(3_000, (8, 64), 2), (40_000, (16, 256), 2), (300_000, (32, 512), 1), (300_000, (32, 512), 0)])
def test_carried_builds_equal_builds_that_sort_every_time(P, hw, own):
    """Eight frames of slowly moving parameters: the carried tracer sorts once (and once more when its age limit of 5 is reached), the
    reference tracer sorts in every build; both build sorts (own onesweep / rocPRIM).  Same images bit for bit -- also the learned first-slab
    widths see the same hit counts --, same touched sets, gradients equal to the backward's summation noise."""
    sc = scenes.make_scene(P, seed=3 + P, radius_scale=0.5 if P > 100_000 else 0.25)
    o, d = scenes.kitti_rays(*hw)
    dL = scenes.upstream_grad(*hw)
    a, b = _tracer(carry_order=1, carry_max_age=5, carry_max_inv=1000000, own_sort=own), _tracer(carry_order=0, own_sort=own)      # (the age limit alone decides here)
    rng = np.random.default_rng(1)
    ages = []
    for it in range(8):
        m = _moved(sc, rng, it)
        ra, rb = _step(a, m, o, d, dL), _step(b, m, o, d, dL)
        ages.append(a.optix_context.get_option("carry_age", DEV))
        np.testing.assert_array_equal(ra["out"], rb["out"])
        np.testing.assert_array_equal(ra["accum"] > 0, rb["accum"] > 0)
        for k in GRADS:
            assert rel_l2(ra["grads"][k], rb["grads"][k]) < 2e-6, (it, k)
    assert ages == [0, 1, 2, 3, 4, 5, 0, 1], ages                            # one sort, five carried builds, the age limit, again
    assert b.optix_context.get_option("carry_age", DEV) == 0


def test_an_order_that_has_decayed_is_sorted_again():
    """The device counts neighbours of the carried order whose Morton cells are out of order; the count travels with the forward's status words.
    Small steps leave it near zero; a scene whose Gaussians have been shuffled in space makes the next-but-one build sort again."""
    P, hw = 60_000, (16, 256)
    sc = scenes.make_scene(P, seed=9, radius_scale=0.3)
    o, d = scenes.kitti_rays(*hw); dL = scenes.upstream_grad(*hw)
    tr = _tracer(carry_order=1, carry_max_age=1000)
    rng = np.random.default_rng(2)
    for it in range(4):
        _step(tr, _moved(sc, rng, it, step=1e-4), o, d, dL)
    tr.check(DEV)
    assert tr.optix_context.get_option("carry_age", DEV) == 3
    _step(tr, _moved(sc, rng, 1, step=1e-4), o, d, dL); tr.check(DEV)
    small = tr.optix_context.get_option("carry_inversions_last", DEV)
    assert small < P // 100, small
    shuffled = dict(sc); shuffled["means"] = (sc["means"] + rng.normal(0, 1.0, sc["means"].shape)).astype(np.float32)      # every centre a metre away: half of them in another Morton cell
    ref = _step(_tracer(carry_order=0), shuffled, o, d, dL)
    got = _step(tr, shuffled, o, d, dL); tr.check(DEV)                        # still on the old order: correct, and counted as decayed
    np.testing.assert_allclose(got["out"], ref["out"], rtol=2e-6, atol=1e-7)  # (another tracer state: other learned slab widths, other partial sums)
    assert tr.optix_context.get_option("carry_inversions_last", DEV) > P // 20
    got = _step(tr, shuffled, o, d, dL)
    assert tr.optix_context.get_option("carry_age", DEV) == 0                 # sorted again
    np.testing.assert_allclose(got["out"], ref["out"], rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("drift", [0.0, 0.02, 0.5, 30.0])
def test_a_stale_cull_index_with_drifted_gaussians_loses_no_primitive(drift):
    """a rank's culled build selects ranges of the carried order by an index whose boxes are `age` builds old.  The
    Gaussians drift away from the snapshot -- centres by up to `drift` metres, quads growing up to 3x, transparent ones becoming visible --
    and the culled build must still hold everything its rays can hit: image bit-identical to the unculled build of the SAME parameters, the
    same touched set, the same gradients.  (30 m: the index knows nothing any more and keeps every range.)"""
    from lidar_rt_amd.parallel import column_slab
    P = 50_000
    sc = scenes.make_scene(P, seed=41, radius_scale=0.3)
    sc["opacities"][::7] = 0.002                                              # cannot be hit at index time ...
    o, d = scenes.kitti_rays(16, 512)
    dL = scenes.upstream_grad(16, 512)
    rng = np.random.default_rng(7)
    for n_slabs, r in ((8, 3), (4, 0), (2, 1)):
        a_, b_ = column_slab(512, r, n_slabs)
        os_, ds_, g_ = np.ascontiguousarray(o[:, a_:b_]), np.ascontiguousarray(d[:, a_:b_]), np.ascontiguousarray(dL[:, a_:b_])
        # (spec_cull = 0: every culled build reads its kept count back -- the drift makes the index keep several times the ranges of the build
        # before, which a speculative size would report as error code 8: ShardedTracer's business, tested in tests/test_sharded_gpu.py)
        tr = _tracer(carry_order=1, carry_max_age=1000, carry_max_inv=1000000, spec_cull=0, learn_slab=0)      # (learn_slab = 0 on both sides: the same slab partition, bit-identical partial sums)
        first = _step(tr, sc, os_, ds_, g_, cull=True)                        # sorts, takes the snapshot, forms the range boxes
        kept0 = tr.optix_context.built_count(DEV)
        assert 0 < kept0 < (0.9 if n_slabs == 2 else 0.6) * P, (n_slabs, kept0)
        np.testing.assert_array_equal(first["out"], _step(_tracer(carry_order=0, learn_slab=0), sc, os_, ds_, g_)["out"])
        m = dict(sc)
        dirs = rng.normal(size=(P, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        m["means"] = (sc["means"] + drift * rng.uniform(0, 1, (P, 1)) * dirs).astype(np.float32)
        m["scales"] = (sc["scales"] * rng.uniform(0.5, 3.0 if drift > 0 else 1.0, (P, 2))).astype(np.float32)
        op = sc["opacities"].copy()
        if drift >= 0.5:
            op[::14] = 0.6                                                    # ... and half of them can now (their snapshot says nothing: every range is kept)
        m["opacities"] = op
        for rep in range(2):
            got = _step(tr, m, os_, ds_, g_, cull=True)                       # carried: k_drift + k_cone_cull on the stale index
            assert tr.optix_context.get_option("carry_age", DEV) == 1 + rep
            ref = _step(_tracer(carry_order=0, learn_slab=0), m, os_, ds_, g_)      # unculled, freshly sorted
            np.testing.assert_array_equal(got["out"], ref["out"])
            np.testing.assert_array_equal(got["accum"] > 0, ref["accum"] > 0)
            assert rel_l2(got["accum"], ref["accum"]) < 2e-6
            for k in GRADS:
                assert rel_l2(got["grads"][k], ref["grads"][k]) < 1e-5, (n_slabs, rep, k)
        slots = tr.optix_context.get_option("cull_last", DEV)                  # slots of the ranges the index kept (the exact test inside k_make_tree culls further)
        if drift >= 0.5:
            assert slots > 0.9 * P, slots                                     # the index knows nothing any more: (nearly) every range
        elif n_slabs == 8:
            assert slots < 0.5 * P, slots                                     # a little drift costs a little culling power, not all of it


def test_learnt_tile_tables_are_kept_per_named_ray_set():
    """Option ray_set names the rays of the next forwards (a training loop's frame index): first-slab widths, tile lengths and the queue boundaries made
    from them are learnt per name (256 names, least recently used replaced).  They are performance state only: two poses traced in turn under their names,
    under no name, and under 300 names in rotation (more than the library keeps) give the images of a fresh tracer to the rounding of the slab partial sums."""
    P, hw = 40_000, (16, 256)
    sc = scenes.make_scene(P, seed=5, radius_scale=0.3)
    o, d = scenes.kitti_rays(*hw); dL = scenes.upstream_grad(*hw)
    c_, s_ = np.cos(0.7), np.sin(0.7)
    R = np.array([[c_, -s_, 0.0], [s_, c_, 0.0], [0.0, 0.0, 1.0]], np.float32)
    poses = [(o, d), (np.ascontiguousarray(o + np.float32([0.3, -0.2, 0.05])), np.ascontiguousarray(d @ R.T))]
    ref = [_step(_tracer(), sc, po, pd, dL) for po, pd in poses]
    named, unnamed, many = _tracer(), _tracer(), _tracer()
    for it in range(12):
        k = it & 1
        named.optix_context.set_option("ray_set", 1000 + k)
        for tr in (named, unnamed):
            got = _step(tr, sc, *poses[k], dL)
            np.testing.assert_allclose(got["out"], ref[k]["out"], rtol=2e-6, atol=1e-7)
            np.testing.assert_array_equal(got["accum"] > 0, ref[k]["accum"] > 0)
            for g in GRADS:
                assert rel_l2(got["grads"][g], ref[k]["grads"][g]) < 5e-6, (it, g)
    for it in range(310):                                                     # 300 names in rotation, then the first ones again (replaced meanwhile)
        k = it & 1
        many.optix_context.set_option("ray_set", it % 300)
        if it % 31 == 0 or it >= 300:
            got = _step(many, sc, *poses[k], dL)
            np.testing.assert_allclose(got["out"], ref[k]["out"], rtol=2e-6, atol=1e-7)
        else:                                                                 # (forward only: the table is taken and learnt in the forward)
            t_ = {q: torch.as_tensor(np.asarray(v, np.float32), device=DEV) for q, v in sc.items()}
            many(torch.as_tensor(poses[k][0], device=DEV), torch.as_tensor(poses[k][1], device=DEV), None, t_["means"], torch.zeros_like(t_["means"]), shs=t_["shs"],
                 opacities=t_["opacities"], scales=t_["scales"], rotations=t_["rotations"], tracer_settings=settings(scenes.BG_DEFAULT, 3))
    for tr in (named, unnamed, many):
        tr.check(DEV)
