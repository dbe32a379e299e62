"""GPU parity tests: the HIP tracer (through the drop-in Tracer API -> `_C` torch extension -> C ABI) against the CPU oracle.

Tolerances (BASELINE.json north_star): outputs within 1e-4 relative, gradients within 1e-3 relative, fp32.
"Relative" is taken element-wise against max(|ref|, 1e-3 * max|ref|) (values far below the tensor's scale are
compared absolutely).  Two implementations of this algorithm cannot agree on EVERY ray: when two quads are hit at
distances closer than fp32 resolution their compositing order -- which the reference leaves to the rounding of
OptiX's triangle intersector -- flips, and thresholds (alpha >= 1/255, T < 1e-4) can flip for values within one ulp.
The tests therefore bound (a) the fraction of elements outside the tolerance and (b) the relative L2 error; the
same statistics for the fp32 oracle against the fp64 oracle are the noise floor of the algorithm itself.
"""
import os

import numpy as np
import pytest
import torch

from lidar_rt_amd import scenes
from oracle import oracle
from lidar_rt_amd.diff_lidar_tracer import Tracer

pytestmark = pytest.mark.gpu

from lidar_rt_amd import _capi
if torch.cuda.is_available():
    from tests.hip_util import run_hip, rel_l2, frac_outside, parity_report
    from tests.hip_util import DEFAULT_OPTS as DEFAULT_OPTS_

LEGACY = _capi.has_legacy()       # this process runs on liblrt_hip_legacy.so (tests/test_legacy_crosscheck_gpu.py starts it that way)
needs_legacy = pytest.mark.skipif(not LEGACY, reason="cross-check of a retired kernel generation: runs on the -DLRT_LEGACY library (tests/test_legacy_crosscheck_gpu.py)")
# the product library's modes: the collect & resolve forward with the bucketed replay (4 / 8 / 16 waves per tile, both build sorts), the
# re-tracing backward, the packet kernel both ways.  All SH degrees for the first three, degree 3 for the rest.
PRODUCT_MODES = [{"fwd_mode": 2, "bwd_mode": 3, "defer_colour": 1}, {"fwd_mode": 2, "bwd_mode": 0, "defer_colour": 1}, {"fwd_mode": 0, "bwd_mode": 0},
                 {"fwd_mode": 0, "bwd_mode": 3}, {"fwd_mode": 2, "bwd_mode": 3, "defer_colour": 1, "c4_waves": 4},
                 {"fwd_mode": 2, "bwd_mode": 3, "defer_colour": 1, "c4_waves": 8}, {"fwd_mode": 2, "bwd_mode": 3, "defer_colour": 1, "c4_waves": 16},
                 {"fwd_mode": 2, "bwd_mode": 3, "defer_colour": 1, "own_sort": 1}, {"fwd_mode": 2, "bwd_mode": 3, "defer_colour": 1, "own_sort": 0}]
# ... and what only the cross-check library has: bwd_mode 1 / 2, colours inside the trace kernel
LEGACY_MODES = [{"fwd_mode": 0, "bwd_mode": 2}, {"fwd_mode": 2, "bwd_mode": 1, "defer_colour": 0},
                {"fwd_mode": 2, "bwd_mode": 2, "defer_colour": 0}, {"fwd_mode": 2, "bwd_mode": 2, "defer_colour": 1},
                {"fwd_mode": 2, "bwd_mode": 2, "defer_colour": 0, "c4_waves": 8}, {"fwd_mode": 2, "bwd_mode": 2, "defer_colour": 1, "c4_waves": 8},
                {"fwd_mode": 2, "bwd_mode": 2, "defer_colour": 1, "own_sort": 1}, {"fwd_mode": 2, "bwd_mode": 2, "defer_colour": 1, "own_sort": 0},
                {"fwd_mode": 2, "bwd_mode": 3, "defer_colour": 0}]
DEGS = [(3, (0, 0, 1)), (0, (0, 0, 0)), (1, (0.3, 0.7, 0.2)), (2, (0, 0, 1))]
MODE_CASES = ([(PRODUCT_MODES[0], dg, bg) for dg, bg in DEGS] + [(PRODUCT_MODES[1], *DEGS[0]), (PRODUCT_MODES[1], *DEGS[1]), (PRODUCT_MODES[2], *DEGS[0]),
               (PRODUCT_MODES[2], *DEGS[2])] + [(m, 3, (0, 0, 1)) for m in (PRODUCT_MODES[3], PRODUCT_MODES[4], PRODUCT_MODES[6], PRODUCT_MODES[8])])
if LEGACY:      # the cross-check run: the retired modes (two SH degrees each), and the product's default beside them
    MODE_CASES = [(m, dg, bg) for m in LEGACY_MODES for dg, bg in (DEGS[0], DEGS[2])] + [(PRODUCT_MODES[0], 3, (0, 0, 1))]
GRADS = ("means", "scales", "rotations", "opacities", "shs")


def _mode_id(m):
    return f"fwd{m['fwd_mode']}-bwd{m['bwd_mode']}" + ("-defer" if m.get("defer_colour") else "") + (f"-w{m['c4_waves']}" if m.get("c4_waves") else "") + (f"-sort{m['own_sort']}" if "own_sort" in m else "")


def oracle_run(sc, o, d, deg, bg, dL=None, prec="f32", mod=1.0):
    orc = oracle.Oracle(sc["means"], sc["scales"], sc["rotations"], sc["opacities"], prec, scale_modifier=mod)
    fw = orc.forward(o, d, sc["shs"], deg, bg, stats=True)
    bw = orc.backward(o, d, sc["shs"], deg, bg, fw["out"], dL) if dL is not None else None
    return fw, bw


@pytest.fixture(scope="module")
def s10k():
    sc, o, d = scenes.s10k()
    rng = np.random.default_rng(5)
    dL = scenes.upstream_grad(16, 256)
    dL[..., 4:9] = rng.normal(size=(16, 256, 5)).astype(np.float32) / 4096     # channels 4, 8 ignored (D2); 5-7 live (D3)
    return sc, o, d, dL


@pytest.mark.parametrize("mode,deg,bg", MODE_CASES, ids=[f"{_mode_id(m)}-deg{dg}" for m, dg, _ in MODE_CASES])
def test_s10k_forward_backward_match_oracle(s10k, mode, deg, bg):
    sc, o, d, dL = s10k
    fw, bw = oracle_run(sc, o, d, deg, np.array(bg, np.float32), dL)
    h = run_hip(sc, o, d, deg, bg, dL, opts=mode)
    assert not np.isnan(h["out"]).any()
    assert frac_outside(h["out"], fw["out"], 1e-4) <= 1e-3, "rendered channels outside 1e-4"
    assert rel_l2(h["out"], fw["out"]) < 1e-4
    assert frac_outside(h["accum"], fw["accum"], 1e-4) <= 1e-3
    np.testing.assert_array_equal(h["out"][..., 5:8], 0.0)                    # normals are never accumulated (forward.cu:301-303)
    for k in GRADS:
        ref = bw[k]; got = h["grads"][k].reshape(ref.shape)
        assert frac_outside(got, ref, 1e-3) <= 2e-3, f"d_{k} outside 1e-3"
        assert rel_l2(got, ref) < 1e-3, f"d_{k} L2"
    if deg < 3:
        assert np.all(h["grads"]["shs"][:, (deg + 1) ** 2:, :] == 0)           # inactive SH bands get no gradient


def test_sh_tables_narrower_than_16(s10k):
    """shs (P, M, 3) with M < 16 (a model that has not reached SH degree 3, or never will): the row stride of the SH table
    and of its gradient is M, and only (deg+1)^2 <= M coefficients are read / written."""
    for M, deg in ((1, 0), (4, 1), (9, 2), (16, 1)):
        sc, o, d, dL = s10k
        sc = dict(sc); sc["shs"] = np.ascontiguousarray(sc["shs"][:, :M])
        fw, bw = oracle_run(sc, o, d, deg, scenes.BG_DEFAULT, dL)
        h = run_hip(sc, o, d, deg, scenes.BG_DEFAULT, dL)
        assert rel_l2(h["out"], fw["out"]) < 1e-5 and frac_outside(h["out"], fw["out"], 1e-4) <= 1e-3, (M, deg)
        assert h["grads"]["shs"].shape == (sc["means"].shape[0], M, 3)
        for k in GRADS:
            assert rel_l2(h["grads"][k].reshape(bw[k].shape), bw[k]) < 1e-3, (M, deg, k)
        if (deg + 1) ** 2 < M:                                                 # coefficients above the active degree get no gradient
            assert not np.any(h["grads"]["shs"][:, (deg + 1) ** 2:])


def test_sh_table_wider_than_a_wave_row(s10k):
    """shs (P, 25, 3) -- a model allocated for SH degree 4 and traced at degree <= 3: a gradient row has 10 + 75 = 85 components, more
    than the 64 lanes the bucketed reduction writes a row with, so the backward takes the sorted path (which zero-fills first); every
    element of d_shs must be defined: the active coefficients match the oracle, the others are exactly zero."""
    sc, o, d, dL = s10k
    sc = dict(sc)
    extra = np.random.default_rng(3).normal(0, 0.02, (sc["shs"].shape[0], 9, 3)).astype(np.float32)
    sc["shs"] = np.ascontiguousarray(np.concatenate([sc["shs"], extra], axis=1))
    fw, bw = oracle_run(sc, o, d, 3, scenes.BG_DEFAULT, dL)
    h = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL)
    assert rel_l2(h["out"], fw["out"]) < 1e-5 and frac_outside(h["out"], fw["out"], 1e-4) <= 1e-3
    assert h["grads"]["shs"].shape == (sc["means"].shape[0], 25, 3)
    assert np.all(np.isfinite(h["grads"]["shs"])) and not np.any(h["grads"]["shs"][:, 16:])
    for k in GRADS:
        assert rel_l2(h["grads"][k].reshape(bw[k].shape), bw[k]) < 1e-3, k


def test_build_acceleration_structure_rebuild_0_refits(s10k):
    """`build_acceleration_structure(state, vertices, triangles, rebuild=0)` is the reference's OPTIX_BUILD_OPERATION_UPDATE
    (trace_surfels.cpp:63-73): the next trace of an unchanged number of Gaussians refits the LBVH (lrt_refit) instead of
    rebuilding it; results equal those of a rebuild; a changed P falls back to a full build."""
    from lidar_rt_amd.diff_lidar_tracer import _C
    from tests.hip_util import settings, DEFAULT_OPTS
    sc, o, d, dL = s10k
    r = np.random.default_rng(9)
    tr = Tracer()
    for k, v in DEFAULT_OPTS.items():
        tr.optix_context.set_option(k, v)
    ro, rd = torch.as_tensor(o, device="cuda:0"), torch.as_tensor(d, device="cuda:0")
    vtx = torch.zeros((4, 3), device="cuda:0"); tri = torch.zeros((2, 3), dtype=torch.int32, device="cuda:0")

    def trace(cur, rebuild):
        t = {k: torch.as_tensor(v, device="cuda:0").requires_grad_(True) for k, v in cur.items()}
        tr.build_acceleration_structure(vtx, tri, rebuild=rebuild)
        out, acc = tr(ro, rd, None, t["means"], torch.zeros_like(t["means"]), shs=t["shs"], opacities=t["opacities"],
                      scales=t["scales"], rotations=t["rotations"], tracer_settings=settings(scenes.BG_DEFAULT, 3))
        out.backward(torch.as_tensor(dL, device="cuda:0"))
        return out.detach().cpu().numpy(), {k: t[k].grad.cpu().numpy() for k in GRADS}

    since = lambda: dict(tr.optix_context._since_full)[0]
    trace(sc, 1)
    assert since() == 0
    moved = dict(sc); moved["means"] = (sc["means"] + 0.03 * r.normal(size=sc["means"].shape)).astype(np.float32)
    out, g = trace(moved, 0)
    assert since() == 1                                                        # a refit, not a build
    ref = run_hip(moved, o, d, 3, scenes.BG_DEFAULT, dL)
    assert rel_l2(out, ref["out"]) < 1e-6
    for k in GRADS:
        assert rel_l2(g[k], ref["grads"][k]) < 1e-5, k
    fewer = {k: np.ascontiguousarray(v[:-7]) for k, v in moved.items()}       # P changed: rebuild = 0 cannot update, a full build runs
    out2, _ = trace(fewer, 0)
    ref2 = run_hip(fewer, o, d, 3, scenes.BG_DEFAULT, dL)
    assert rel_l2(out2, ref2["out"]) < 1e-6


def test_side_stream(s10k):
    """Everything the library enqueues (kernels, rocPRIM sorts, fills, the pinned flag copy) goes to the caller's current
    stream: build + forward + backward issued on a side stream give the default-stream results."""
    from tests.hip_util import settings, DEFAULT_OPTS
    sc, o, d, dL = s10k
    ref = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL)
    side = torch.cuda.Stream(device="cuda:0")
    tr = Tracer()
    for k, v in DEFAULT_OPTS.items():
        tr.optix_context.set_option(k, v)
    t = {k: torch.as_tensor(v, device="cuda:0").requires_grad_(True) for k, v in sc.items()}
    ro, rd = torch.as_tensor(o, device="cuda:0"), torch.as_tensor(d, device="cuda:0")
    g = torch.as_tensor(dL, device="cuda:0")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(2):
            for v in t.values():
                v.grad = None
            tr.build_from_gaussians(t["means"], t["scales"], t["rotations"], t["opacities"])
            out, _ = tr(ro, rd, None, t["means"], torch.zeros_like(t["means"]), shs=t["shs"], opacities=t["opacities"],
                        scales=t["scales"], rotations=t["rotations"], tracer_settings=settings(scenes.BG_DEFAULT, 3))
            out.backward(g)
    side.synchronize()
    assert rel_l2(out.detach().cpu().numpy(), ref["out"]) < 1e-6
    for k in GRADS:
        assert rel_l2(t[k].grad.cpu().numpy(), ref["grads"][k]) < 1e-5, k


def test_committed_golden_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "s10k_golden.npz"))
    sc, o, d = scenes.s10k()
    h = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, scenes.upstream_grad(16, 256))
    assert frac_outside(h["out"], g["out"], 1e-4) <= 1e-3 and rel_l2(h["out"], g["out"]) < 1e-4
    for k in GRADS:
        assert rel_l2(h["grads"][k].reshape(g["d_" + k].shape), g["d_" + k]) < 1e-3


def test_scale_modifier(s10k):
    sc, o, d, dL = s10k
    fw, bw = oracle_run(sc, o, d, 3, scenes.BG_DEFAULT, dL, mod=1.3)
    h = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL, mod=1.3)
    assert rel_l2(h["out"], fw["out"]) < 1e-4
    for k in GRADS:
        assert rel_l2(h["grads"][k].reshape(bw[k].shape), bw[k]) < 1e-3


def test_eval_mode_backward_retraces(s10k):
    """module.eval() -> training=False -> no hit record -> the backward re-traces like the reference."""
    sc, o, d, dL = s10k
    fw, bw = oracle_run(sc, o, d, 3, scenes.BG_DEFAULT, dL)
    h = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL, training=False)
    for k in GRADS:
        assert rel_l2(h["grads"][k].reshape(bw[k].shape), bw[k]) < 1e-3


def test_retracing_backward_right_behind_a_build_sees_a_finished_tree():
    """a fused build leaves the tree levels >= 4 to the NEXT forward's prologue.  build -> backward with no forward in
    between (legal in the C ABI; the re-tracing backward walks the tree) must finish them itself: the sequence build(A), forward(A),
    build(B), re-tracing backward(B) must give the gradients of build(B), forward(B), backward(B).  40,000 Gaussians: five tree levels."""
    from lidar_rt_amd.parallel import HipBackend
    dev = torch.device("cuda:0")
    scB = scenes.make_scene(40_000, seed=5, radius_scale=0.4)
    scA = {k: v.copy() for k, v in scB.items()}
    scA["means"] = scA["means"] + np.array([7.0, -5.0, 0.5], np.float32)          # another tree top: stale boxes would cull the wrong space
    o, d = scenes.kitti_rays(16, 128)
    dL = scenes.upstream_grad(16, 128)
    tB = {k: torch.as_tensor(v, device=dev) for k, v in scB.items()}
    tA = {k: torch.as_tensor(v, device=dev) for k, v in scA.items()}
    ro, rd, g_up = torch.as_tensor(o, device=dev), torch.as_tensor(d, device=dev), torch.as_tensor(dL, device=dev)
    bg = torch.as_tensor(scenes.BG_DEFAULT, device=dev)
    be = HipBackend()
    for k, v in DEFAULT_OPTS_.items():
        be.state.set_option(k, v)
    be.state.set_option("bwd_mode", 0)                                              # the backward re-traces like the reference
    argsB = (tB["means"], tB["scales"], tB["rotations"], tB["opacities"], tB["shs"], 3, bg)
    argsA = (tA["means"], tA["scales"], tA["rotations"], tA["opacities"], tA["shs"], 3, bg)
    be.build(*argsB[:4])
    outB, _ = be.forward(ro, rd, *argsB)
    ref = {k: v.clone() for k, v in be.backward(ro, rd, *argsB, outB, g_up).items()}
    be.build(*argsA[:4]); be.forward(ro, rd, *argsA)
    be.build(*argsB[:4])                                                            # no forward behind this build
    got = be.backward(ro, rd, *argsB, outB, g_up)
    torch.cuda.synchronize()
    be.state.check(dev, wait=True)
    be.state.set_option("bwd_mode", 3)
    for k in ("means", "scales", "rotations", "opacities", "shs"):
        assert float(ref[k].abs().sum()) > 0
        assert rel_l2(got[k].cpu().numpy(), ref[k].cpu().numpy()) < 2e-3, k          # another Morton box (carried from build A) = another tree: the float additions and exact-tie orders differ; a stale tree top gives O(1)


# ---------------------------------------------------------------------------------- known answers / edge cases
def _facing(xs, ops, sh_dc=(0.3, 0.1, -0.2)):
    n = len(xs)
    q = np.tile(np.array([np.cos(np.pi / 4), 0.0, np.sin(np.pi / 4), 0.0], np.float32), (n, 1))
    sc = {"means": np.stack([np.array(xs), np.zeros(n), np.zeros(n)], 1).astype(np.float32),
          "scales": np.full((n, 2), 0.5, np.float32), "rotations": q,
          "opacities": np.array(ops, np.float32)[:, None], "shs": np.zeros((n, 16, 3), np.float32)}
    sc["shs"][:, 0, :] = np.array(sh_dc, np.float32) / 0.28209479177387814
    return sc


@pytest.mark.parametrize("mode", [PRODUCT_MODES[0], PRODUCT_MODES[2]] + ([LEGACY_MODES[2]] if LEGACY else []), ids=["collect4-defer", "packet"] + (["collect4"] if LEGACY else []))
def test_known_answers(mode):
    o = np.zeros((1, 1, 3), np.float32); d = np.array([[[1.0, 0, 0]]], np.float32)
    # miss -> background
    h = run_hip(_facing([5.0], [0.5]), o, -d, 0, (0, 0, 1), opts=mode)
    np.testing.assert_allclose(h["out"][0, 0], [0, 0, 1, 0, 0, 0, 0, 0, 1], atol=1e-7)
    # single hit
    h = run_hip(_facing([5.0], [0.5]), o, d, 0, (0, 0, 1), opts=mode)
    np.testing.assert_allclose(h["out"][0, 0], [0.4, 0.3, 0.15 + 0.5, 2.5, 0.5, 0, 0, 0, 0.5], rtol=1e-5, atol=1e-6)
    # hit closer than 0.2 m is skipped (forward.cu:214)
    h = run_hip(_facing([0.1, 5.0], [0.5, 0.5]), o, d, 0, (0, 0, 0), opts=mode)
    np.testing.assert_allclose(h["out"][0, 0, 3], 2.5, rtol=1e-5)
    # 40 hits: crosses two chunk boundaries
    xs = list(np.linspace(2.0, 21.0, 40))
    h = run_hip(_facing(xs, [0.05] * 40), o, d, 0, (0, 0, 0), opts=mode)
    T = 0.95 ** 40
    np.testing.assert_allclose(h["out"][0, 0, 8], T, rtol=2e-5)
    # a hit 5e-6 behind the 16th is dropped by the restart epsilon (forward.cu:282-291): 17 of 18 composited
    xs = list(np.linspace(2.0, 9.5, 16)) + [9.5 + 5e-6, 12.0]
    fw, _ = oracle_run(_facing(xs, [0.05] * 18), o, d, 0, np.zeros(3, np.float32))
    h = run_hip(_facing(xs, [0.05] * 18), o, d, 0, (0, 0, 0), opts=mode)
    np.testing.assert_allclose(h["out"][0, 0], fw["out"][0, 0], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(h["out"][0, 0, 8], 0.95 ** 17, rtol=2e-5)
    # opaque wall stops the ray: the stopping hit is not composited (forward.cu:253-257)
    h = run_hip(_facing([3.0, 4.0, 5.0, 6.0], [0.999] * 4), o, d, 0, (0, 0, 0), opts=mode)
    assert h["out"][0, 0, 8] >= 1e-4 * (1 - 1e-5)


def test_empty_scene_and_unhittable_gaussians():
    o, d = scenes.kitti_rays(5, 37)                                           # ragged: not a multiple of any tile
    empty = {"means": np.zeros((0, 3), np.float32), "scales": np.zeros((0, 2), np.float32),
             "rotations": np.zeros((0, 4), np.float32), "opacities": np.zeros((0, 1), np.float32),
             "shs": np.zeros((0, 16, 3), np.float32)}
    h = run_hip(empty, o, d, 3, (0.1, 0.2, 0.3))
    np.testing.assert_allclose(h["out"][..., :3], np.broadcast_to([0.1, 0.2, 0.3], (5, 37, 3)), atol=1e-7)
    np.testing.assert_array_equal(h["out"][..., 8], 1.0)
    sc = scenes.make_scene(3000, seed=2, radius_scale=0.3)
    sc["opacities"][::3] = 0.003                                               # <= 1/255: NaN quads in the reference -> unhittable
    fw, bw = oracle_run(sc, o, d, 3, scenes.BG_DEFAULT, scenes.upstream_grad(5, 37))
    h = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, scenes.upstream_grad(5, 37))
    assert rel_l2(h["out"], fw["out"]) < 1e-4
    assert np.all(h["accum"][::3] == 0) and np.all(h["grads"]["means"][::3] == 0)


def test_forward_modes_agree_and_backward_is_deterministic(s10k):
    sc, o, d, dL = s10k
    bm = 2 if LEGACY else 3
    a = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL, opts={"fwd_mode": 2, "bwd_mode": bm})
    b = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL, opts={"fwd_mode": 0, "bwd_mode": bm})
    assert rel_l2(a["out"], b["out"]) < 2e-5
    c = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL, opts={"fwd_mode": 2, "bwd_mode": bm})
    np.testing.assert_array_equal(a["out"], c["out"])                          # forward: hits ordered by (t, gidx), no atomics in the image
    for k in GRADS:
        # bucketed reduction: the order of the additions inside a Gaussian's run is the arrival order of integer LDS atomics (as with the
        # reference's float atomics): run-to-run equal to rounding.  Sorted reduction (cross-check library): identical except where a
        # Gaussian's hits span > 2 reduction chunks
        assert rel_l2(a["grads"][k], c["grads"][k]) < (1e-7 if LEGACY else 1e-6)
        if LEGACY:
            assert (a["grads"][k] != c["grads"][k]).mean() < 1e-3
    if LEGACY:
        e = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL, opts={"fwd_mode": 2, "defer_colour": 0})     # colour in the trace kernel: other summation order
        assert rel_l2(e["out"], a["out"]) < 1e-6


def test_own_radix_sort_and_rocprim_give_the_same_results():
    """lrt_radix.inc (own onesweep: epoch-tagged look-back words, ticket counter, no fills) against rocPRIM for both sorts of a step,
    on sizes with one tile, a few tiles and hundreds of tiles per pass; twice in a row (epochs / tickets carry over)."""
    for P, (H, W) in ((3_000, (8, 64)), (40_000, (16, 256)), (300_000, (32, 512))):
        sc = scenes.make_scene(P, radius_scale=0.5 if P > 100_000 else 0.25)
        o, d = scenes.kitti_rays(H, W)
        dL = scenes.upstream_grad(H, W)
        ref = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL, opts={"own_sort": 0})
        for rep in range(2):
            got = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL, opts={"own_sort": 1})
            np.testing.assert_array_equal(got["out"], ref["out"])              # the image does not depend on the order inside a Morton cell
            for k in GRADS:                                                    # both sorts are stable: same runs, same summation order --
                assert rel_l2(got["grads"][k], ref["grads"][k]) < 5e-7, (P, k)   # up to the float atomics of runs that span two waves (1.3e-7 seen)


def _read_build(tr, which, n_bytes):
    """Internal buffer of the current build (lrt_debug_read: 0 sorted order, 1 records, 2 SoA nodes)."""
    import ctypes as C
    st = tr.optix_context
    _, h = st.handle(torch.device("cuda:0"))
    buf = np.empty(n_bytes // 4, np.uint32)
    st._lib.lrt_debug_read.restype = C.c_longlong
    got = st._lib.lrt_debug_read(h, which, buf.ctypes.data_as(C.c_void_p), C.c_longlong(buf.nbytes), None)
    return buf[:min(got, n_bytes) // 4]


@needs_legacy
@pytest.mark.parametrize("P", [5, 64, 65, 513, 4097, 40_000, 300_000] if LEGACY else [5])
def test_fused_tree_build_equals_the_level_by_level_one(P):
    """k_make_tree (records + levels 1-3 per workgroup, boxes in LDS) + k_tree_top, and k_morton's fused digit histograms, against the
    round-1..3 build (k_make_records, k_level1, one k_upper per level, k_rs_hist): same sorted order, bit-identical records and -- in
    the 50 meaningful words of every node -- bit-identical trees, for tree depths 1 .. 6 and partial last nodes on every level."""
    from tests.hip_util import DEFAULT_OPTS
    sc = scenes.make_scene(P, seed=17 + P, radius_scale=0.5 if P > 100_000 else 0.25)
    t = {k: torch.as_tensor(v, device="cuda:0") for k, v in sc.items()}
    got = {}
    for fused in (0, 1, 2):
        tr = Tracer()
        for k, v in {**DEFAULT_OPTS, "own_sort": 1, "fused_tree": fused, "fused_hist": min(fused, 1)}.items():
            tr.optix_context.set_option(k, v)
        for rep in range(2):                                                   # twice: the histogram's zero-on-exit invariant must hold
            tr.build_from_gaussians(t["means"], t["scales"], t["rotations"], t["opacities"])
        torch.cuda.synchronize()
        n_leaves = (P + 7) // 8
        n_nodes, c = 0, n_leaves
        while True:
            c = max((c + 7) // 8, 1); n_nodes += c
            if c == 1:
                break
        got[fused] = (_read_build(tr, 0, P * 4), _read_build(tr, 1, P * 64), _read_build(tr, 2, n_nodes * 256).reshape(-1, 64)[:, :50])
        for k in ("fused_tree", "fused_hist"):
            tr.optix_context.set_option(k, 1)
    for fused in (1, 2):
        for part in range(3):
            np.testing.assert_array_equal(got[0][part], got[fused][part])


def test_bucketed_backward_against_the_sorted_one_on_awkward_index_layouts():
    """bwd_mode 3 (count / scatter / LDS sort per bucket / flat reduction) against bwd_mode 2 (key sort + segmented reduction) where the
    bucket machinery is stressed: Gaussian indices ordered by distance from the sensor (the near, heavily hit Gaussians share a few
    buckets: tens of thousands of records in one bucket, empty buckets elsewhere), P not a multiple of the bucket size, P smaller than
    one bucket, a single ray column, an image smaller than one ray group."""
    cases = []
    sc = scenes.make_scene(30_011, seed=11, radius_scale=0.25)
    order = np.argsort(np.linalg.norm(sc["means"], axis=1), kind="stable")
    cases.append(({k: np.ascontiguousarray(v[order]) for k, v in sc.items()}, (16, 256)))
    cases.append((scenes.make_scene(19, seed=12, radius_scale=0.1), (8, 64)))
    cases.append((scenes.make_scene(5_000, seed=13, radius_scale=0.2), (64, 1)))
    cases.append((scenes.make_scene(700, seed=14, radius_scale=0.1), (2, 3)))
    for sc, (H, W) in cases:
        o, d = scenes.kitti_rays(H, W)
        dL = scenes.upstream_grad(H, W)
        ref = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL, opts={"bwd_mode": 2 if LEGACY else 0})      # product library: against the re-tracing backward's atomics
        for rep in range(2):                                                   # twice: the buffers of the first call are reused
            got = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL, opts={"bwd_mode": 3})
            np.testing.assert_array_equal(got["out"], ref["out"])
            for k in GRADS:
                assert np.isfinite(got["grads"][k]).all()
                assert rel_l2(got["grads"][k], ref["grads"][k]) < (5e-6 if LEGACY else 5e-5), (len(sc["means"]), H, W, k)     # other summation order inside a Gaussian's run (product: the packet kernel's own hit distances too)
                assert ((got["grads"][k] != 0) == (ref["grads"][k] != 0)).mean() > 0.9999              # rows of untouched Gaussians are zeros, not leftovers


# ---------------------------------------------------------------------------------- larger scenes, statistical parity
def test_s200k_matches_oracle_within_noise_floor():
    sc = scenes.make_scene(200_000, radius_scale=0.5)
    o, d = scenes.kitti_rays(32, 512)
    dL = scenes.upstream_grad(32, 512)
    f32 = oracle_run(sc, o, d, 3, scenes.BG_DEFAULT, dL)
    f64 = oracle_run(sc, o, d, 3, scenes.BG_DEFAULT, dL, prec="f64")
    h = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL)
    assert frac_outside(h["out"], f32[0]["out"], 5e-2) <= 2e-4                 # no gross outliers
    # the unmasked table: floor-relative bounds against the fp32 oracle asserted, the fp64-arbitrated ratios recorded ...
    parity_report("s200k", h, f32, f64, f64_k=None, extra={"config": "S200k: 200,000 Gaussians, 32x512 rays", "rays": [32, 512], "gaussians": 200_000})
    # ... and the claim itself -- (1.1, 1.25), no factor, no 2 x tol clause -- asserted with the threshold events named, counted and certified
    from tests.event_gate import event_masked_gate
    event_masked_gate("s200k", sc, o, d, 3, scenes.BG_DEFAULT, dL, f32[0], f64[0])


def test_s1m_full_size_parity_and_invariants(golden_dir):
    """BASELINE configs[1] at full size: oracle comparison + size-independent properties."""
    import json
    sc, o, d = scenes.s1m()
    H, W = o.shape[:2]
    dL = scenes.upstream_grad(H, W)
    h = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL)
    stats = json.load(open(os.path.join(golden_dir, "s1m_stats.json")))
    np.testing.assert_allclose(h["out"].reshape(-1, 9).mean(0), stats["out_channel_means"], rtol=2e-4, atol=1e-6)
    # energy conservation per ray: sum of weights + final transmittance = 1
    np.testing.assert_allclose(h["out"][..., 4] + h["out"][..., 8], 1.0, atol=2e-5)
    # accum is the transpose-sum of the same weights
    np.testing.assert_allclose(h["accum"].sum(dtype=np.float64), h["out"][..., 4].sum(dtype=np.float64), rtol=1e-5)
    fw, bw = oracle_run(sc, o, d, 3, scenes.BG_DEFAULT, dL)
    f64 = oracle_run(sc, o, d, 3, scenes.BG_DEFAULT, dL, prec="f64")
    assert frac_outside(h["out"], fw["out"], 5e-2) <= 2e-4                     # no gross outliers
    # every channel and every gradient within k x the fp32-vs-fp64 oracle floor (tests/hip_util.py: FLOOR_K); numbers -> gpurun_out/parity/s1m.json
    parity_report("s1m", h, (fw, bw), f64, extra={"config": "BASELINE configs[1]: S1M, 1,000,000 Gaussians, 64x2048 rays", "rays": [H, W],
                                                  "gaussians": 1_000_000, "C_mean": float(fw["n_cand"].mean()), "K_mean": float(fw["n_comp"].mean())})
    # the same claim with the threshold events named, counted (no more than the fp32 oracle's own), certified by the brute-force float64 restatement and
    # masked (tests/event_gate.py); and once with the fp64 re-ordering of sub-2-ulp neighbours switched off, for the record: which side of the floor it moves
    from tests.event_gate import event_masked_gate
    rec = event_masked_gate("s1m", sc, o, d, 3, scenes.BG_DEFAULT, dL, fw, f64[0])
    rec0 = event_masked_gate("s1m_no_refine", sc, o, d, 3, scenes.BG_DEFAULT, dL, fw, f64[0], opts={"refine_ties": 0}, count_only=True)
    assert rec0["event_rays"]["hip"] >= rec["event_rays"]["hip"], (rec0["event_rays"], rec["event_rays"])     # the re-ordering removes order events, it adds none
    # permutation invariance: the input order of the Gaussians must not matter
    perm = np.random.default_rng(0).permutation(sc["means"].shape[0])
    scp = {k: np.ascontiguousarray(v[perm]) for k, v in sc.items()}
    hp = run_hip(scp, o, d, 3, scenes.BG_DEFAULT, dL)
    assert frac_outside(hp["out"], h["out"], 1e-4) <= 5e-3 and rel_l2(hp["out"], h["out"]) < 2e-3
    assert rel_l2(hp["grads"]["means"], h["grads"]["means"][perm]) < 2e-2
    # backward is linear in the upstream gradient
    dL2 = scenes.upstream_grad(H, W, seed=99)
    h2 = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL2)
    h3 = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, 2.0 * dL - 0.5 * dL2)
    for k in ("means", "opacities", "shs"):
        lin = 2.0 * h["grads"][k] - 0.5 * h2["grads"][k]
        assert rel_l2(h3["grads"][k], lin) < 1e-4


def test_deferred_colour_beyond_the_hit_record(s10k):
    """More composited hits than the per-ray record holds: the deferred colour pass takes the rest from the overflow
    list (forward stays exact) and the backward falls back to re-tracing like the reference."""
    sc, o, d, dL = s10k
    a = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL, opts={"fwd_mode": 2, "defer_colour": 0} if LEGACY else {"fwd_mode": 2})      # (product: the record that holds every hit)
    for fm in (2,):
        b = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL, opts={"fwd_mode": fm, "defer_colour": 1, "hit_cap": 4})
        assert rel_l2(b["out"], a["out"]) < 1e-5 and frac_outside(b["out"], a["out"], 1e-4) <= 1e-3
        for k in GRADS:
            assert rel_l2(b["grads"][k], a["grads"][k]) < 1e-3, k


def test_hit_record_grows_after_an_overflow(s10k):
    """A frame whose rays composite more hits than the record holds is re-traced in the backward (slow); the capacity then
    doubles, so the following frames replay the record again.  The gradients are the same all along."""
    from tests.hip_util import settings, DEFAULT_OPTS
    sc, o, d, dL = s10k
    ref = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL)
    tr = Tracer()
    for k, v in {**DEFAULT_OPTS, "hit_cap": 4}.items():
        tr.optix_context.set_option(k, v)
    caps = []
    for it in range(7):
        t = {k: torch.as_tensor(v, device="cuda:0").requires_grad_(True) for k, v in sc.items()}
        ro, rd = torch.as_tensor(o, device="cuda:0"), torch.as_tensor(d, device="cuda:0")
        tr.build_from_gaussians(t["means"], t["scales"], t["rotations"], t["opacities"])
        out, _ = tr(ro, rd, None, t["means"], torch.zeros_like(t["means"]), shs=t["shs"], opacities=t["opacities"],
                    scales=t["scales"], rotations=t["rotations"], tracer_settings=settings(scenes.BG_DEFAULT, 3))
        out.backward(torch.as_tensor(dL, device="cuda:0"))
        caps.append(tr.optix_context.get_option("hit_cap", "cuda:0"))
        for k in GRADS:
            assert rel_l2(t[k].grad.cpu().numpy(), ref["grads"][k]) < 1e-3, (it, k)
    assert caps[0] == 8 and caps[-1] == caps[-2] and 16 <= caps[-1] <= 256, caps    # grew, then settled
    tr.optix_context.set_option("hit_cap", 256)


def test_dense_translucent_scene_with_4_8_and_16_waves_per_tile():
    """every wave count of the trace kernel on the scene that found the round-3 spill fault (hundreds of candidates per ray,
    list and queue overflows, dozens of slabs with done rays, the per-ray node tests after a queue overflow): same image and same gradients
    whatever the number of waves per tile, and the oracle's image."""
    sc, o, d = scenes.dense_translucent()
    dL = scenes.upstream_grad(4, 48, seed=2)
    fw, _ = oracle_run(sc, o, d, 3, scenes.BG_DEFAULT, dL)
    res = {nw: run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL, opts={"c4_waves": nw}) for nw in (4, 8, 16)}
    for nw in (8, 16):
        assert rel_l2(res[nw]["out"], res[4]["out"]) < 1e-6, nw
        for k in GRADS:
            assert rel_l2(res[nw]["grads"][k], res[4]["grads"][k]) < 1e-4, (nw, k)
    assert frac_outside(res[4]["out"], fw["out"], 1e-3) <= 1e-2                 # (the knife-edge rays of this scene are examined by the test below)


def test_dense_translucent_scene_overflows_every_capacity_once():
    """Many large Gaussians of opacity 0.03: ~130 candidates and ~100 composited hits per ray on average, up to ~600 / ~500.  The
    per-slab candidate lists (256) overflow and the slabs are halved; frame 0 exceeds the hit record (256 per ray): its backward
    re-traces and the record doubles; frame 1 exceeds the dense key list (64 per ray on average): its backward uses the atomic
    replay and the key list doubles; frame 2 takes the sorted reduction.  All three agree with each other and with the oracle."""
    from tests.hip_util import settings, DEFAULT_OPTS
    sc, o, d = scenes.dense_translucent()
    dL = scenes.upstream_grad(4, 48, seed=2)
    fw, bw = oracle_run(sc, o, d, 3, scenes.BG_DEFAULT, dL)
    assert fw["n_comp"].mean() > 64 and fw["n_comp"].max() > 256 and fw["n_cand"].max() > 256     # the scene does what it is for
    tr = Tracer()
    for k, v in DEFAULT_OPTS.items():
        tr.optix_context.set_option(k, v)
    frames = []
    for it in range(3):
        t = {k: torch.as_tensor(v, device="cuda:0").requires_grad_(True) for k, v in sc.items()}
        ro, rd = torch.as_tensor(o, device="cuda:0"), torch.as_tensor(d, device="cuda:0")
        tr.build_from_gaussians(t["means"], t["scales"], t["rotations"], t["opacities"])
        out, _ = tr(ro, rd, None, t["means"], torch.zeros_like(t["means"]), shs=t["shs"], opacities=t["opacities"],
                    scales=t["scales"], rotations=t["rotations"], tracer_settings=settings(scenes.BG_DEFAULT, 3))
        out.backward(torch.as_tensor(dL, device="cuda:0"))
        frames.append({"out": out.detach().cpu().numpy(), "cap": tr.optix_context.get_option("hit_cap", "cuda:0"),
                       **{k: t[k].grad.cpu().numpy() for k in GRADS}})
    assert [f["cap"] for f in frames] == [512, 512, 512]
    # ~100 hits per ray a few mm apart and a chunk boundary every 16: here and there two candidates straddle the reference's
    # restart epsilon (t16 + 1e-5) within float32 rounding (tests/tools/dense_arbiter.py shows one such ray: 1.0e-5 apart at a
    # boundary), and implementations with other FMA contractions decide differently -> one hit, i.e. up to a few per cent of
    # one ray.  Hence statistical bounds, as for the large scenes.
    assert rel_l2(frames[0]["out"], fw["out"]) < 1e-3 and frac_outside(frames[0]["out"], fw["out"], 1e-4) <= 2e-2
    f64 = oracle_run(sc, o, d, 3, scenes.BG_DEFAULT, dL, prec="f64")
    # the unmasked table, for the record (192 rays: one ray is 5e-3 of the image)
    parity_report("dense_translucent", {"out": frames[2]["out"], "accum": fw["accum"], "grads": {k: frames[2][k] for k in GRADS}}, (fw, bw), f64, check=False,
                  extra={"config": "dense translucent stress scene: 192 rays, ~130 candidates / ~100 composited hits per ray (up to ~600 / ~500)",
                         "rays": [4, 48], "gaussians": int(sc["means"].shape[0]), "note": "accum row = oracle against itself (not captured per frame)"})
    # ASSERTED (round 4): the rays on which the HIP image leaves the oracle's are few (<= 2 of 192), each is CERTIFIED as a restart-epsilon
    # knife edge by the brute-force float64 restatement (a candidate within 5 ulp of t16 + 1e-5 at one of the ray's restarts: which side it
    # falls on is decided by the last bit of t), and with those rays masked (upstream gradient zero, image rows taken from the oracle) every
    # channel and every gradient is within the floor-relative gates of the large scenes
    from oracle.bruteforce import QuadScene
    H_, W_ = o.shape[:2]
    ho, fo = frames[2]["out"].reshape(-1, 9), fw["out"].reshape(-1, 9)
    ch_ = [0, 1, 2, 3, 4, 8]                                                  # (channels 5-7, the normals, are all-zero: no 0 / 0)
    scale = np.maximum(np.abs(fo[:, ch_]), 1e-3 * np.abs(fo[:, ch_]).max(0, keepdims=True))
    ray_err = (np.abs(ho[:, ch_] - fo[:, ch_]) / scale).max(1)
    bad = np.nonzero(ray_err > 1e-4)[0]
    assert len(bad) <= 2, (len(bad), ray_err[bad])
    qs = QuadScene(sc["means"], sc["scales"], sc["rotations"], sc["opacities"])
    for r in bad:
        g_, tt, al = qs.candidates(o.reshape(-1, 3)[r].astype(np.float64), d.reshape(-1, 3)[r].astype(np.float64))
        margins = []                                                          # the reference's loop: distance of the nearest candidate to every restart point
        T, start, i = 1.0, -1.0, 0
        while True:
            while i < len(g_) and not (tt[i] > start):
                i += 1
            chunk = list(range(i, min(i + 16, len(g_)))); i += len(chunk)
            stop = False
            for k in chunk:
                if tt[k] < 0.2 or al[k] < 1 / 255:
                    continue
                if T * (1 - al[k]) < 1e-4:
                    stop = True; break
                T *= 1 - al[k]
            if stop or len(chunk) < 16:
                break
            start = tt[chunk[-1]] + 1e-5
            margins.append(float(np.abs(tt - start).min() / start))
        assert margins and min(margins) < 6e-7, (int(r), float(ray_err[r]), sorted(margins)[:3])     # ~5 ulp of t: a knife edge
    dLm = dL.copy().reshape(-1, 9); dLm[bad] = 0.0; dLm = dLm.reshape(dL.shape)
    fwm, bwm = oracle_run(sc, o, d, 3, scenes.BG_DEFAULT, dLm)
    f64m = oracle_run(sc, o, d, 3, scenes.BG_DEFAULT, dLm, prec="f64")
    t = {k: torch.as_tensor(v, device="cuda:0").requires_grad_(True) for k, v in sc.items()}
    tr.build_from_gaussians(t["means"], t["scales"], t["rotations"], t["opacities"])
    outm, _ = tr(torch.as_tensor(o, device="cuda:0"), torch.as_tensor(d, device="cuda:0"), None, t["means"], torch.zeros_like(t["means"]), shs=t["shs"],
                 opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"], tracer_settings=settings(scenes.BG_DEFAULT, 3))
    outm.backward(torch.as_tensor(dLm, device="cuda:0"))
    hm = outm.detach().cpu().numpy().reshape(-1, 9).copy(); hm[bad] = fwm["out"].reshape(-1, 9)[bad]
    parity_report("dense_translucent_masked", {"out": hm.reshape(H_, W_, 9), "accum": fwm["accum"], "grads": {k: t[k].grad.cpu().numpy() for k in GRADS}},
                  (fwm, bwm), f64m, check=True,
                  extra={"config": "dense translucent stress scene with the certified knife-edge rays masked", "masked_rays": [int(r) for r in bad],
                         "rays": [4, 48], "gaussians": int(sc["means"].shape[0]), "note": "accum row = oracle against itself"})
    for k in GRADS:
        ref = bw[k]
        for f in frames:
            assert rel_l2(f[k].reshape(ref.shape), ref) < 2e-2 and frac_outside(f[k].reshape(ref.shape), ref, 1e-3) <= 3e-2, k
        for f in frames[1:]:
            assert rel_l2(f[k], frames[0][k]) < 2e-2, k
    if LEGACY:
        assert rel_l2(frames[2]["shs"], frames[1]["shs"]) < 1e-5               # atomic replay vs sorted reduction: same hits
    tr.optix_context.set_option("hit_cap", 256)


def test_unrecoverable_overflow_is_reported_loudly(s10k):
    """A trace that cannot complete (here: a queue limit no slab width can satisfy) must raise, in eval mode right
    away (no backward follows) and in training mode at the backward."""
    from lidar_rt_amd._capi import LrtError
    sc, o, d, dL = s10k
    tr = Tracer()
    with pytest.raises(LrtError, match="internal overflow"):
        run_hip(sc, o, d, 3, scenes.BG_DEFAULT, opts={"c4_queue_limit": 136, "c4_waves": 4}, tracer=tr, training=False)
    tr.train()
    with pytest.raises(LrtError, match="internal overflow"):
        run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL, opts={"c4_queue_limit": 136, "c4_waves": 4}, tracer=tr)
    h = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL, tracer=tr)                 # the state recovers with sane options
    assert not np.isnan(h["out"]).any()


def test_queue_overflow_falls_back_to_narrower_slabs(s10k):
    """k_fwd_cr4: a BVH queue that would overflow makes the tile halve its depth slab and collect again; with an
    artificially small limit this happens on many tiles and the result must not change."""
    sc, o, d, dL = s10k
    a = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL)
    b = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL, opts={"c4_queue_limit": 330, "c4_waves": 4})    # 74 entries before a round's 256 possible appends
    assert rel_l2(b["out"], a["out"]) < 1e-6 and frac_outside(b["out"], a["out"], 1e-5) <= 1e-4
    for k in GRADS:
        assert rel_l2(b["grads"][k], a["grads"][k]) < 1e-5, k


def test_coincident_gaussians_are_ordered_by_index():
    """Every Gaussian twice (same geometry, other opacity and SH): each ray gets pairs of hits with bit-identical t.  The rank
    sort of k_fwd_cr4 sees them as a rank total below n(n-1)/2 and adds the gidx tie-break in a second pass; the result must
    be the (t, gidx) order of the one-wave kernel, whatever the waves per tile and the order of collection."""
    base = scenes.make_scene(1500, seed=11, radius_scale=0.3)
    r = np.random.default_rng(12)
    sc = {k: np.concatenate([v, v], 0) for k, v in base.items()}
    n = base["means"].shape[0]
    sc["opacities"][n:] = np.clip(r.uniform(0.05, 0.9, (n, 1)), 0.01, 0.99).astype(np.float32)
    sc["shs"][n:] = (0.5 * r.normal(size=base["shs"].shape)).astype(np.float32)
    perm = r.permutation(2 * n)                                                # the two copies far apart in index
    sc = {k: np.ascontiguousarray(v[perm]) for k, v in sc.items()}
    o, d = scenes.kitti_rays(8, 96)
    dL = scenes.upstream_grad(8, 96, seed=3)
    # reference with the (t, gidx) order: the brute-force float64 restatement of the raygen loop sorts stably by t, i.e. equal t
    # in index order (oracle/bruteforce.py); 40 rays spread over the image.  (The K-buffer kernels keep equal t in arrival order.)
    from oracle.bruteforce import QuadScene, raygen_loop, sh_colour
    qs = QuadScene(sc["means"], sc["scales"], sc["rotations"], sc["opacities"])
    H, W = o.shape[:2]
    rays = np.random.default_rng(3).choice(H * W, 40, replace=False)
    ref = np.zeros((len(rays), 9))
    for i, r in enumerate(rays):
        oo, dd = o.reshape(-1, 3)[r], d.reshape(-1, 3)[r]
        g, tt, al = qs.candidates(oo, dd)
        comp, T, _, _ = raygen_loop(g, tt, al)
        for gi, ti, wi in comp:
            ref[i, 0:3] += wi * sh_colour(sc["shs"][gi], dd, 3); ref[i, 3] += wi * ti; ref[i, 4] += wi
        ref[i, 0:3] += T * scenes.BG_DEFAULT; ref[i, 8] = T
    assert (ref[:, 4] > 0.05).sum() > 12                                       # most sampled rays composite coincident pairs
    for nw in (4, 8, 16):
        a = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL, opts={"fwd_mode": 2, "c4_waves": nw})
        b = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL, opts={"fwd_mode": 2, "c4_waves": nw, "slab0_mm": 3000})
        assert rel_l2(a["out"], b["out"]) < 1e-6, nw                           # other slabs, other collection order: same image
        got = a["out"].reshape(-1, 9)[rays]
        assert rel_l2(got, ref) < 2e-5, nw
        for k in GRADS:
            assert rel_l2(a["grads"][k], b["grads"][k]) < 1e-4, (nw, k)


@pytest.mark.parametrize("seed", [101, 102, 103, 104])
def test_randomised_scenes_cr4_against_the_legacy_packet_kernel(seed):
    """Two independent implementations (k_fwd_cr4 + sorted-reduction backward vs the K-buffer packet kernel + re-trace
    backward) on random scenes, ray grids, SH degrees, first-slab widths, waves per tile and record capacities."""
    r = np.random.default_rng(seed)
    P = int(r.integers(500, 20000))
    sc = scenes.make_scene(P, seed=seed, radius_scale=float(r.uniform(0.15, 0.4)))
    H, W = int(r.integers(3, 24)), int(r.integers(17, 300))
    o, d = scenes.kitti_rays(H, W)
    if seed % 2:                                                               # a moved, tilted sensor: origins off the centre
        o = o + np.array([0.3, -0.2, 0.1], np.float32)
        c, s_ = np.cos(0.1), np.sin(0.1)
        d = (d.reshape(-1, 3) @ np.array([[c, 0, s_], [0, 1, 0], [-s_, 0, c]], np.float32).T).reshape(H, W, 3).astype(np.float32)
    deg = int(r.integers(0, 4))
    dL = scenes.upstream_grad(H, W, seed=seed)
    opts = {"fwd_mode": 2, "bwd_mode": 2 if LEGACY else 3, "slab0_mm": int(r.choice([2000, 8000, 24000, 100000])), "c4_waves": int(r.choice([4, 8])),
            "hit_cap": int(r.choice([16, 64, 256]))}
    a = run_hip(sc, o, d, deg, scenes.BG_DEFAULT, dL, opts=opts)
    b = run_hip(sc, o, d, deg, scenes.BG_DEFAULT, dL, opts={"fwd_mode": 0, "bwd_mode": 0})
    # the two kernels evaluate the ray/quad intersection with differently contracted FMAs (packed fp32 in k_fwd_cr4): ulps in
    # alpha flip a 1/255 or 0.99 threshold for an isolated hit now and then; one such flip in a small image is ~1e-4 in L2
    assert rel_l2(a["out"], b["out"]) < 5e-4 and frac_outside(a["out"], b["out"], 1e-4) <= 2e-3
    # a hit whose alpha or transmittance sits on a threshold is composited by one implementation and not by the other (1-ulp
    # differences in t): single weights differ, so the per-Gaussian quantities are compared statistically
    assert rel_l2(a["accum"], b["accum"]) < 2e-3 and frac_outside(a["accum"], b["accum"], 1e-3) <= 1e-2
    for k in GRADS:
        assert rel_l2(a["grads"][k], b["grads"][k]) < 5e-3 and frac_outside(a["grads"][k], b["grads"][k], 1e-3) <= 2e-2, k


@pytest.mark.parametrize("case", ["rolling_shutter_origins", "scattered_directions", "mixed"])
def test_tiles_without_a_ray_pyramid_take_the_per_ray_path(case):
    """Round 5: the forward culls quads and child boxes against the PYRAMID of a tile's 16 rays -- which exists only when they share one origin
    and lie in a cone.  Rays with origins of their own (a rolling-shutter sweep: the sensor moves from column to column) and rays whose
    directions are scattered over the sphere inside one tile must take the per-ray tests of round 4 and give the oracle's image and gradients."""
    sc = scenes.make_scene(6000, seed=21, radius_scale=0.25)
    H, W = 8, 96
    o, d = scenes.kitti_rays(H, W)
    o = o.copy(); d = d.copy()
    r = np.random.default_rng(5)
    if case in ("rolling_shutter_origins", "mixed"):
        o = o + (np.arange(W, dtype=np.float32)[None, :, None] * np.array([0.004, -0.002, 0.0005], np.float32)[None, None, :])    # 0.4 m over the sweep
    if case in ("scattered_directions", "mixed"):
        flat = d.reshape(-1, 3); perm = r.permutation(len(flat))
        d = flat[perm].reshape(H, W, 3).copy()                                 # every 16-ray tile now looks in 16 unrelated directions
        if case == "mixed":
            o = o.reshape(-1, 3)[perm].reshape(H, W, 3).copy()
    dL = scenes.upstream_grad(H, W, seed=3)
    fw, bw = oracle_run(sc, o, d, 3, scenes.BG_DEFAULT, dL)
    a = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL)
    b = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL, opts={"fwd_mode": 0, "bwd_mode": 0})
    assert frac_outside(a["out"], fw["out"], 1e-4) <= 2e-3 and rel_l2(a["out"], fw["out"]) < 2e-4, case
    assert rel_l2(a["out"], b["out"]) < 5e-4, case                             # the independent K-buffer packet kernel
    assert rel_l2(a["accum"], fw["accum"]) < 2e-3
    for k in GRADS:
        assert rel_l2(a["grads"][k].reshape(bw[k].shape), bw[k]) < 5e-3 and frac_outside(a["grads"][k].reshape(bw[k].shape), bw[k], 1e-3) <= 2e-2, (case, k)


def test_backward_twice_through_one_forward(s10k):
    """retain_graph: the second backward finds the recorded colours overwritten by the first one and must recompute
    them without changing the result."""
    sc, o, d, dL = s10k
    for place in (0,):
        tr = Tracer()
        for k, v in {"fwd_mode": 2, "bwd_mode": 2 if LEGACY else 3, "defer_colour": 1, "hit_cap": 256}.items():
            tr.optix_context.set_option(k, v)
        t = {k: torch.as_tensor(v, device="cuda:0").requires_grad_(True) for k, v in sc.items()}
        ro, rd = torch.as_tensor(o, device="cuda:0"), torch.as_tensor(d, device="cuda:0")
        from tests.hip_util import settings
        tr.build_from_gaussians(t["means"], t["scales"], t["rotations"], t["opacities"])
        out, _ = tr(ro, rd, None, t["means"], torch.zeros_like(t["means"]), shs=t["shs"], opacities=t["opacities"],
                    scales=t["scales"], rotations=t["rotations"], tracer_settings=settings(scenes.BG_DEFAULT, 3))
        g = torch.as_tensor(dL, device="cuda:0")
        grads = []
        for rep in range(3):
            for v in t.values():
                v.grad = None
            out.backward(g, retain_graph=True)
            grads.append({k: v.grad.detach().cpu().numpy().copy() for k, v in t.items()})
        for k in GRADS:
            assert rel_l2(grads[1][k], grads[0][k]) < 1e-5 and rel_l2(grads[2][k], grads[0][k]) < 1e-5, (place, k)


def test_two_forwards_before_backward(s10k):
    """The hit record belongs to the LAST forward; the backward of an earlier forward must notice and re-trace."""
    from lidar_rt_amd.diff_lidar_tracer import Tracer
    from tests.hip_util import settings, DEV
    sc, o, d, dL = s10k
    fw, bw = oracle_run(sc, o, d, 3, scenes.BG_DEFAULT, dL)
    tr = Tracer()
    t = {k: torch.as_tensor(v, device=DEV).requires_grad_(True) for k, v in sc.items()}
    ro, rd = torch.as_tensor(o, device=DEV), torch.as_tensor(d, device=DEV)
    ts = settings(scenes.BG_DEFAULT, 3)
    tr.build_from_gaussians(t["means"], t["scales"], t["rotations"], t["opacities"])
    kw = dict(shs=t["shs"], opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"], tracer_settings=ts)
    out1, _ = tr(ro, rd, None, t["means"], torch.zeros_like(t["means"]), **kw)
    out2, _ = tr(ro[:8], rd[:8], None, t["means"], torch.zeros_like(t["means"]), **kw)      # a different ray set in between
    out1.backward(torch.as_tensor(dL, device=DEV))
    for k in GRADS:
        assert rel_l2(t[k].grad.cpu().numpy().reshape(bw[k].shape), bw[k]) < 1e-3


@pytest.mark.parametrize("n_slabs", [2, 3, 4, 8])
def test_ray_cone_culled_build_gives_the_same_results(n_slabs):
    """lrt_build_for_slab / lrt_build_for_rays (an N-way azimuth split builds the LBVH for its slab's rays only): conservative culling --
    the cone around the rays and, for (H, W, 3) slabs, the wedge between the planes of the slab's edge columns (what is left at two or three
    ranks, where a 180 / 120 degree slab has no useful cone) -- so the slab's image is bit-identical ((t, gidx) order) and the gradients
    agree up to summation order."""
    from lidar_rt_amd.parallel import column_slab
    sc = scenes.make_scene(30000, seed=31, radius_scale=0.3)
    o, d = scenes.kitti_rays(16, 512)
    dL = scenes.upstream_grad(16, 512)
    t = {k: torch.as_tensor(v, device="cuda:0") for k, v in sc.items()}
    for r in (0, n_slabs // 2, n_slabs - 1):
        a_, b_ = column_slab(512, r, n_slabs)
        os_, ds_, g_ = o[:, a_:b_].copy(), d[:, a_:b_].copy(), dL[:, a_:b_].copy()
        full = run_hip(sc, os_, ds_, 3, scenes.BG_DEFAULT, g_)
        tr = Tracer()
        ro, rd = torch.as_tensor(os_, device="cuda:0"), torch.as_tensor(ds_, device="cuda:0")
        from lidar_rt_amd.diff_lidar_tracer import _C
        from tests.hip_util import settings
        for rep in range(3):      # from the second culled build of a size on, the sort and the tree are sized speculatively
            tt = {k: v.clone().requires_grad_(True) for k, v in t.items()}
            _C.build_from_gaussians(tr.optix_context, tt["means"], tt["scales"], tt["rotations"], tt["opacities"], 1.0, cull_rays=(ro, rd))
            out, acc = tr(ro, rd, None, tt["means"], torch.zeros_like(tt["means"]), shs=tt["shs"], opacities=tt["opacities"],
                          scales=tt["scales"], rotations=tt["rotations"], tracer_settings=settings(scenes.BG_DEFAULT, 3))
            out.backward(torch.as_tensor(g_, device="cuda:0"))
            kept = tr.optix_context.built_count(torch.device("cuda:0"))
            assert 0 < kept < (0.9 if n_slabs == 2 else 0.75) * 30000, kept    # something was actually left out (two ranks: about half plus the margins)
            np.testing.assert_array_equal(out.detach().cpu().numpy(), full["out"])
            assert rel_l2(acc.cpu().numpy(), full["accum"]) < 1e-6
            for k in GRADS:
                assert rel_l2(tt[k].grad.cpu().numpy().reshape(full["grads"][k].shape), full["grads"][k]) < 1e-5, (rep, k)


def test_refit_keeps_the_order_and_gives_the_results_of_a_rebuild(s10k):
    """lrt_refit after the Gaussians moved: same primitive order and tree topology, new records and boxes.  The traversal is
    exhaustive and hits are ordered by (t, index), so image and gradients equal those of a full rebuild."""
    from lidar_rt_amd.diff_lidar_tracer import _C
    from tests.hip_util import settings, DEFAULT_OPTS
    sc, o, d, dL = s10k
    r = np.random.default_rng(5)
    tr = Tracer()
    for k, v in DEFAULT_OPTS.items():
        tr.optix_context.set_option(k, v)
    ro, rd = torch.as_tensor(o, device="cuda:0"), torch.as_tensor(d, device="cuda:0")
    tr.optix_context.refit_interval = 3
    try:
        for it in range(6):                                                    # build, 3 refits, build, refit
            moved = dict(sc)
            moved["means"] = (sc["means"] + 0.02 * it * r.normal(size=sc["means"].shape)).astype(np.float32)
            moved["scales"] = (sc["scales"] * np.exp(0.05 * it * r.normal(size=sc["scales"].shape))).astype(np.float32)
            moved["opacities"] = np.clip(sc["opacities"] + 0.05 * it * r.normal(size=sc["opacities"].shape), 0.001, 0.99).astype(np.float32)
            t = {k: torch.as_tensor(v, device="cuda:0").requires_grad_(True) for k, v in moved.items()}
            _C.build_from_gaussians(tr.optix_context, t["means"], t["scales"], t["rotations"], t["opacities"], 1.0)
            out, acc = tr(ro, rd, None, t["means"], torch.zeros_like(t["means"]), shs=t["shs"], opacities=t["opacities"],
                          scales=t["scales"], rotations=t["rotations"], tracer_settings=settings(scenes.BG_DEFAULT, 3))
            out.backward(torch.as_tensor(dL, device="cuda:0"))
            tr.optix_context.refit_interval = 0
            ref = run_hip(moved, o, d, 3, scenes.BG_DEFAULT, dL)
            tr.optix_context.refit_interval = 3
            assert rel_l2(out.detach().cpu().numpy(), ref["out"]) < 1e-6, it
            for k in GRADS:
                assert rel_l2(t[k].grad.cpu().numpy(), ref["grads"][k]) < 1e-5, (it, k)
    finally:
        tr.optix_context.refit_interval = 0


def test_speculative_culled_build_reports_lost_primitives():
    """The second culled build of a size is sized from the first one's kept count; if more primitives are kept than fit
    (forced here through the test hook), the forward built on it must not pass silently, and the build after it recovers."""
    from lidar_rt_amd.diff_lidar_tracer import _C
    from tests.hip_util import settings, DEFAULT_OPTS
    sc = scenes.make_scene(20000, seed=33, radius_scale=0.3)
    o, d = scenes.kitti_rays(8, 64)
    t = {k: torch.as_tensor(v, device="cuda:0") for k, v in sc.items()}
    ro, rd = torch.as_tensor(o, device="cuda:0"), torch.as_tensor(d, device="cuda:0")
    tr = Tracer()
    for k, v in DEFAULT_OPTS.items():
        tr.optix_context.set_option(k, v)
    tr.eval()

    def frame():
        _C.build_from_gaussians(tr.optix_context, t["means"], t["scales"], t["rotations"], t["opacities"], 1.0, cull_rays=(ro, rd))
        out, _ = tr(ro, rd, None, t["means"], torch.zeros_like(t["means"]), shs=t["shs"], opacities=t["opacities"],
                    scales=t["scales"], rotations=t["rotations"], tracer_settings=settings(scenes.BG_DEFAULT, 3))
        return out.cpu().numpy()

    ref = frame()                                                              # reads the kept count back
    np.testing.assert_array_equal(frame(), ref)                                # speculative size, large enough
    tr.optix_context.set_option("cull_guess", 128)
    with pytest.raises(RuntimeError, match="lost primitives"):
        frame()
    np.testing.assert_array_equal(frame(), ref)                                # reads the count back again
    np.testing.assert_array_equal(frame(), ref)


def test_ctypes_binding_gives_the_results_of_the_torch_extension(s10k, tmp_path):
    """`_C` runs on the torch C++ extension by default; `LRT_TORCH_EXT=0` selects the ctypes binding of the same C ABI.  Same
    inputs, same kernels: the image is bit-identical, the gradients agree to the backward's run-to-run noise."""
    import subprocess, sys
    from lidar_rt_amd.diff_lidar_tracer import _C
    assert _C.BACKEND == "torch-extension", _C._ext_error
    sc, o, d, dL = s10k
    a = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL)
    np.savez(str(tmp_path / "in.npz"), o=o, d=d, dL=dL, **sc)
    code = ("import sys, numpy as np, torch\n"
            "sys.path.insert(0, %r)\n"
            "from lidar_rt_amd.diff_lidar_tracer import _C\n"
            "assert _C.BACKEND == 'ctypes'\n"
            "from lidar_rt_amd import scenes\n"
            "from tests.hip_util import run_hip\n"
            "z = np.load(%r)\n"
            "sc = {k: z[k] for k in ('means', 'scales', 'rotations', 'opacities', 'shs')}\n"
            "r = run_hip(sc, z['o'], z['d'], 3, scenes.BG_DEFAULT, z['dL'])\n"
            "np.savez(%r, out=r['out'], accum=r['accum'], **{'g_' + k: v for k, v in r['grads'].items()})\n"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / "in.npz"), str(tmp_path / "out.npz")))
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, LRT_TORCH_EXT="0"), timeout=600,
                   cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    b = np.load(str(tmp_path / "out.npz"))
    np.testing.assert_array_equal(a["out"], b["out"])
    for k in GRADS:
        assert rel_l2(a["grads"][k], b["g_" + k]) < 1e-5, k
