"""Hand-checkable known-answer cases for the oracle forward
(reference forward.cu:146-308)."""
import numpy as np

from oracle import oracle

C0 = 0.28209479177387814


def _facing_quad(x, op=0.9, s=0.5):
    """A Gaussian at (x,0,0) whose normal is +x (rotate z->x: 90 deg about y)."""
    q = np.array([np.cos(np.pi / 4), 0.0, np.sin(np.pi / 4), 0.0])
    return np.array([x, 0.0, 0.0]), np.array([s, s]), q, op


def _scene(xs, ops=None, sh_dc=(0.3, 0.1, -0.2)):
    n = len(xs)
    ops = ops if ops is not None else [0.9] * n
    m, s, q, o = zip(*[_facing_quad(x, op) for x, op in zip(xs, ops)])
    shs = np.zeros((n, 16, 3)); shs[:, 0, :] = np.array(sh_dc) / C0
    return np.array(m), np.array(s), np.array(q), np.array(o)[:, None], shs


def _run(xs, ops=None, bg=(0.0, 0.0, 1.0), prec="f64", d=(1.0, 0.0, 0.0)):
    m, s, q, o, shs = _scene(xs, ops)
    orc = oracle.Oracle(m, s, q, o, prec)
    ro = np.zeros((1, 1, 3)); rd = np.array(d, float).reshape(1, 1, 3)
    r = orc.forward(ro, rd, shs, 0, np.array(bg), stats=True)
    return r, shs


def test_miss_returns_background():
    r, _ = _run([5.0], d=(-1.0, 0.0, 0.0))
    np.testing.assert_allclose(r["out"][0, 0], [0, 0, 1, 0, 0, 0, 0, 0, 1])
    assert r["accum"][0] == 0 and r["n_cand"][0, 0] == 0


def test_single_hit_known_answer():
    r, _ = _run([5.0], ops=[0.5])
    a = 0.5; c = np.array([0.8, 0.6, 0.3])           # sh_dc + 0.5
    exp = np.concatenate([a * c + (1 - a) * np.array([0, 0, 1]), [a * 5.0, a, 0, 0, 0, 1 - a]])
    np.testing.assert_allclose(r["out"][0, 0], exp, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(r["accum"], [a], rtol=1e-12)


def test_channel0_only_is_clamped():
    m, s, q, o, shs = _scene([5.0], [0.5], sh_dc=(-0.9, -0.9, -0.9))
    orc = oracle.Oracle(m, s, q, o, "f64")
    r = orc.forward(np.zeros((1, 1, 3)), np.array([[[1.0, 0, 0]]]), shs, 0, np.zeros(3))
    np.testing.assert_allclose(r["out"][0, 0, :3], [0.0, 0.5 * -0.4, 0.5 * -0.4], atol=1e-12)


def test_near_hit_below_0p2_is_skipped():
    r, _ = _run([0.1, 5.0], ops=[0.5, 0.5])
    assert r["n_comp"][0, 0] == 1
    np.testing.assert_allclose(r["out"][0, 0, 3], 0.5 * 5.0, rtol=1e-12)


def test_alpha_clamped_at_0p99_and_transmittance_stop():
    # op=0.999 -> alpha = 0.99 ; T: 1 -> 0.01 -> 1e-4 (not < 1e-4) -> third hit would give 1e-6 -> stop
    r, _ = _run([3.0, 4.0, 5.0, 6.0], ops=[0.999] * 4)
    T = r["out"][0, 0, 8]
    assert r["n_comp"][0, 0] in (1, 2)
    assert T >= 1e-4 * (1 - 1e-9)
    # the stopping hit is NOT composited (forward.cu:253-257)
    w_sum = r["out"][0, 0, 4]
    np.testing.assert_allclose(w_sum, 1 - T, rtol=1e-9)


def test_more_than_16_hits_cross_the_chunk_boundary():
    xs = list(np.linspace(2.0, 21.0, 40))
    ops = [0.05] * 40
    r, _ = _run(xs, ops)
    assert r["n_cand"][0, 0] == 40 and r["n_comp"][0, 0] == 40
    T = 1.0; D = 0.0
    for x in xs:
        D += 0.05 * T * x; T *= 0.95
    np.testing.assert_allclose(r["out"][0, 0, 3], D, rtol=1e-10)
    np.testing.assert_allclose(r["out"][0, 0, 8], T, rtol=1e-10)
    # float32 oracle agrees within fp32 noise
    r32, _ = _run(xs, ops, prec="f32")
    np.testing.assert_allclose(r32["out"][0, 0], r["out"][0, 0], rtol=2e-5, atol=1e-6)


def test_hit_within_step_epsilon_after_chunk_boundary_is_dropped():
    # 16 hits, then one 5e-6 behind the 16th: the restart at t16+1e-5 skips it (forward.cu:282-291)
    xs = list(np.linspace(2.0, 9.5, 16)) + [9.5 + 5e-6, 12.0]
    r, _ = _run(xs, [0.05] * 18)
    assert r["n_comp"][0, 0] == 17


def test_input_order_invariance():
    xs = list(np.linspace(2.0, 15.0, 23)); ops = list(np.linspace(0.1, 0.6, 23))
    r1, _ = _run(xs, ops)
    perm = np.random.default_rng(0).permutation(23)
    r2, _ = _run([xs[i] for i in perm], [ops[i] for i in perm])
    np.testing.assert_allclose(r1["out"], r2["out"], rtol=1e-12, atol=1e-14)
