"""The product's per-hit math header (lidar_rt_amd/csrc/lrt_math.h), compiled for the host and driven by a
brute-force loop (tests/host_check/host_check.cpp), against the oracle.  Catches formula errors in what the
HIP kernels compute without needing a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from lidar_rt_amd import scenes
from oracle import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_check", "host_check.cpp")
LIB = os.path.join(HERE, "host_check", "libhost_check.so")


@pytest.fixture(scope="module")
def hc():
    hdr = os.path.join(HERE, "..", "lidar_rt_amd", "csrc", "lrt_math.h")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", LIB, SRC])
    return C.CDLL(LIB)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def run_host(hc, sc, o, d, deg, bg, mod=1.0, dL=None, out=None):
    P = sc["means"].shape[0]
    f = lambda a: np.ascontiguousarray(a, np.float32)
    m, s, r, op, sh = f(sc["means"]), f(sc["scales"]), f(sc["rotations"]), f(sc["opacities"]).reshape(-1), f(sc["shs"])
    ro, rd = f(o).reshape(-1, 3), f(d).reshape(-1, 3)
    n = ro.shape[0]; M = sh.shape[1]; bg = f(bg)
    if dL is None:
        out9 = np.zeros((n, 9), np.float32); acc = np.zeros(P, np.float32)
        hc.hc_trace(P, _p(m), _p(s), _p(r), _p(op), C.c_float(mod), n, _p(ro), _p(rd), M, deg, _p(sh), _p(bg), 0,
                    _p(out9), _p(acc), None, None, None, None, None, None)
        return out9.reshape(o.shape[0], o.shape[1], 9), acc
    g = {"means": np.zeros((P, 3)), "shs": np.zeros((P, M, 3)), "opacities": np.zeros((P, 1)),
         "scales": np.zeros((P, 2)), "rotations": np.zeros((P, 4))}
    out9 = f(out).reshape(-1, 9); dLf = f(dL).reshape(-1, 9)
    hc.hc_trace(P, _p(m), _p(s), _p(r), _p(op), C.c_float(mod), n, _p(ro), _p(rd), M, deg, _p(sh), _p(bg), 1,
                _p(out9), None, _p(dLf), _p(g["means"]), _p(g["shs"]), _p(g["opacities"]), _p(g["scales"]),
                _p(g["rotations"]))
    return g


@pytest.mark.parametrize("deg,mod", [(3, 1.0), (1, 1.0), (0, 1.0), (3, 1.3)])
def test_product_math_matches_oracle_s10k(hc, deg, mod):
    sc, o, d = scenes.s10k()
    bg = scenes.BG_DEFAULT
    orc = oracle.Oracle(sc["means"], sc["scales"], sc["rotations"], sc["opacities"], "f64", scale_modifier=mod)
    fw = orc.forward(o, d, sc["shs"], deg, bg)
    out, acc = run_host(hc, sc, o, d, deg, bg, mod)
    scale = np.abs(fw["out"]).reshape(-1, 9).max(0) + 1e-12
    rel = np.abs(out - fw["out"]) / scale
    assert rel.max() < 2e-4, rel.reshape(-1, 9).max(0)          # fp32 product math vs f64 oracle
    assert np.abs(acc - fw["accum"]).max() < 2e-4 * fw["accum"].max()
    rng = np.random.default_rng(1)
    dL = scenes.upstream_grad(16, 256)
    dL[..., 5:8] = rng.normal(size=(16, 256, 3)).astype(np.float32) / (16 * 256)   # exercise the normal route (D3)
    bw = orc.backward(o, d, sc["shs"], deg, bg, fw["out"], dL)
    g = run_host(hc, sc, o, d, deg, bg, mod, dL=dL, out=fw["out"])
    for k in bw:
        ref = bw[k]
        err = np.abs(g[k] - ref).max() / np.abs(ref).max()
        assert err < 1e-3, (k, err)


def _literal_vertex_route_f64(o, d, t, mu, sc, q, op, dL_dG, dL_dD, dN):
    """backward.cu:339-431 + :621-652 + auxiliary.h:389-433 in float64 (one hit), the way rounds 1-4 evaluated it in float32."""
    o, d, mu, sc, q, dN = (np.asarray(a, np.float64) for a in (o, d, mu, sc, q, dN))
    qn = q / np.linalg.norm(q); w, x, y, z = qn
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    L0, L1 = R[:, 0] / sc[0], R[:, 1] / sc[1]
    pd = o + t * d - mu
    u, v = L0 @ pd, L1 @ pd
    G = np.exp(-0.5 * (u * u + v * v))
    nsign = 1.0 if -((mu - o) @ R[:, 2]) > 0 else -1.0
    dL_du, dL_dv = dL_dG * -G * u, dL_dG * -G * v
    dR0, dR1, dR2 = dL_du * pd / sc[0], dL_dv * pd / sc[1], dN * nsign
    d_scale = np.array([dL_dG * G * u * u / sc[0], dL_dG * G * v * v / sc[1]])
    d_mean = dL_dG * G * (L0 * u + L1 * v)
    dL_dd = (dL_du * L0 + dL_dv * L1) @ d + dL_dD
    cut = np.sqrt(2 * np.log(op * 255.0)) + 0.01
    ex, ey = sc[0] * cut, sc[1] * cut
    cs = [(-1, 1), (-1, -1), (1, 1), (1, -1)]
    V = [a * ex * R[:, 0] + b * ey * R[:, 1] + mu for a, b in cs]
    odd = v < u
    v1, v2, v3 = (V[1], V[2], V[3]) if odd else (V[0], V[1], V[2])
    h1, h2, h3 = ((-cut, -cut), (cut, cut), (cut, -cut)) if odd else ((-cut, cut), (-cut, -cut), (cut, cut))
    n = np.cross(v2 - v1, v3 - v1); c = v1 - o
    p_, qq = n @ c, n @ d
    gn = (c - p_ / qq * d) / qq
    dv1 = (np.cross(v2 - v3, gn) + n / qq) * dL_dd; dv2 = np.cross(v3 - v1, gn) * dL_dd; dv3 = np.cross(v1 - v2, gn) * dL_dd
    sxv = h1[0] * dv1 + h2[0] * dv2 + h3[0] * dv3; syv = h1[1] * dv1 + h2[1] * dv2 + h3[1] * dv3
    dR0 = dR0 + sc[0] * sxv; dR1 = dR1 + sc[1] * syv
    d_mean = d_mean + dv1 + dv2 + dv3
    d_scale = d_scale + np.array([sc[0] * (L0 @ sxv), sc[1] * (L1 @ syv)])
    d_rot = 2 * np.array([x * (dR1[2] - dR2[1]) + y * (dR2[0] - dR0[2]) + z * (dR0[1] - dR1[0]),
                          -2 * x * (dR1[1] + dR2[2]) + y * (dR0[1] + dR1[0]) + z * (dR0[2] + dR2[0]) + w * (dR1[2] - dR2[1]),
                          x * (dR0[1] + dR1[0]) - 2 * y * (dR0[0] + dR2[2]) + z * (dR1[2] + dR2[1]) + w * (dR2[0] - dR0[2]),
                          x * (dR0[2] + dR2[0]) + y * (dR1[2] + dR2[1]) - 2 * z * (dR0[0] + dR1[1]) + w * (dR0[1] - dR1[0])])
    return np.concatenate([d_mean, d_scale, d_rot])


def test_closed_form_hit_backward_equals_the_literal_vertex_route(hc):
    """Round 5: lrt_hit_backward evaluates the reference's depth-through-the-vertices route (backward.cu:339-431, :621-652) in closed form.
    On random hits (random Gaussians, rays through random points of their quads, either triangle) it must agree with the literal chain
    evaluated in float64, and be CLOSER to it than the float32 literal chain of rounds 1-4 (kept in tests/host_check/host_check.cpp),
    whose corner terms cancel."""
    rng = np.random.default_rng(3)
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    rows_c, rows_l, rows_x = [], [], []
    for i in range(1500):
        mu = f32(rng.uniform(-30, 30, 3)); sc = f32(np.exp(rng.uniform(np.log(0.03), np.log(0.25), 2)))
        q = rng.standard_normal(4); q = f32(q / np.linalg.norm(q) * rng.uniform(0.5, 2.0))       # the kernels normalise it
        op = np.float32(rng.uniform(0.05, 0.99))
        qn = q.astype(np.float64) / np.linalg.norm(q.astype(np.float64))
        w, x, y, z = qn
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        cut = np.sqrt(2 * np.log(float(op) * 255.0)) + 0.01
        uv = rng.uniform(-1, 1, 2) * cut
        hit = mu + R[:, 0] * sc[0] * uv[0] + R[:, 1] * sc[1] * uv[1]
        o = f32(rng.uniform(-2, 2, 3))
        dvec = hit - o; d = f32(dvec / np.linalg.norm(dvec))
        if abs(d.astype(np.float64) @ R[:, 2]) < 0.05:
            continue                                                  # grazing rays: every form divides by (n.d)^2
        # the hit distance as the kernels have it: on the plane, in float32
        t = np.float32(((mu.astype(np.float64) - o) @ R[:, 2]) / (d.astype(np.float64) @ R[:, 2]))
        dN = f32(rng.standard_normal(3) * 1e-3)
        dG, dD = np.float32(rng.standard_normal() * 1e-2), np.float32(rng.standard_normal() * 1e-2)
        oc, ol = np.zeros(9, np.float32), np.zeros(9, np.float32)
        hc.hc_hit_backward_both(_p(o), _p(d), C.c_float(float(t)), _p(mu), _p(sc), _p(q), C.c_float(float(op)), C.c_float(1.0),
                                C.c_float(float(dG)), C.c_float(float(dD)), _p(dN), _p(oc), _p(ol))
        rows_c.append(oc.astype(np.float64)); rows_l.append(ol.astype(np.float64))
        rows_x.append(_literal_vertex_route_f64(o, d, float(t), mu, sc, q, float(op), float(dG), float(dD), dN))
    c, l, x = np.array(rows_c), np.array(rows_l), np.array(rows_x)
    assert len(c) > 1000 and np.isfinite(c).all() and np.isfinite(l).all() and np.isfinite(x).all()
    e_closed, e_literal = np.linalg.norm(c - x) / np.linalg.norm(x), np.linalg.norm(l - x) / np.linalg.norm(x)
    assert e_closed < 2e-5, (e_closed, e_literal)                    # the closed form IS the literal route ...
    assert e_closed <= e_literal, (e_closed, e_literal)              # ... and no farther from it than its own float32 evaluation
    scale = np.abs(x).mean(0) + 1e-30
    assert (np.abs(c - x) / scale).max() < 5e-3, (np.abs(c - x) / scale).max(0)
