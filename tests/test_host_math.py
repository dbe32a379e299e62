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
