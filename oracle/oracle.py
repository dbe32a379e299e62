"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE, not product code).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module.  See ``lrt_oracle_impl.inc`` for
what the oracle restates and how far its parity is pinned.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblrt_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("lrt_oracle.c", "lrt_oracle_impl.inc", "chamfer_oracle.c")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liblrt_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        for sfx in ("f32", "f64"):
            getattr(_lib, f"orc_build_{sfx}").restype = C.c_void_p
            getattr(_lib, f"orc_destroy_{sfx}").restype = None
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _dt(prec):
    return (np.float32, "f32", C.c_float) if prec == "f32" else (np.float64, "f64", C.c_double)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Oracle:
    """One built scene (quads + CPU BVH).  ``prec`` is 'f32' (the reference's
    arithmetic type) or 'f64'."""

    def __init__(self, means, scales, rotations, opacities, prec: str = "f32",
                 scale_modifier: float = 1.0):
        self.np_t, self.sfx, self.c_t = _dt(prec)
        t = self.np_t
        self.means = np.ascontiguousarray(means, t).reshape(-1, 3)
        self.P = self.means.shape[0]
        self.scales = np.ascontiguousarray(scales, t).reshape(self.P, 2)
        self.rot = np.ascontiguousarray(rotations, t).reshape(self.P, 4)
        self.opac = np.ascontiguousarray(opacities, t).reshape(self.P)
        self.mod = float(scale_modifier)
        self.vertices = np.empty((4 * self.P, 3), t)
        L = lib()
        self._h = C.c_void_p(getattr(L, f"orc_build_{self.sfx}")(
            C.c_int(self.P), _p(self.means), _p(self.scales), _p(self.rot), _p(self.opac),
            _p(self.vertices)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                getattr(lib(), f"orc_destroy_{self.sfx}")(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def faces(self):
        base = np.array([[0, 1, 2], [2, 3, 1]], np.int32)
        return (base[None] + 4 * np.arange(self.P, dtype=np.int32)[:, None, None]).reshape(-1, 3)

    def forward(self, ray_o, ray_d, shs, sh_degree: int, bg, stats: bool = False) -> Dict[str, np.ndarray]:
        t = self.np_t
        ray_o = np.ascontiguousarray(ray_o, t); ray_d = np.ascontiguousarray(ray_d, t)
        H, W = ray_o.shape[:2]
        shs = np.ascontiguousarray(shs, t).reshape(self.P, -1, 3)
        M = shs.shape[1]
        bg = np.ascontiguousarray(bg, t).reshape(3)
        out = np.zeros((H, W, 9), t)
        accum = np.zeros(self.P, t)
        nc = np.zeros((H, W), np.int32) if stats else None
        nk = np.zeros((H, W), np.int32) if stats else None
        getattr(lib(), f"orc_forward_{self.sfx}")(
            self._h, C.c_int(H), C.c_int(W), _p(ray_o), _p(ray_d), C.c_int(M), C.c_int(sh_degree),
            _p(shs), _p(self.means), _p(self.scales), _p(self.rot), _p(self.opac),
            self.c_t(self.mod), _p(bg), _p(out), _p(accum), _p(nc), _p(nk))
        res = {"out": out, "accum": accum}
        if stats:
            res["n_cand"] = nc; res["n_comp"] = nk
        return res

    def forward_trace(self, ray_o, ray_d, shs, sh_degree: int, bg, cap: int = 160) -> Dict[str, np.ndarray]:
        """The forward with a per-ray EVENT TRACE (tests/tools/parity_events.py): every candidate in the order the raygen loop looks at it:
        n (H,W), g / t / alpha (un-clamped) / flags (H,W,cap); flags: 1 composited, 2 skipped (alpha < 1/255), 4 stopped the ray,
        8 first candidate after a restart."""
        t = self.np_t
        ray_o = np.ascontiguousarray(ray_o, t); ray_d = np.ascontiguousarray(ray_d, t)
        H, W = ray_o.shape[:2]
        shs = np.ascontiguousarray(shs, t).reshape(self.P, -1, 3)
        bg = np.ascontiguousarray(bg, t).reshape(3)
        out = np.zeros((H, W, 9), t)
        n = np.zeros((H, W), np.int32); g = np.zeros((H, W, cap), np.int32)
        tt = np.zeros((H, W, cap), t); aa = np.zeros((H, W, cap), t); ff = np.zeros((H, W, cap), np.uint8)
        getattr(lib(), f"orc_forward_trace_{self.sfx}")(
            self._h, C.c_int(H), C.c_int(W), _p(ray_o), _p(ray_d), C.c_int(shs.shape[1]), C.c_int(sh_degree),
            _p(shs), _p(self.means), _p(self.scales), _p(self.rot), _p(self.opac),
            self.c_t(self.mod), _p(bg), _p(out), C.c_int(cap), _p(n), _p(g), _p(tt), _p(aa), _p(ff))
        return {"out": out, "n": n, "g": g, "t": tt, "alpha": aa, "flags": ff}

    def backward(self, ray_o, ray_d, shs, sh_degree: int, bg, out, dL_dout) -> Dict[str, np.ndarray]:
        t = self.np_t
        ray_o = np.ascontiguousarray(ray_o, t); ray_d = np.ascontiguousarray(ray_d, t)
        H, W = ray_o.shape[:2]
        shs = np.ascontiguousarray(shs, t).reshape(self.P, -1, 3)
        M = shs.shape[1]
        bg = np.ascontiguousarray(bg, t).reshape(3)
        out = np.ascontiguousarray(out, t); dL = np.ascontiguousarray(dL_dout, t)
        g = {"means": np.zeros((self.P, 3), t), "shs": np.zeros((self.P, M, 3), t),
             "opacities": np.zeros((self.P, 1), t), "scales": np.zeros((self.P, 2), t),
             "rotations": np.zeros((self.P, 4), t)}
        getattr(lib(), f"orc_backward_{self.sfx}")(
            self._h, C.c_int(H), C.c_int(W), _p(ray_o), _p(ray_d), C.c_int(M), C.c_int(sh_degree),
            _p(shs), _p(self.means), _p(self.scales), _p(self.rot), _p(self.opac),
            self.c_t(self.mod), _p(bg), _p(out), _p(dL),
            _p(g["means"]), _p(g["shs"]), _p(g["opacities"]), _p(g["scales"]), _p(g["rotations"]))
        return g


def quat_to_R(q, prec="f64"):
    t, sfx, _ = _dt(prec)
    q = np.ascontiguousarray(q, t).reshape(-1, 4)
    R = np.empty((q.shape[0], 3, 3), t)
    f = getattr(lib(), f"orc_quat_to_R_{sfx}")
    for i in range(q.shape[0]):
        f(_p(q[i]), _p(R[i]))
    return R


def sh_basis(deg, dirs, prec="f64"):
    t, sfx, _ = _dt(prec)
    d = np.ascontiguousarray(dirs, t).reshape(-1, 3)
    b = np.empty((d.shape[0], 16), t)
    f = getattr(lib(), f"orc_sh_basis_{sfx}")
    for i in range(d.shape[0]):
        f(C.c_int(deg), _p(d[i]), _p(b[i]))
    return b


def set_sorted_anyhit(on: bool) -> None:
    """Feed the any-hit program in ascending t instead of BVH order (matters only for rays with a hit closer than 0.2 m: the
    reference's stale-slot behaviour is order-dependent there; this is the realisation the HIP path replays)."""
    lib().orc_set_sorted_anyhit(C.c_int(1 if on else 0))


def num_threads() -> int:
    return int(lib().orc_num_threads())


def set_num_threads(n: int) -> None:
    lib().orc_set_num_threads(C.c_int(n))
