"""ctypes binding of the CPU Chamfer oracle (TEST INFRASTRUCTURE, not product code).

Restates lib/utils/chamfer3D/chamfer3D.cu of the reference; see chamfer_oracle.c for citations and the
parity-pinning status ("parity unpinned" against the CUDA binary; pinned against the float64 definition).
Only ``tests/``, ``__graft_entry__.smoke()`` and benchmark tools' ``cpu_baseline`` legs may import this module.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .oracle import lib, _p


def chamfer_forward(xyz1, xyz2, variant: int = 0):
    """xyz1 (B,N,3), xyz2 (B,M,3) float32 -> dist1 (B,N), dist2 (B,M) float32, idx1, idx2 int32."""
    a = np.ascontiguousarray(xyz1, np.float32); b = np.ascontiguousarray(xyz2, np.float32)
    assert a.ndim == 3 and b.ndim == 3 and a.shape[2] == 3 and b.shape[2] == 3 and a.shape[0] == b.shape[0]
    B, N, M = a.shape[0], a.shape[1], b.shape[1]
    d1 = np.zeros((B, N), np.float32); d2 = np.zeros((B, M), np.float32)
    i1 = np.zeros((B, N), np.int32); i2 = np.zeros((B, M), np.int32)
    lib().orc_chamfer_forward(C.c_int(B), C.c_int(N), _p(a), C.c_int(M), _p(b), _p(d1), _p(d2), _p(i1), _p(i2),
                              C.c_int(variant))
    return d1, d2, i1, i2


def chamfer_backward(xyz1, xyz2, graddist1, graddist2, idx1, idx2, prec: str = "f32"):
    """Gradients w.r.t. xyz1, xyz2 (accumulated from zero)."""
    a = np.ascontiguousarray(xyz1, np.float32); b = np.ascontiguousarray(xyz2, np.float32)
    B, N, M = a.shape[0], a.shape[1], b.shape[1]
    g1 = np.ascontiguousarray(graddist1, np.float32).reshape(B, N); g2 = np.ascontiguousarray(graddist2, np.float32).reshape(B, M)
    i1 = np.ascontiguousarray(idx1, np.int32).reshape(B, N); i2 = np.ascontiguousarray(idx2, np.int32).reshape(B, M)
    t = np.float32 if prec == "f32" else np.float64
    ga = np.zeros((B, N, 3), t); gb = np.zeros((B, M, 3), t)
    getattr(lib(), f"orc_chamfer_backward_{prec}")(C.c_int(B), C.c_int(N), _p(a), C.c_int(M), _p(b), _p(g1), _p(g2),
                                                   _p(i1), _p(i2), _p(ga), _p(gb))
    return ga, gb


def knn_mean_dist2(points):
    """simple-knn `distCUDA2`: points (P,3) float32 -> (P,) mean squared distance to the 3 nearest other points."""
    a = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
    out = np.zeros(a.shape[0], np.float32)
    lib().orc_knn_mean_dist2(C.c_int(a.shape[0]), _p(a), _p(out))
    return out
