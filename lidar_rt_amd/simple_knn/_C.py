"""`simple_knn._C` -- the reference's init-time k-NN extension on the MI355X C ABI (include/lrt_knn.h).

Mirrors submodules/simple-knn/ext.cpp:15-17: one function, ``distCUDA2(points) -> (P,) float32``, the mean squared
distance of every point to its three nearest neighbours (spatial.cu:15-26).  Called once per scene by
lib/scene/gaussian_model.py:167 to seed the Gaussian scales.  HIP tensors only; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import _capi
from ..chamfer3D._C import state_for


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    if not isinstance(points, torch.Tensor) or not points.is_cuda:
        raise RuntimeError("distCUDA2: points must be a HIP (cuda) tensor; there is no CPU path")
    if points.dim() != 2 or points.shape[1] != 3:
        raise RuntimeError(f"distCUDA2: points must be (P,3), got {tuple(points.shape)}")
    if points.dtype != torch.float32:
        raise RuntimeError(f"distCUDA2: points must be float32, got {points.dtype}")     # the reference reinterprets the bytes
    pts = points.contiguous()
    P = pts.shape[0]
    means = torch.zeros(P, device=pts.device, dtype=torch.float32)
    if P == 0:
        return means
    st = state_for(pts.device)
    stream = C.c_void_p(torch.cuda.current_stream(pts.device).cuda_stream)
    with torch.cuda.device(pts.device):
        _capi.check(_capi.load().lrt_knn_mean_dist2(st._h, P, _capi.ptr(pts), _capi.ptr(means), stream), "lrt_knn_mean_dist2")
    return means
