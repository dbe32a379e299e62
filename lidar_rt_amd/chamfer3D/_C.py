"""`chamfer_3D` -- the native module of the reference's Chamfer extension, on the MI355X C ABI.

Mirrors the pybind module built from lib/utils/chamfer3D/chamfer_cuda.cpp:29-32 of the reference:
``forward(xyz1, xyz2, dist1, dist2, idx1, idx2)`` and
``backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2)``, both writing into caller-allocated
tensors and returning 1.  Every tensor must be a contiguous HIP tensor; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from .. import _capi

_states: Dict[int, "ChamferState"] = {}


class ChamferState:
    """Per-device workspace owner (`lrt_chamfer_create`)."""

    def __init__(self, device_index: int):
        lib = _capi.load()
        self.device_index = device_index
        self._h = lib.lrt_chamfer_create(device_index)
        if not self._h:
            raise _capi.LrtError(f"lrt_chamfer_create failed: {lib.lrt_last_error().decode()}")

    def set_option(self, name: str, value: int):
        _capi.check(_capi.load().lrt_chamfer_set_option(self._h, name.encode(), int(value)), "lrt_chamfer_set_option")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _capi.load().lrt_chamfer_destroy(self._h)
                self._h = None
        except Exception:
            pass


def state_for(device: torch.device) -> ChamferState:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _states.get(idx)
    if st is None:
        st = _states[idx] = ChamferState(idx)
    return st


def set_option(name: str, value: int, device=None):
    """`mode` 0 brute force / 1 tree / 2 auto; `brute_max_pairs_log2`."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    state_for(dev).set_option(name, value)


def _chk(t: torch.Tensor, name: str, dtype, shape):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"chamfer_3D: {name} must be a HIP (cuda) tensor; there is no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"chamfer_3D: {name} must be {dtype}, got {t.dtype}")
    if tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"chamfer_3D: {name} must have shape {tuple(shape)}, got {tuple(t.shape)}")
    if not t.is_contiguous():
        raise RuntimeError(f"chamfer_3D: {name} must be contiguous")


def _dims(xyz1, xyz2):
    if xyz1.dim() != 3 or xyz2.dim() != 3 or xyz1.shape[2] != 3 or xyz2.shape[2] != 3 or xyz1.shape[0] != xyz2.shape[0]:
        raise RuntimeError(f"chamfer_3D: xyz1/xyz2 must be (B,N,3)/(B,M,3), got {tuple(xyz1.shape)} / {tuple(xyz2.shape)}")
    return xyz1.shape[0], xyz1.shape[1], xyz2.shape[1]


def forward(xyz1, xyz2, dist1, dist2, idx1, idx2) -> int:
    B, N, M = _dims(xyz1, xyz2)
    _chk(xyz1, "xyz1", torch.float32, (B, N, 3)); _chk(xyz2, "xyz2", torch.float32, (B, M, 3))
    _chk(dist1, "dist1", torch.float32, (B, N)); _chk(dist2, "dist2", torch.float32, (B, M))
    _chk(idx1, "idx1", torch.int32, (B, N)); _chk(idx2, "idx2", torch.int32, (B, M))
    st = state_for(xyz1.device)
    stream = C.c_void_p(torch.cuda.current_stream(xyz1.device).cuda_stream)
    p = _capi.ptr
    with torch.cuda.device(xyz1.device):
        _capi.check(_capi.load().lrt_chamfer_forward(st._h, B, N, p(xyz1), M, p(xyz2), p(dist1), p(dist2), p(idx1),
                                                     p(idx2), stream), "lrt_chamfer_forward")
    return 1


def backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2) -> int:
    B, N, M = _dims(xyz1, xyz2)
    _chk(xyz1, "xyz1", torch.float32, (B, N, 3)); _chk(xyz2, "xyz2", torch.float32, (B, M, 3))
    _chk(gradxyz1, "gradxyz1", torch.float32, (B, N, 3)); _chk(gradxyz2, "gradxyz2", torch.float32, (B, M, 3))
    _chk(graddist1, "graddist1", torch.float32, (B, N)); _chk(graddist2, "graddist2", torch.float32, (B, M))
    _chk(idx1, "idx1", torch.int32, (B, N)); _chk(idx2, "idx2", torch.int32, (B, M))
    st = state_for(xyz1.device)
    stream = C.c_void_p(torch.cuda.current_stream(xyz1.device).cuda_stream)
    p = _capi.ptr
    with torch.cuda.device(xyz1.device):
        _capi.check(_capi.load().lrt_chamfer_backward(st._h, B, N, p(xyz1), M, p(xyz2), p(graddist1), p(graddist2),
                                                      p(idx1), p(idx2), p(gradxyz1), p(gradxyz2), stream),
                    "lrt_chamfer_backward")
    return 1
