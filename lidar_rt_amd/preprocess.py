"""Fused pre-processing of the Gaussian parameters: one HIP launch per direction instead of
the reference's chain of PyTorch kernels in front of every trace
(lib/scene/gaussian_model.py:112-148 getters + lib/gaussian_renderer/__init__.py:76-132 concatenation / composition).

    means, scales, rotations, opacities = fused_activations(xyz, log_scales, rot_raw, opacity_logit, seg_start, poses)

* ``xyz`` (P,3), ``log_scales`` (P,2), ``rot_raw`` (P,4), ``opacity_logit`` (P,1): the raw parameters of all assets,
  concatenated in asset order (``torch.cat`` of leaves is differentiable; gradients flow back to each asset).
* ``seg_start`` (A+1,) int32 device tensor: asset a owns Gaussians ``[seg_start[a], seg_start[a+1])``.
* ``poses`` (A,8) float32 device tensor ``[tx,ty,tz, qw,qx,qy,qz, posed]``; ``posed = 0`` for the background / static assets.
  Poses carry no gradient (the reference's do not either, lib/scene/bounding_box.py:53,72).

HIP tensors only; there is no CPU path (the CPU restatement used by the tests lives outside this package).
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence, Tuple

import torch

from . import _capi


def _chk(t, name, shape):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"fused_activations: {name} must be a HIP (cuda) tensor; there is no CPU path")
    if t.dtype != torch.float32 or tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"fused_activations: {name} must be float32 {tuple(shape)}, got {t.dtype} {tuple(t.shape)}")


class _FusedActivations(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, log_scales, rot_raw, opacity_logit, seg_start, poses):
        P = xyz.shape[0]
        _chk(xyz, "xyz", (P, 3)); _chk(log_scales, "log_scales", (P, 2)); _chk(rot_raw, "rot_raw", (P, 4))
        _chk(opacity_logit, "opacity_logit", (P, 1))
        A = poses.shape[0]
        _chk(poses, "poses", (A, 8))
        if seg_start.dtype != torch.int32 or tuple(seg_start.shape) != (A + 1,) or not seg_start.is_cuda:
            raise RuntimeError("fused_activations: seg_start must be an int32 HIP tensor of A+1 segment starts")
        xyz, log_scales, rot_raw, opacity_logit = (t.contiguous() for t in (xyz, log_scales, rot_raw, opacity_logit))
        seg_start, poses = seg_start.contiguous(), poses.contiguous()
        dev = xyz.device
        means = torch.empty_like(xyz); scales = torch.empty_like(log_scales)
        rots = torch.empty_like(rot_raw); opac = torch.empty_like(opacity_logit)
        p = _capi.ptr
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        with torch.cuda.device(idx):
            _capi.check(_capi.load().lrt_preprocess_forward(idx, P, A, p(seg_start), p(poses), p(xyz), p(log_scales), p(rot_raw),
                                                            p(opacity_logit), p(means), p(scales), p(rots), p(opac), stream),
                        "lrt_preprocess_forward")
        ctx.save_for_backward(rot_raw, scales, opac, seg_start, poses)
        return means, scales, rots, opac

    @staticmethod
    def backward(ctx, d_means, d_scales, d_rots, d_opac):
        rot_raw, scales, opac, seg_start, poses = ctx.saved_tensors
        P, A = rot_raw.shape[0], poses.shape[0]
        dev = rot_raw.device
        z = lambda g, like: (torch.zeros_like(like) if g is None else g.contiguous().to(torch.float32))
        d_means = z(d_means, scales.new_empty(P, 3)); d_scales = z(d_scales, scales)
        d_rots = z(d_rots, rot_raw); d_opac = z(d_opac, opac)
        d_xyz = torch.empty(P, 3, device=dev); d_ls = torch.empty_like(scales)
        d_rot = torch.empty_like(rot_raw); d_lo = torch.empty_like(opac)
        p = _capi.ptr
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        with torch.cuda.device(idx):
            _capi.check(_capi.load().lrt_preprocess_backward(idx, P, A, p(seg_start), p(poses), p(rot_raw), p(scales), p(opac),
                                                             p(d_means), p(d_scales), p(d_rots), p(d_opac), p(d_xyz), p(d_ls),
                                                             p(d_rot), p(d_lo), stream), "lrt_preprocess_backward")
        return d_xyz, d_ls, d_rot, d_lo, None, None


def fused_activations(xyz, log_scales, rot_raw, opacity_logit, seg_start, poses):
    """World means (P,3), scales (P,2), unit rotations (P,4), opacities (P,1) of the concatenated assets."""
    return _FusedActivations.apply(xyz, log_scales, rot_raw, opacity_logit, seg_start, poses)


_SEG_CACHE = {}        # (device, counts) -> (segment starts, identity pose row, ones(1)): host->device copies wait for the stream, so they
                       # are made once per asset layout, not per rendered frame


def pack_poses(poses: Sequence[Tuple], counts: Sequence[int], device) -> Tuple[torch.Tensor, torch.Tensor]:
    """Segment starts (A+1,) int32 and the (A,8) pose table from per-asset ``(t (3,), q (4,) or (1,4)) | None`` and sizes.
    No host->device copy after the first call for an asset layout (pose tensors that already live on the device are only
    concatenated there): the rendering loop stays stream-ordered."""
    device = torch.device(device)
    key = (str(device), tuple(int(c) for c in counts))
    ent = _SEG_CACHE.get(key)
    if ent is None:
        A = len(counts)
        seg = torch.zeros(A + 1, dtype=torch.int32)
        seg[1:] = torch.cumsum(torch.tensor(list(counts), dtype=torch.int64), 0).to(torch.int32)
        ident = torch.zeros(8, dtype=torch.float32); ident[3] = 1.0
        if len(_SEG_CACHE) > 64:
            _SEG_CACHE.clear()
        ent = _SEG_CACHE[key] = (seg.to(device), ident.to(device), torch.ones(1, dtype=torch.float32, device=device))
    seg, ident, one = ent
    rows = []
    for ps in poses:
        if ps is None:
            rows.append(ident)
        else:
            t, q = ps
            rows.append(torch.cat([t.reshape(3).to(device, torch.float32), q.reshape(4).to(device, torch.float32), one]))
    return seg, torch.stack(rows, 0).contiguous()
