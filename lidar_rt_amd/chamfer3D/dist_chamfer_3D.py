"""Host side of the Chamfer operator: the counterpart of the reference's
lib/utils/chamfer3D/dist_chamfer_3D.py:31-82 (`chamfer_3DFunction`, `chamfer_3DDist`), same names, argument
meaning and return tuple, over the MI355X C ABI instead of a JIT-compiled CUDA extension.

    dist1, dist2, idx1, idx2 = chamfer_3DDist()(xyz1, xyz2)      # (B,N,3), (B,M,3) float32 HIP tensors

dist are SQUARED nearest-neighbour distances (the reference's convention, metric_utils.py:18-19), idx int32.
"""
from __future__ import annotations

import torch
from torch import nn
from torch.autograd import Function

from . import _C as chamfer_3D


class chamfer_3DFunction(Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        batchsize, n, dim = xyz1.size()
        assert dim == 3, "Wrong last dimension for the chamfer distance 's input! Check with .size()"
        _, m, dim = xyz2.size()
        assert dim == 3, "Wrong last dimension for the chamfer distance 's input! Check with .size()"
        device = xyz1.device
        dist1 = torch.empty(batchsize, n, device=device, dtype=torch.float32)
        dist2 = torch.empty(batchsize, m, device=device, dtype=torch.float32)
        idx1 = torch.empty(batchsize, n, device=device, dtype=torch.int32)
        idx2 = torch.empty(batchsize, m, device=device, dtype=torch.int32)
        chamfer_3D.forward(xyz1, xyz2, dist1, dist2, idx1, idx2)
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        return dist1, dist2, idx1, idx2

    @staticmethod
    def backward(ctx, graddist1, graddist2, gradidx1, gradidx2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        graddist1 = graddist1.contiguous()
        graddist2 = graddist2.contiguous()
        gradxyz1 = torch.zeros_like(xyz1)
        gradxyz2 = torch.zeros_like(xyz2)
        chamfer_3D.backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2)
        return gradxyz1, gradxyz2


class chamfer_3DDist(nn.Module):
    def __init__(self):
        super(chamfer_3DDist, self).__init__()

    def forward(self, input1, input2):
        input1 = input1.contiguous()
        input2 = input2.contiguous()
        return chamfer_3DFunction.apply(input1, input2)
