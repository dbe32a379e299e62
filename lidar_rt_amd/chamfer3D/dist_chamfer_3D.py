"""Host side of the Chamfer operator on the MI355X C ABI.

Public names and call contract follow the reference's wrapper (lib/utils/chamfer3D/dist_chamfer_3D.py:31-82) so that
``chamLoss = chamfer_3DDist(); dist1, dist2, idx1, idx2 = chamLoss(a, b)`` (train.py:197-205, eval.py:355-359) reads the
same: ``a`` (B,N,3), ``b`` (B,M,3) float32 HIP tensors -> SQUARED nearest-neighbour distances (B,N), (B,M) (the
reference's convention, metric_utils.py:18-19) and int32 neighbour indices.  The autograd rule is the analytic one of
chamfer3D.cu:154-173; the index outputs carry no gradient.
"""
from __future__ import annotations

import torch

from . import _C as chamfer_3D


def _nearest(a: torch.Tensor, b: torch.Tensor):
    """Allocate the four outputs and run the forward kernels."""
    if a.dim() != 3 or b.dim() != 3 or a.size(2) != 3 or b.size(2) != 3:
        raise AssertionError("Wrong last dimension for the chamfer distance 's input! Check with .size()")
    shape_a, shape_b = a.shape[:2], b.shape[:2]
    new = lambda shape, dtype: torch.empty(shape, device=a.device, dtype=dtype)
    res = (new(shape_a, torch.float32), new(shape_b, torch.float32), new(shape_a, torch.int32), new(shape_b, torch.int32))
    chamfer_3D.forward(a, b, *res)
    return res


class chamfer_3DFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        d_ab, d_ba, nn_ab, nn_ba = _nearest(xyz1, xyz2)
        ctx.save_for_backward(xyz1, xyz2, nn_ab, nn_ba)
        ctx.mark_non_differentiable(nn_ab, nn_ba)
        return d_ab, d_ba, nn_ab, nn_ba

    @staticmethod
    def backward(ctx, g_ab, g_ba, _g_nn_ab, _g_nn_ba):
        a, b, nn_ab, nn_ba = ctx.saved_tensors
        grad_a, grad_b = torch.zeros_like(a), torch.zeros_like(b)        # the kernels accumulate into them
        chamfer_3D.backward(a, b, grad_a, grad_b, g_ab.contiguous(), g_ba.contiguous(), nn_ab, nn_ba)
        return grad_a, grad_b


class chamfer_3DDist(torch.nn.Module):
    """``forward(input1, input2)`` -> ``(dist1, dist2, idx1, idx2)``."""

    def forward(self, input1, input2):
        return chamfer_3DFunction.apply(input1.contiguous(), input2.contiguous())
