"""CPU restatement of the Gaussian pre-processing chain (TEST INFRASTRUCTURE, not product code).

Follows, with plain torch CPU ops (so that torch autograd gives the reference gradients):
  lib/scene/gaussian_model.py:112-113   get_scaling  = exp(_scaling)
  lib/scene/gaussian_model.py:147-148   get_opacity  = sigmoid(_opacity)
  lib/scene/gaussian_model.py:116-127   get_rotation = (actor quaternion, F.normalize(_rotation))
  lib/scene/gaussian_model.py:129-134   get_world_xyz = _xyz @ build_rotation(q_actor)^T + t_actor
  lib/utils/general_utils.py:176-197    build_rotation (normalises the quaternion)
  lib/utils/general_utils.py:156-174    quaternion_raw_multiply
  lib/gaussian_renderer/__init__.py:111-132  concatenation over assets, actor rotations = q_actor (x) normalize(q_local)

PARITY PINNING: pinned against the reference's own Python functions -- tests/golden/preprocess_golden.npz holds inputs
and the outputs of the reference's build_rotation / quaternion_raw_multiply / activations executed on CPU by
oracle/gen_golden.py (and torch autograd through them).
Only tests/ may import this module.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def build_rotation(r: torch.Tensor) -> torch.Tensor:
    q = r / torch.sqrt((r * r).sum(1, keepdim=True))
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)
    return R.view(-1, 3, 3)


def quaternion_raw_multiply(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    aw, ax, ay, az = torch.unbind(a, -1)
    bw, bx, by, bz = torch.unbind(b, -1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw), -1)


def preprocess(xyz, log_scales, rot_raw, opacity_logit, seg_start, poses):
    """Same contract as lidar_rt_amd.preprocess.fused_activations, CPU tensors, differentiable."""
    means, rots = [], []
    A = poses.shape[0]
    for a in range(A):
        s, e = int(seg_start[a]), int(seg_start[a + 1])
        x = xyz[s:e]; b = F.normalize(rot_raw[s:e], dim=1)
        if float(poses[a, 7]) != 0.0:
            q = poses[a, 3:7].reshape(1, 4)
            R = build_rotation(q).squeeze(0)
            means.append(x @ R.T + poses[a, 0:3])
            rots.append(quaternion_raw_multiply(q.expand(e - s, -1), b))
        else:
            means.append(x); rots.append(b)
    return torch.cat(means, 0), torch.exp(log_scales), torch.cat(rots, 0), torch.sigmoid(opacity_logit)
