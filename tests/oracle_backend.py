"""TEST INFRASTRUCTURE: a CPU stand-in for the local tracer of lidar_rt_amd.parallel.ShardedTracer, backed by the
oracle, so that the sharding / collective logic can be exercised with gloo on a machine without GPUs."""
import numpy as np
import torch

from oracle import oracle


class OracleBackend:
    def __init__(self):
        self.orc = None

    def build(self, means, scales, rotations, opacities, mod=1.0):
        self.orc = oracle.Oracle(means.numpy(), scales.numpy(), rotations.numpy(), opacities.numpy(), "f64", mod)

    def forward(self, ray_o, ray_d, means, scales, rotations, opacities, shs, deg, bg, mod=1.0):
        fw = self.orc.forward(ray_o.numpy(), ray_d.numpy(), shs.numpy(), deg, bg.numpy())
        return torch.from_numpy(fw["out"].astype(np.float32)), torch.from_numpy(fw["accum"].astype(np.float32))

    def backward(self, ray_o, ray_d, means, scales, rotations, opacities, shs, deg, bg, out, dL, mod=1.0):
        g = self.orc.backward(ray_o.numpy(), ray_d.numpy(), shs.numpy(), deg, bg.numpy(), out.numpy(), dL.numpy())
        return {k: torch.from_numpy(v.astype(np.float32)) for k, v in g.items()}
