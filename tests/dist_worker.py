"""Worker for tests/test_distributed_cpu.py (launched once per rank, gloo backend)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from lidar_rt_amd import scenes                       # noqa: E402
from lidar_rt_amd.parallel import ShardedTracer, column_slab   # noqa: E402
from tests.oracle_backend import OracleBackend        # noqa: E402


def main():
    out_path = sys.argv[1]
    exchange = sys.argv[2] if len(sys.argv) > 2 else "auto"
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    sc = scenes.make_scene(1500, seed=4, radius_scale=0.2)
    o, d = scenes.kitti_rays(6, 45)                   # W=45 is not divisible by the world size
    dL = scenes.upstream_grad(6, 45)
    t = {k: torch.from_numpy(v) for k, v in sc.items()}
    bg = torch.tensor([0.0, 0.0, 1.0])
    want = "dense" if exchange in ("dense", "auto-dense") else ("owner" if exchange == "owner" else "sparse")
    tr = ShardedTracer(backend=OracleBackend(), exchange="auto" if exchange == "auto-dense" else exchange)
    if exchange == "auto-dense":
        tr.sparse_max_fraction = 0.0                  # "auto" must fall back to the dense all_reduce (decided identically on every rank)
    for step in range(3 if exchange == "owner" else 1):   # owner: exact capacity first, then capacities speculated from the counts
        out, _ = tr.forward(torch.from_numpy(o), torch.from_numpy(d), t["means"], t["scales"], t["rotations"],
                            t["opacities"], t["shs"], 3, bg)
        g = tr.backward(t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg, torch.from_numpy(dL))
    tr.check()
    a, b = column_slab(45, rank, world)
    assert tr._slab == (a, b)
    assert tr.last_exchange == want, tr.last_exchange
    extra = {"owner": tr.last_owner.numpy()} if exchange == "owner" else {}
    np.savez(out_path + f".rank{rank}.npz", out=out.numpy(), **{k: v.numpy() for k, v in g.items()}, **extra)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
