"""Worker for tests/test_sharded_gpu.py::test_rccl_preflight_world_of_one: ShardedTracer on backend "nccl" (RCCL) with one rank."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from lidar_rt_amd import scenes                       # noqa: E402
from lidar_rt_amd.parallel import ShardedTracer       # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", device_id=dev)
    sc, o, d = scenes.s10k()
    dL = scenes.upstream_grad(*o.shape[:2])
    t = {k: torch.as_tensor(v, device=dev) for k, v in sc.items()}
    ro, rd, g_up = torch.as_tensor(o, device=dev), torch.as_tensor(d, device=dev), torch.as_tensor(dL, device=dev)
    bg = torch.as_tensor(scenes.BG_DEFAULT, device=dev)
    args = (t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg)
    plain = ShardedTracer()
    out0, _ = plain.forward(ro, rd, *args)
    g0 = {k: v.clone() for k, v in plain.backward(*args, g_up).items()}
    rel = {}
    for ex in ("owner", "dense", "sparse"):
        tr = ShardedTracer(exchange=ex)
        tr.force_collectives = True
        for _ in range(3):
            out, _ = tr.forward(ro, rd, *args)
            g = tr.backward(*args, g_up)
        tr.check()
        assert tr.last_exchange == ex, tr.last_exchange
        rel[ex + ".out"] = float((out - out0).norm() / out0.norm())
        for k in ("means", "shs", "accum"):
            rel[f"{ex}.{k}"] = float((g[k] - g0[k]).norm() / g0[k].norm().clamp_min(1e-30))
    torch.cuda.synchronize()
    print(json.dumps({"backend": dist.get_backend(), "exchanges": ["owner", "dense", "sparse"], "rel_err": rel}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
