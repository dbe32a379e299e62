"""Worker for tests/test_sharded_gpu.py::test_an_overflow_on_one_rank_is_raised_by_all_ranks (three ranks on one GPU, gloo)."""
import json
import os
import sys

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from lidar_rt_amd import scenes                       # noqa: E402
from lidar_rt_amd.parallel import ShardedTracer       # noqa: E402
from lidar_rt_amd._capi import LrtError               # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="gloo")
    rank = dist.get_rank()
    sc, o, d = scenes.s10k()
    dL = scenes.upstream_grad(*o.shape[:2])
    t = {k: torch.as_tensor(v, device=dev) for k, v in sc.items()}
    ro, rd, g_up = torch.as_tensor(o, device=dev), torch.as_tensor(d, device=dev), torch.as_tensor(dL, device=dev)
    bg = torch.as_tensor(scenes.BG_DEFAULT, device=dev)
    args = (t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg)
    tr = ShardedTracer(exchange="owner")
    tr.cull_build = True
    log = []
    for step in range(6):
        if step == 3 and rank == 1:
            sizing = tr._cull_sizing                           # this rank's culled build is sized far too small once: it loses primitives
            def too_small(key, _orig=sizing):
                _orig(key); tr.backend.state.set_option("cull_next", 64); tr._cull_sizing = _orig
            tr._cull_sizing = too_small
        try:
            out, _ = tr.forward(ro, rd, *args)
            tr.backward(*args, g_up)
            log.append("ok")
        except LrtError as e:
            log.append("raised")
    try:
        tr.check()
        log.append("clean")
    except LrtError:
        log.append("raised-at-end")
    torch.cuda.synchronize()
    gathered = [None] * dist.get_world_size()
    dist.all_gather_object(gathered, log)
    if rank == 0:
        print(json.dumps({"logs": gathered}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
