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
    # ---- the sharded step (forward + slab all_gather + backward + device-side gathering exchange) enqueued behind a busy GPU: it must
    # return to the host without waiting for the device (no count read-back inside the exchange since round 4)
    import time
    tr = ShardedTracer(exchange="sparse"); tr.force_collectives = True
    for _ in range(4):
        tr.forward(ro, rd, *args); tr.backward(*args, g_up)
    torch.cuda.synchronize()
    torch.cuda._sleep(1_000_000); torch.cuda.synchronize()
    t0 = time.perf_counter(); torch.cuda._sleep(20_000_000); torch.cuda.synchronize()
    rate = 20_000_000 / max(time.perf_counter() - t0, 1e-6)
    torch.cuda._sleep(int(0.4 * rate))
    marker = torch.cuda.Event(); marker.record()
    t0 = time.perf_counter()
    tr.forward(ro, rd, *args); tr.backward(*args, g_up)
    host_s = time.perf_counter() - t0
    still_busy = not marker.query()
    torch.cuda.synchronize()
    tr.check()
    print(json.dumps({"backend": dist.get_backend(), "exchanges": ["owner", "dense", "sparse"], "rel_err": rel,
                      "sharded_step_host_s": host_s, "gpu_still_busy_after_enqueue": bool(still_busy)}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
