#!/usr/bin/env python3
"""Developer tool: where does the HOST time of one rank's step go?  cProfile over the enqueue loop of a rank of an N-way split (no device wait
inside the loop); the GPU step of a rank of 8 is 0.4 ms, the host must stay below it.  env SLAB_N (8), WORKLOAD, STEPS (200)"""
import cProfile, os, pstats, sys, time
if os.environ.get("SLAB_N", "8") != "1": os.environ.setdefault("LRT_PREZERO", "force")
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from lidar_rt_amd import scenes
from lidar_rt_amd.parallel import ShardedTracer, column_slab

dev = torch.device("cuda:0")
wl = os.environ.get("WORKLOAD", "s1m")
sc, ro, rd = scenes.waymo_dynamic_4m() if wl == "waymo4m" else scenes.s10k() if wl == "s10k" else scenes.s1m()
t = {k: torch.as_tensor(v, device=dev) for k, v in sc.items()}
H, W = ro.shape[:2]
N = int(os.environ.get("SLAB_N", "8")); steps = int(os.environ.get("STEPS", "200"))
a, b = column_slab(W, 0, N)
o = torch.as_tensor(ro[:, a:b].copy(), device=dev); d = torch.as_tensor(rd[:, a:b].copy(), device=dev)
g = torch.as_tensor(scenes.upstream_grad(H, W), device=dev)[:, a:b].contiguous()
bg = torch.as_tensor(scenes.BG_DEFAULT, device=dev)
args = (t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg)
tr = ShardedTracer(); tr.cull_build = N >= 4


def loop(n):
    for _ in range(n):
        tr.forward(o, d, *args, cull_key="k"); tr.backward(*args, g)


loop(10); torch.cuda.synchronize()
t0 = time.perf_counter(); loop(steps); t_host = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(f"N={N}: host enqueue {t_host / steps * 1e3:.3f} ms per step, wall incl. the GPU {t_all / steps * 1e3:.3f} ms per step")
pr = cProfile.Profile(); pr.enable(); loop(steps); pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(18)
