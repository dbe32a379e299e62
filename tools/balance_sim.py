#!/usr/bin/env python3
"""Developer tool: does ShardedTracer's slab balancer converge on REAL per-rank times?  One GPU plays the N ranks of an azimuth split one
after the other (each rank's whole step on its slab: culled build + forward + backward incl. the prezero bookkeeping, HIP events around
it), the N times go into `ShardedTracer._rebalance` -- the code the ranks run on the gathered times -- and the new edges are measured again.
Prints, per round, the edges, the per-rank times and max / mean (the step of an N-GPU job is the max).
env: WORKLOAD=s1m|waymo4m, SLAB_N="4,8", ROUNDS=6"""
import os, sys
os.environ.setdefault("LRT_PREZERO", "force")
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from lidar_rt_amd import scenes
from lidar_rt_amd.parallel import ShardedTracer, column_slab

dev = torch.device("cuda:0")
wl = os.environ.get("WORKLOAD", "s1m")
sc, ro, rd = scenes.waymo_dynamic_4m() if wl == "waymo4m" else scenes.s1m()
t = {k: torch.as_tensor(v, device=dev) for k, v in sc.items()}
H, W = ro.shape[:2]
dL = torch.as_tensor(scenes.upstream_grad(H, W), device=dev)
bg = torch.as_tensor(scenes.BG_DEFAULT, device=dev)
ro_t, rd_t = torch.as_tensor(ro, device=dev), torch.as_tensor(rd, device=dev)
args = (t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg)


def rank_time(tr, a, b, n_it=6, n_warm=3):
    o = ro_t[:, a:b].contiguous(); d = rd_t[:, a:b].contiguous(); g = dL[:, a:b].contiguous()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(n_it):
        if it == n_warm: ev[0].record()
        tr.forward(o, d, *args, cull_key=(a, b))
        tr.backward(*args, g)
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / (n_it - n_warm)


for N in tuple(int(x) for x in os.environ.get("SLAB_N", "4,8").split(",")):
    tr = ShardedTracer()                                  # plays one rank at a time (world 1); only its balancer state is driven as for N ranks
    tr.cull_build = N >= 4
    bal = ShardedTracer.__new__(ShardedTracer); bal.world = N; bal.rank = 0; bal._edges = None; bal._edges_key = None; bal._times = None
    edges = [column_slab(W, r, N)[0] for r in range(N)] + [W]
    print(f"{wl}, N={N}: per-rank whole step in ms, one rank at a time on one GPU")
    for rnd in range(int(os.environ.get("ROUNDS", "6"))):
        times = [rank_time(tr, edges[r], edges[r + 1]) for r in range(N)]
        print(f"  round {rnd}: max {max(times):.3f}  mean {sum(times) / N:.3f}  max/mean {max(times) / (sum(times) / N):.3f}   widths {[edges[r + 1] - edges[r] for r in range(N)]}"
              f"   times {[round(x, 3) for x in times]}")
        bal._edges = list(edges); bal._edges_key = (W, N); bal._times = times
        bal._rebalance(W)
        edges = list(bal._edges)
