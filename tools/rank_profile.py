#!/usr/bin/env python3
"""Developer tool: every rank of an N-way split, one at a time on one GPU: build / forward / backward ms and the forward's tile statistics
(tiles, sum and max of the per-tile clocks at 100 MHz, candidates, composited hits).  env WORKLOAD=s1m|waymo4m, SLAB_N=8"""
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
args = (t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg)
N = int(os.environ.get("SLAB_N", "8"))
tr = ShardedTracer(); tr.cull_build = N >= 4
be = tr.backend
for kv in os.environ.get("LRT_OPTS", "").split(","):
    if kv: be.state.set_option(kv.split("=")[0], int(kv.split("=")[1]))
for r in range(N):
    a, b = column_slab(W, r, N)
    o = torch.as_tensor(ro[:, a:b].copy(), device=dev); d = torch.as_tensor(rd[:, a:b].copy(), device=dev); g = dL[:, a:b].contiguous()
    be.state.enable_timing(True)
    for it in range(7):
        if it == 3: be.state.get_timing(dev)
        tr.forward(o, d, *args, cull_key=(N, r)); tr.backward(*args, g)
    torch.cuda.synchronize()
    tm = be.state.get_timing(dev); f = lambda k: tm[k][0] / max(tm[k][1], 1)
    be.state.enable_timing(False)
    be.state.enable_stats(True)
    tr.forward(o, d, *args, cull_key=(N, r)); torch.cuda.synchronize()
    s = be.state.get_stats(dev)
    be.state.enable_stats(False)
    tiles = ((b - a + 7) // 8) * ((H + 1) // 2)
    print(f"{wl} N={N} rank {r}: cols {b - a}  build {f('build'):.3f} fwd {f('fwd'):.3f} (colour {f('colour'):.3f}) bwd {f('bwd'):.3f} ms | tiles {tiles}  tile clocks: sum {s['tile_clk_sum'] / 100:.0f} us"
          f"  max {s['tile_clk_max'] / 100:.0f} us  mean {s['tile_clk_sum'] / 100 / max(tiles, 1):.1f} us | candidates {s['candidates']}  composited {s['composited']}  passes {s['passes']}")
