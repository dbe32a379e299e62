#!/usr/bin/env python3
"""Developer tool: per-rank compute time of an N-way azimuth split, measured on ONE GPU by tracing a single slab
(rank 0's columns) of the S1M frame: build / forward / backward HIP-event times for N = 1, 2, 4, 8."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from lidar_rt_amd import scenes
from lidar_rt_amd.parallel import HipBackend, column_slab, GradLayout

dev = torch.device("cuda:0")
sc, ro, rd = scenes.s1m()
t = {k: torch.as_tensor(v, device=dev) for k, v in sc.items()}
H, W = ro.shape[:2]
dL = torch.as_tensor(scenes.upstream_grad(H, W), device=dev)
bg = torch.as_tensor(scenes.BG_DEFAULT, device=dev)
be = HipBackend()
for kv in os.environ.get("LRT_OPTS", "").split(","):
    if kv: be.state.set_option(kv.split("=")[0], int(kv.split("=")[1]))
lay = GradLayout(1000000, 16, dev)
grads = {k: lay.views[k] for k in ("means", "scales", "rotations", "opacities", "shs")}
for N in (1, 2, 4, 8):
    for r in sorted({0, N // 2}):
        a, b = column_slab(W, r, N)
        o = torch.as_tensor(ro[:, a:b].copy(), device=dev); d = torch.as_tensor(rd[:, a:b].copy(), device=dev)
        g = dL[:, a:b].contiguous()
        be.state.enable_timing(True)
        for it in range(6):
            if it == 2: be.state.get_timing(dev)          # drop the warm-up samples
            be.build(t["means"], t["scales"], t["rotations"], t["opacities"], cull_rays=(o, d) if (N >= 3 and os.environ.get("CULL", "0") == "1") else None)
            out, acc = be.forward(o, d, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg)
            be.backward(o, d, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg, out, g, grads_out=grads)
        torch.cuda.synchronize()
        tm = be.state.get_timing(dev)
        f = lambda k: tm[k][0] / max(tm[k][1], 1)
        kept = be.state.built_count(dev)
        print(f"N={N} rank {r}: cols {b - a:4d}  kept {kept:7d}  build {f('build'):.3f}  fwd {f('fwd'):.3f}  bwd {f('bwd'):.3f}  sum {f('build') + f('fwd') + f('bwd'):.3f} ms")
