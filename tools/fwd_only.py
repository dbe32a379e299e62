#!/usr/bin/env python3
"""Developer tool: run build + N forward (+backward) traces of S1M for profiling under rocprofv3."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from lidar_rt_amd import scenes
from lidar_rt_amd.parallel import HipBackend
dev = torch.device("cuda:0")
sc, ro, rd = scenes.s1m()
t = {k: torch.as_tensor(v, device=dev) for k, v in sc.items()}
ray_o, ray_d = torch.as_tensor(ro, device=dev), torch.as_tensor(rd, device=dev)
bg = torch.as_tensor(scenes.BG_DEFAULT, device=dev)
dL = torch.as_tensor(scenes.upstream_grad(64, 2048), device=dev)
be = HipBackend()
n = int(os.environ.get("N", "3"))
for _ in range(n):
    be.build(t["means"], t["scales"], t["rotations"], t["opacities"])
    out, acc = be.forward(ray_o, ray_d, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg)
    if "--bwd" in sys.argv:
        be.backward(ray_o, ray_d, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg, out, dL)
torch.cuda.synchronize()
print("done")
