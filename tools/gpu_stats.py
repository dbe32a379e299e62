#!/usr/bin/env python3
"""Developer tool: per-kernel traversal statistics of one S1M forward (tile clock sum/max etc.)."""
import os, sys, json
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from lidar_rt_amd import scenes
from lidar_rt_amd.parallel import HipBackend
dev = torch.device("cuda:0")
sc, ro, rd = scenes.s1m() if "--s200k" not in sys.argv else (scenes.make_scene(200_000, radius_scale=0.5), *scenes.kitti_rays(32, 512))
t = {k: torch.as_tensor(v, device=dev) for k, v in sc.items()}
ray_o, ray_d = torch.as_tensor(ro, device=dev), torch.as_tensor(rd, device=dev)
bg = torch.as_tensor(scenes.BG_DEFAULT, device=dev)
be = HipBackend()
for tw in [int(a.split("=")[1]) for a in sys.argv if a.startswith("--tw=")] or [16]:
    be.state.set_option("tile_w", tw)
    be.build(t["means"], t["scales"], t["rotations"], t["opacities"])
    for _ in range(2):
        out, acc = be.forward(ray_o, ray_d, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg)
    torch.cuda.synchronize()
    be.state.enable_stats(True); be.state.enable_timing(True)
    out, acc = be.forward(ray_o, ray_d, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg)
    st = be.state.get_stats(dev); tm = be.state.get_timing(dev)
    be.state.enable_stats(False); be.state.enable_timing(False)
    ntiles = (ro.shape[0] * ro.shape[1] + 63) // 64
    print(f"tile_w={tw} fwd kernel ms={tm['fwd'][0]:.3f} stats={st}")
    print(f"   per tile: passes {st['passes']/ntiles:.2f} nodes {st['nodes_visited']/ntiles:.0f} prims {st['prims_tested']/ntiles:.0f} "
          f"inserts {st['wave_inserts']/ntiles:.0f}; tile clk avg {st['tile_clk_sum']/ntiles:.0f} max {st['tile_clk_max']} (100MHz ticks: avg {st['tile_clk_sum']/ntiles/100:.1f} us, max {st['tile_clk_max']/100:.1f} us)")
