import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidar_rt_amd import scenes
from lidar_rt_amd.parallel import HipBackend, GradLayout
dev = torch.device("cuda:0")
P = 4_000_000
t0 = time.time(); sc = scenes.make_scene(P, seed=5); print("scene", round(time.time() - t0, 1), "s")
o, d = scenes.kitti_rays(64, 2048)
t = {k: torch.as_tensor(v, device=dev) for k, v in sc.items()}
ro, rd = torch.as_tensor(o, device=dev), torch.as_tensor(d, device=dev)
dL = torch.as_tensor(scenes.upstream_grad(64, 2048), device=dev); bg = torch.as_tensor(scenes.BG_DEFAULT, device=dev)
be = HipBackend(); lay = GradLayout(P, 16, dev)
grads = {k: lay.views[k] for k in ("means", "scales", "rotations", "opacities", "shs")}
be.state.enable_timing(True); be.state.enable_stats(True)
for it in range(4):
    if it == 1: be.state.get_timing(dev)
    be.build(t["means"], t["scales"], t["rotations"], t["opacities"])
    out, acc = be.forward(ro, rd, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg)
    be.backward(ro, rd, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg, out, dL, grads_out=grads)
torch.cuda.synchronize()
be.state.check(dev)
tm = be.state.get_timing(dev); f = lambda k: tm[k][0] / max(tm[k][1], 1)
print(f"P=4M: build {f('build'):.3f} fwd {f('fwd'):.3f} bwd {f('bwd'):.3f} ms; stats", be.state.get_stats(dev))
print("energy check", float((out[..., 4] + out[..., 8] - 1).abs().max()), "mem GB", torch.cuda.max_memory_allocated() / 1e9)
