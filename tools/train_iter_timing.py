#!/usr/bin/env python3
"""Developer tool: wall time of whole training iterations (lidar_rt_amd.training.training_step) at S1M scale and the
share of the library's kernels in it (torch profiler, kernel names grouped)."""
import os, sys, time, types
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from lidar_rt_amd import scenes, training
from lidar_rt_amd.renderer import raytracing

dev = torch.device("cuda:0")
sc, ro, rd = scenes.s1m()
t = lambda a: torch.as_tensor(a, device=dev)
op = sc["opacities"]
def asset(noise):
    r = np.random.default_rng(1)
    a = training.GaussianAsset.from_tensors(t(sc["means"] + noise * r.normal(size=sc["means"].shape).astype(np.float32)),
                                            t(sc["shs"][:, :1]), t(sc["shs"][:, 1:]), t(np.log(sc["scales"])), t(sc["rotations"]),
                                            t(np.log(op / (1 - op))), extent=60.0)
    a.active_sh_degree = 3
    return a
opt = training.default_options()
opt.bvh_refit_interval = int(os.environ.get("REFIT", "0"))
bg = torch.tensor([0.0, 0.0, 1.0], device=dev)
frames = training.RangeFrames()
args = types.SimpleNamespace(dynamic=False, opt=opt, pipe=types.SimpleNamespace())
with torch.no_grad():
    pk = raytracing(0, [asset(0.0)], (t(ro), t(rd), torch.zeros(3, device=dev)), bg, args)
frames.add_frame(0, t(ro), t(rd), pk["depth"].squeeze(-1).detach(), pk["intensity"].squeeze(-1).detach(), pk["raydrop"].squeeze(-1) < 0.6)
scene = training.GaussianScene([asset(0.02)])
scene.training_setup(opt)
for it in range(1, 4):
    training.training_step(scene, frames, 0, it, opt, bg)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for it in range(4, 4 + n):
    training.training_step(scene, frames, 0, it, opt, bg)
torch.cuda.synchronize()
print(f"training iteration: {(time.perf_counter() - t0) / n * 1e3:.2f} ms wall (S1M, 1 asset, all losses, Adam step)")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for it in range(20, 23):
        training.training_step(scene, frames, 0, it, opt, bg)
    torch.cuda.synchronize()
rows = [(e.key, e.device_time_total / 3.0, e.count / 3) for e in prof.key_averages() if e.device_time_total > 0]
rows.sort(key=lambda r: -r[1])
tot = sum(r[1] for r in rows)
print(f"GPU kernel time per iteration: {tot / 1e3:.2f} ms")
for k, us, c in rows[:int(os.environ.get('ROWS', '22'))]:
    print(f"  {us:8.1f} us  x{c:4.1f}  {k[:90]}")
