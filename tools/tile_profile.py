#!/usr/bin/env python3
"""Developer tool: per-tile clock/slab/node/prim profile of the collect&resolve forward on S1M."""
import os, sys, ctypes as C
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from lidar_rt_amd import scenes
from lidar_rt_amd.parallel import HipBackend
dev = torch.device("cuda:0")
sc, ro, rd = scenes.s1m()
t = {k: torch.as_tensor(v, device=dev) for k, v in sc.items()}
ray_o, ray_d = torch.as_tensor(ro, device=dev), torch.as_tensor(rd, device=dev)
bg = torch.as_tensor(scenes.BG_DEFAULT, device=dev)
be = HipBackend()
be.state.set_option("debug_rays", 16384)   # >= 8 floats per tile
for kv in os.environ.get("LRT_OPTS", "").split(","):
    if kv: be.state.set_option(kv.split("=")[0], int(kv.split("=")[1]))
NT = 8192 * (2 if "s8" in os.environ.get("LRT_HIP_LIB", "") else 1)
be.build(t["means"], t["scales"], t["rotations"], t["opacities"])
for _ in range(2):
    be.forward(ray_o, ray_d, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg)
be.state.enable_stats(True); be.state.enable_timing(True)
be.forward(ray_o, ray_d, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg)
st = be.state.get_stats(dev); tm = be.state.get_timing(dev)
idx, h = be.state.handle(dev)
buf = np.empty((2, 16 * NT // 8192, 512, 4), np.float32)
be.state._lib.lrt_debug_read(h, 4, buf.ctypes.data_as(C.c_void_p), buf.nbytes, None)
print("fwd ms", tm["fwd"], "stats", st)
buf2 = buf[1]; buf = buf[0]
clk = buf[..., 0] / 100.0   # us
print("phase A us by row:", np.round(buf2[..., 0].mean(1) / 100, 0).tolist())
print("phase B us by row:", np.round(buf2[..., 1].mean(1) / 100, 0).tolist())
print("node batches by row:", np.round(buf2[..., 2].mean(1), 1).tolist())
print("leaf batches by row:", np.round(buf2[..., 3].mean(1), 1).tolist())
print("tile clk us: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f" % (clk.mean(), np.percentile(clk, 50), np.percentile(clk, 90), np.percentile(clk, 99), clk.max()))
for name, k in (("clk_us", None), ("slabs", 1), ("nodes", 2), ("prim_rounds", 3)):
    v = clk if k is None else buf[..., k]
    print(name, "by tile row:", np.round(v.mean(1), 1).tolist())
    print(name, "max by tile row:", np.round(v.max(1), 1).tolist())

if os.environ.get("C4_PROF"):
    raw = np.empty(16384 * 64, np.float32)
    be.state._lib.lrt_debug_read(h, 4, raw.ctypes.data_as(C.c_void_p), raw.nbytes, None)
    pr = raw[8 * NT: 8 * NT + 16 * NT].reshape(NT, 16)
    names = ["first select+commit", "select_issue(next)", "process leaves", "process nodes", "round barrier", "push counters + late select", "commit (vmcnt wait + LDS stores)",
             "phase A tail (waitcnt, barrier, overflow test)", "slab prologue (barriers, flags)", "phase B: prologue, per-ray sums + state write, tail",
             "phase B: list loads waited for, exp, stage t in LDS", "phase B: rank pass", "phase B: rank check + scatter by rank", "phase B: ray state, tie refine, chunk rule",
             "phase B: alpha, prefix product, stop test", "phase B: composite (weights, accum atomics, hit record)"]
    tot = pr.sum()
    print("wave-0 cycles per tile by segment (mean; share):")
    for k, n in enumerate(names):
        print("  %-48s %9.0f  %5.1f %%" % (n, pr[:, k].mean(), 100 * pr[:, k].sum() / tot))
    print("  total cycles per tile %.0f (%.1f us at 2.4 GHz)" % (pr.sum(1).mean(), pr.sum(1).mean() / 2400))
