#!/usr/bin/env python3
"""Developer tool: the S1M step with hit weights complete at the forward (A) and written by the backward (B, option deferred_accum), in alternating
blocks on one GPU: wall time per step and the library's region timers of every block.  ONLY=A / ONLY=B: one configuration (for a kernel trace)."""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from lidar_rt_amd import scenes
from lidar_rt_amd.parallel import ShardedTracer

dev = torch.device("cuda:0")
sc, ro, rd = scenes.s1m()
t = {k: torch.as_tensor(v, device=dev) for k, v in sc.items()}
H, W = ro.shape[:2]
ro, rd = torch.as_tensor(ro, device=dev), torch.as_tensor(rd, device=dev)
dL = torch.as_tensor(scenes.upstream_grad(H, W), device=dev)
bg = torch.as_tensor(scenes.BG_DEFAULT, device=dev)
only = os.environ.get("ONLY", "")
trs = {k: ShardedTracer(deferred_accum=(k == "B")) for k in ("A", "B") if only in ("", k)}
for tr in trs.values():
    for kv in os.environ.get("LRT_OPTS", "").split(","):
        if kv: tr.backend.state.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    tr.backend.state.set_option("timing_every", 8)


def block(tr, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        tr.forward(ro, rd, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg, cull_key="f")
        tr.backward(t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg, dL)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for tr in trs.values():
    block(tr, 20)
N = int(os.environ.get("STEPS", "400"))
for rep in range(int(os.environ.get("REPS", "3"))):
    for k, tr in trs.items():
        tr.backend.state.enable_timing(True); tr.backend.state.get_timing(dev)
        ms = block(tr, N)
        tm = tr.backend.state.get_timing(dev); tr.backend.state.enable_timing(False)
        ms_plain = block(tr, N)
        f = lambda q: tm[q][0] / max(tm[q][1], 1)
        print(f"{k} rep {rep}: {ms_plain:.4f} ms/step = {H * W / ms_plain / 1e3:.1f} M rays/s  (with region timers {ms:.4f}: build {f('build'):.3f} fwd {f('fwd'):.3f} bwd {f('bwd'):.3f})", flush=True)
