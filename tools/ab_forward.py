#!/usr/bin/env python3
"""Developer tool: A/B of forward-kernel variants on S1M in ONE process / GPU call.

    python tools/ab_forward.py "fwd_mode=2" "fwd_mode=3" "fwd_mode=3,wg4_per_cu=5"

For every option set: forward image / accum compared bit for bit with the first set, HIP-event time of the trace kernel
(lrt_get_timing: build / forward region / backward), traversal counters.  LRT_AB_WORKLOAD=s200k|s10k picks a smaller scene,
LRT_AB_BWD=1 adds the backward, LRT_AB_REPS=n the repetitions (default 30).
"""
import os, sys, json
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from lidar_rt_amd import scenes
from lidar_rt_amd.parallel import HipBackend

dev = torch.device("cuda:0")
wl = os.environ.get("LRT_AB_WORKLOAD", "s1m")
if wl == "s1m": sc, ro, rd = scenes.s1m()
elif wl == "s200k": sc = scenes.make_scene(200_000, radius_scale=0.5); ro, rd = scenes.kitti_rays(32, 512)
elif wl == "w4m": sc, ro, rd = scenes.waymo_dynamic_4m()
else: sc, ro, rd = scenes.s10k()
H, W = ro.shape[:2]
t = {k: torch.as_tensor(v, device=dev) for k, v in sc.items()}
ray_o, ray_d = torch.as_tensor(ro, device=dev), torch.as_tensor(rd, device=dev)
bg = torch.as_tensor(scenes.BG_DEFAULT, device=dev)
dL = torch.as_tensor(scenes.upstream_grad(H, W), device=dev)
reps = int(os.environ.get("LRT_AB_REPS", "30"))
with_bwd = os.environ.get("LRT_AB_BWD", "0") == "1"
ref = None
be = HipBackend()
DEFAULTS = {"fwd_mode": 2, "wg4_per_cu": 4, "c4_waves": 0, "root_nodes": 32, "slab0_mm": 100000, "learn_slab": 1}
for spec in (sys.argv[1:] or ["fwd_mode=2", "fwd_mode=3"]):
    opts = dict(DEFAULTS)
    for kv in spec.split(","):
        if kv: opts[kv.split("=")[0]] = int(kv.split("=")[1])
    for k, v in opts.items(): be.state.set_option(k, v)
    args = (ray_o, ray_d, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg)
    try:
        for _ in range(3):
            be.build(t["means"], t["scales"], t["rotations"], t["opacities"])
            out, acc = be.forward(*args)
            if with_bwd: be.backward(*args, out, dL)
        torch.cuda.synchronize()
        be.state.enable_timing(True)
        for _ in range(reps):
            be.build(t["means"], t["scales"], t["rotations"], t["opacities"])
            out, acc = be.forward(*args)
            if with_bwd: g = be.backward(*args, out, dL)
        tm = be.state.get_timing(dev); be.state.enable_timing(False)
        be.state.enable_stats(True)
        out, acc = be.forward(*args)
        st = be.state.get_stats(dev); be.state.enable_stats(False)
        be.state.check(dev, wait=True)
    except Exception as ex:
        print(json.dumps({"opts": spec, "error": str(ex)[:300]}), flush=True); continue
    o_, a_ = out.cpu().numpy(), acc.cpu().numpy()
    res = {"opts": spec, "fwd_ms": tm["fwd"][0] / max(tm["fwd"][1], 1), "build_ms": tm["build"][0] / max(tm["build"][1], 1),
           "bwd_ms": tm["bwd"][0] / max(tm["bwd"][1], 1) if with_bwd else None,
           "sum_out": float(np.abs(o_.astype(np.float64)).sum()), "sum_acc": float(a_.astype(np.float64).sum()),
           "stats": {k: int(v) for k, v in st.items()}}
    if ref is None: ref = (o_, a_)
    else:
        res["out_bit_equal"] = bool(np.array_equal(o_, ref[0])); res["max_abs_out_diff"] = float(np.abs(o_ - ref[0]).max())
        res["acc_rel_l2"] = float(np.linalg.norm(a_.astype(np.float64) - ref[1]) / np.linalg.norm(ref[1].astype(np.float64)))
    print(json.dumps(res), flush=True)
