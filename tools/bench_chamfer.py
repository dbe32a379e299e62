"""Times the Chamfer operator on a BASELINE-size LiDAR frame (64 x 2048 rays, prediction vs ground truth).

    python tools/bench_chamfer.py [--mode 0|1|2] [--steps K] [--cpu]

Prints one JSON line: forward / backward ms (HIP events on the launch stream), pair-evaluation rate of the
brute-force mode against the fp32 VALU peak, and (``--cpu``) the oracle's time on a bounded query sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidar_rt_amd import scenes  # noqa: E402
from lidar_rt_amd.chamfer3D import _C as chamfer_3D  # noqa: E402


def clouds(H, W, seed=1, drop=0.15):
    r = np.random.default_rng(seed)
    _, d = scenes.kitti_rays(H, W)
    d = d.reshape(-1, 3)
    depth = (8.0 + 30.0 * r.random(d.shape[0]) ** 2).astype(np.float32)
    keep = r.random(d.shape[0]) > drop
    gt = d * depth[:, None]
    pred = d * (depth * (1 + 0.02 * r.standard_normal(d.shape[0])).astype(np.float32))[:, None]
    return pred[keep][None].astype(np.float32), gt[keep][None].astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", type=int, default=2)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--H", type=int, default=64)
    ap.add_argument("--W", type=int, default=2048)
    ap.add_argument("--cpu", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    pa, pb = clouds(a.H, a.W)
    ta, tb = torch.as_tensor(pa, device=dev), torch.as_tensor(pb, device=dev)
    N, M = pa.shape[1], pb.shape[1]
    d1 = torch.empty(1, N, device=dev); d2 = torch.empty(1, M, device=dev)
    i1 = torch.empty(1, N, device=dev, dtype=torch.int32); i2 = torch.empty(1, M, device=dev, dtype=torch.int32)
    g1 = torch.full((1, N), 0.5 / N, device=dev); g2 = torch.full((1, M), 0.5 / M, device=dev)
    ga = torch.zeros(1, N, 3, device=dev); gb = torch.zeros(1, M, 3, device=dev)
    chamfer_3D.set_option("mode", a.mode)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb_ = 0.0
    for it in range(a.warmup + a.steps):
        ev[0].record(); chamfer_3D.forward(ta, tb, d1, d2, i1, i2)
        ev[1].record(); ga.zero_(); gb.zero_(); chamfer_3D.backward(ta, tb, ga, gb, g1, g2, i1, i2)
        ev[2].record(); torch.cuda.synchronize()
        if it >= a.warmup:
            tf += ev[0].elapsed_time(ev[1]); tb_ += ev[1].elapsed_time(ev[2])
    tf /= a.steps; tb_ /= a.steps
    res = {"op": "chamfer_3D", "mode": a.mode, "N": N, "M": M, "fwd_ms": round(tf, 4), "bwd_ms": round(tb_, 4),
           "points_per_s": round((N + M) / ((tf + tb_) * 1e-3), 1), "pairs": 2.0 * N * M}
    if a.mode == 0:
        # 9 VALU lane-ops per pair (3 sub, mul, 2 fma, cmp, 2 select); fp32 VALU peak 256 CU x 128 lanes x 2.4 GHz
        res["valu_frac"] = round(2.0 * N * M * 9 / (tf * 1e-3) / (256 * 128 * 2.4e9), 4)
    if a.cpu:
        from oracle import chamfer as och
        from oracle.oracle import num_threads
        n = min(N, 4096)
        t0 = time.perf_counter(); och.chamfer_forward(pa[:, :n], pb); dt = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": round((N + M) / (dt * N / n), 1), "unit": "points/s (fwd only, extrapolated x N/n)", "cores": num_threads(),
                               "kind": "port", "sample": f"{n} queries x {M} candidates + reverse, {dt:.2f} s"}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
