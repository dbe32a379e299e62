#!/usr/bin/env python3
"""Developer tool: the dense translucent scene of tests/test_hip_parity.py through every forward mode, against the oracle."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, REPO)
from lidar_rt_amd import scenes
from tests.test_hip_parity import oracle_run
from tests.hip_util import run_hip, rel_l2

sc, o, d = scenes.dense_translucent()
dL = scenes.upstream_grad(4, 48, seed=2)
fw, bw = oracle_run(sc, o, d, 3, scenes.BG_DEFAULT, dL)
fw64, bw64 = oracle_run(sc, o, d, 3, scenes.BG_DEFAULT, dL, prec="f64")
GR = ("means", "scales", "rotations", "opacities", "shs")
print("oracle grads f32 vs f64:", {k: f"{rel_l2(bw[k], bw64[k]):.1e}" for k in GR})
print("oracle f32 vs f64:", rel_l2(fw["out"], fw64["out"]), "n_comp mean", fw["n_comp"].mean(), "max", fw["n_comp"].max(), "cand max", fw["n_cand"].max())
for name, opts in (("legacy k_trace", {"fwd_mode": 0, "bwd_mode": 0}), ("k_fwd_cr", {"fwd_mode": 1}), ("k_fwd_cr nodefer", {"fwd_mode": 1, "defer_colour": 0}),
                   ("cr4 nw4", {"fwd_mode": 2, "c4_waves": 4}), ("cr4 nw8", {"fwd_mode": 2, "c4_waves": 8}), ("cr4 cap1024", {"fwd_mode": 2, "hit_cap": 1024}), ("cr4 cap1024 atomics", {"fwd_mode": 2, "hit_cap": 1024, "bwd_mode": 1}), ("cr4 cap1024 red1", {"fwd_mode": 2, "hit_cap": 1024, "reduce_mode": 1}),
                   ("cr4 slab 1m", {"fwd_mode": 2, "slab0_mm": 1000, "hit_cap": 1024})):
    h = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL, opts=opts)
    e = np.abs(h["out"] - fw["out"]).reshape(-1, 9)
    worst = np.argsort(-e.max(1))[:3]
    print(f"{name:18s} out vs oracle {rel_l2(h['out'], fw['out']):.2e}  vs f64 {rel_l2(h['out'], fw64['out']):.2e}  worst rays {worst.tolist()} err {e.max(1)[worst].round(4).tolist()} ncomp {fw['n_comp'].reshape(-1)[worst].tolist()}",
          " grads vs f64 oracle", {k: f"{rel_l2(h['grads'][k].reshape(bw64[k].shape), bw64[k]):.1e}" for k in GR})
