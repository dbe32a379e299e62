#!/usr/bin/env python3
"""First-light diagnostics on the GPU box: HIP tracer vs CPU oracle on tiny, S10k and (optionally) S1M
scenes, with timings and traversal statistics.  Writes gpurun_out/probe.json.  (Developer tool.)"""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from lidar_rt_amd import scenes                                   # noqa: E402
from lidar_rt_amd.diff_lidar_tracer import Tracer, TracingSettings  # noqa: E402
from oracle import oracle                                           # noqa: E402

dev = torch.device("cuda:0")


def settings(bg, deg):
    e = torch.empty(0, device=dev)
    return TracingSettings(None, None, None, None, torch.as_tensor(bg, dtype=torch.float32, device=dev), 1.0,
                           e, e, deg, torch.zeros(3, device=dev), False, False)


def run_hip(tr, sc, o, d, deg, bg, dL=None, reps=1):
    t = {k: torch.as_tensor(v, device=dev).requires_grad_(k != "none") for k, v in sc.items()}
    ro, rd = torch.as_tensor(o, device=dev), torch.as_tensor(d, device=dev)
    ts = settings(bg, deg)
    res = {}
    for rep in range(reps):
        for v in t.values():
            v.grad = None
        torch.cuda.synchronize(); t0 = time.time()
        tr.build_from_gaussians(t["means"], t["scales"], t["rotations"], t["opacities"])
        torch.cuda.synchronize(); t1 = time.time()
        out, acc = tr(ro, rd, None, t["means"], torch.zeros_like(t["means"]), shs=t["shs"], opacities=t["opacities"],
                      scales=t["scales"], rotations=t["rotations"], tracer_settings=ts)
        torch.cuda.synchronize(); t2 = time.time()
        if dL is not None:
            out.backward(torch.as_tensor(dL, device=dev))
            torch.cuda.synchronize()
        t3 = time.time()
        res["t_build"], res["t_fwd"], res["t_bwd"] = t1 - t0, t2 - t1, t3 - t2
    res["out"] = out.detach().cpu().numpy(); res["accum"] = acc.detach().cpu().numpy()
    if dL is not None:
        res["grads"] = {k: t[k].grad.detach().cpu().numpy() for k in ("means", "scales", "rotations", "opacities", "shs")}
    return res


def cmp(name, a, b, report):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    err = np.abs(a - b)
    scale = max(np.abs(b).max(), 1e-30)
    rel_el = err / np.maximum(np.abs(b), 1e-3 * scale)
    report[name] = {"max_abs": float(err.max()), "ref_max": float(scale), "normwise": float(err.max() / scale),
                    "rel_p999": float(np.percentile(rel_el, 99.9)), "rel_max": float(rel_el.max()),
                    "frac_gt_1e-4": float((rel_el > 1e-4).mean()), "frac_gt_1e-3": float((rel_el > 1e-3).mean()),
                    "nan": int(np.isnan(a).sum())}
    print(f"  {name:12s} normwise={err.max()/scale:.3e} rel_p99.9={report[name]['rel_p999']:.3e} "
          f"rel_max={rel_el.max():.3e} frac>1e-4={report[name]['frac_gt_1e-4']:.2e} nan={report[name]['nan']}", flush=True)


def case(tr, name, sc, o, d, deg, bg, dL, report, reps=1, prec="f32"):
    print(f"== {name}: P={sc['means'].shape[0]} rays={o.shape[0]}x{o.shape[1]} deg={deg}", flush=True)
    rep = report.setdefault(name, {})
    tr.optix_context.enable_stats(True)
    h = run_hip(tr, sc, o, d, deg, bg, dL, reps)
    st = tr.optix_context.get_stats()
    nr = o.shape[0] * o.shape[1]
    rep["hip_stats_per_call"] = st
    rep["timing_s"] = {k: h[k] for k in ("t_build", "t_fwd", "t_bwd")}
    print("  hip timing build/fwd/bwd (s):", h["t_build"], h["t_fwd"], h["t_bwd"], " stats:", st, flush=True)
    t0 = time.time()
    orc = oracle.Oracle(sc["means"], sc["scales"], sc["rotations"], sc["opacities"], prec)
    fw = orc.forward(o, d, sc["shs"], deg, bg, stats=True)
    t1 = time.time()
    rep["oracle_fwd_s"] = t1 - t0
    rep["C_mean"] = float(fw["n_cand"].mean()); rep["K_mean"] = float(fw["n_comp"].mean())
    print(f"  oracle fwd {t1-t0:.2f}s C={rep['C_mean']:.2f} K={rep['K_mean']:.2f}", flush=True)
    cmp("out", h["out"], fw["out"], rep)
    for ch, nm in ((0, "intensity"), (3, "depth"), (8, "T")):
        cmp("out_" + nm, h["out"][..., ch], fw["out"][..., ch], rep)
    cmp("accum", h["accum"], fw["accum"], rep)
    bad = np.argwhere(np.abs(h["out"] - fw["out"]).max(-1) > 1e-3 * np.abs(fw["out"]).max())
    rep["n_bad_rays"] = int(len(bad))
    if len(bad):
        print("  bad rays (first 5):", bad[:5].tolist())
        for hh, ww in bad[:3]:
            print("   hip", h["out"][hh, ww].tolist()); print("   orc", fw["out"][hh, ww].tolist(),
                  "ncand", int(fw["n_cand"][hh, ww]), "ncomp", int(fw["n_comp"][hh, ww]))
    if dL is not None:
        t0 = time.time()
        bw = orc.backward(o, d, sc["shs"], deg, bg, fw["out"], dL)
        rep["oracle_bwd_s"] = time.time() - t0
        print(f"  oracle bwd {rep['oracle_bwd_s']:.2f}s", flush=True)
        for k in ("means", "scales", "rotations", "opacities", "shs"):
            cmp("d_" + k, h["grads"][k], bw[k], rep)
    return rep


def main():
    full = "--s1m" in sys.argv
    report = {"device": torch.cuda.get_device_name(0)}
    print(report, flush=True)
    tr = Tracer()
    tr.optix_context.set_option("bwd_mode", int(os.environ.get("LRT_BWD_MODE", "2")))
    bg = scenes.BG_DEFAULT
    # tiny
    sc = {k: v.copy() for k, v in scenes.make_scene(64, seed=3, radius_scale=0.2).items()}
    o, d = scenes.kitti_rays(4, 16)
    case(tr, "tiny64", sc, o, d, 3, bg, scenes.upstream_grad(4, 16), report)
    sc, o, d = scenes.s10k()
    case(tr, "s10k", sc, o, d, 3, bg, scenes.upstream_grad(16, 256), report, reps=2)
    case(tr, "s10k_deg1_bg0", sc, o, d, 1, np.zeros(3, np.float32), scenes.upstream_grad(16, 256), report)
    sc = scenes.make_scene(200_000, radius_scale=0.5)
    o, d = scenes.kitti_rays(32, 512)
    case(tr, "s200k", sc, o, d, 3, bg, scenes.upstream_grad(32, 512), report, reps=2)
    if full:
        sc, o, d = scenes.s1m()
        case(tr, "s1m", sc, o, d, 3, bg, scenes.upstream_grad(64, 2048), report, reps=3)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "probe.json"), "w") as f:
        json.dump(report, f, indent=1)
    print("probe done")


if __name__ == "__main__":
    main()
