"""Helpers shared by the GPU parity tests (tests/ only)."""
import numpy as np
import torch

from lidar_rt_amd.diff_lidar_tracer import Tracer, TracingSettings

DEV = torch.device("cuda:0")
DEFAULT_OPTS = {"fwd_mode": 2, "bwd_mode": 2, "defer_colour": 1, "hit_cap": 256, "wg4_per_cu": 4, "c4_queue_limit": 1024, "c4_waves": 0, "slab0_mm": 24000, "learn_slab": 1, "reduce_mode": 2}     # the library defaults


def settings(bg, deg, mod=1.0):
    e = torch.empty(0, device=DEV)
    return TracingSettings(None, None, None, None, torch.as_tensor(np.asarray(bg, np.float32), device=DEV), mod, e, e,
                           deg, torch.zeros(3, device=DEV), False, False)


def run_hip(sc, o, d, deg, bg, dL=None, opts=None, mod=1.0, tracer=None, training=True):
    """Forward (+ backward) through the public Tracer API; returns numpy results."""
    tr = tracer or Tracer()
    if not training:
        tr.eval()
    for k, v in {**DEFAULT_OPTS, **(opts or {})}.items():       # the state is per device: reset what an earlier test set
        tr.optix_context.set_option(k, v)
    t = {k: torch.as_tensor(np.asarray(v, np.float32), device=DEV).requires_grad_(True) for k, v in sc.items()}
    ro, rd = torch.as_tensor(np.asarray(o, np.float32), device=DEV), torch.as_tensor(np.asarray(d, np.float32), device=DEV)
    tr.build_from_gaussians(t["means"], t["scales"], t["rotations"], t["opacities"], mod)
    out, acc = tr(ro, rd, None, t["means"], torch.zeros_like(t["means"]), shs=t["shs"], opacities=t["opacities"],
                  scales=t["scales"], rotations=t["rotations"], tracer_settings=settings(bg, deg, mod))
    res = {"out": None, "accum": None}
    if dL is not None:
        out.backward(torch.as_tensor(np.asarray(dL, np.float32), device=DEV))
        res["grads"] = {k: t[k].grad.detach().cpu().numpy() for k in ("means", "scales", "rotations", "opacities", "shs")}
    torch.cuda.synchronize()
    res["out"] = out.detach().cpu().numpy(); res["accum"] = acc.detach().cpu().numpy()
    return res


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def frac_outside(a, b, rtol, floor=1e-3):
    """Fraction of elements with |a-b| > rtol * max(|b|, floor * max|b|)."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-300)
    return float((np.abs(a - b) > rtol * np.maximum(np.abs(b), floor * scale)).mean())
