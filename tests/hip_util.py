"""Helpers shared by the GPU parity tests (tests/ only)."""
import numpy as np
import torch

from lidar_rt_amd.diff_lidar_tracer import Tracer, TracingSettings

DEV = torch.device("cuda:0")
DEFAULT_OPTS = {"fwd_mode": 2, "bwd_mode": 3, "defer_colour": 1, "hit_cap": 256, "wg4_per_cu": 5, "c4_queue_limit": 1024, "c4_waves": 0, "slab0_mm": 100000, "root_nodes": 32, "learn_slab": 1, "spec_bwd": 1, "spec_margin": 65536, "own_sort": 2, "fused_tree": 1, "fused_hist": 1}     # the library defaults


def legacy_library() -> bool:
    """True when this process runs on the cross-check library (liblrt_hip_legacy.so, -DLRT_LEGACY: env LRT_HIP_LIB, set by
    tests/test_legacy_crosscheck_gpu.py for its subprocess): the retired kernel generations (bwd_mode 1 / 2, defer_colour 0, fused_tree 0 / 2)
    exist there and serve as independent implementations."""
    from lidar_rt_amd import _capi
    return _capi.has_legacy()


def is_legacy_mode(opts) -> bool:
    return opts.get("bwd_mode") in (1, 2) or opts.get("defer_colour") == 0 or opts.get("fused_tree", 1) != 1


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


# ---------------------------------------------------------------------------------------------------------------------
# Quantitative parity: the HIP path against the fp32 oracle, bounded by k x the algorithm's own noise floor (the fp32 oracle
# against the fp64 oracle on the same inputs), and the measured numbers written to gpurun_out/parity/<name>.json
# (tools/collect_parity.py turns those files into profiles/r02_parity.json = BASELINE.md section 6).
# Why k > 1: two float32 implementations differ from EACH OTHER by about twice what one of them differs from the
# fp64 result, and the differences are threshold events, not rounding noise: hits less than ~2 ulp(t) apart change places
# (only the colour channels see it), and a candidate on the edge of the reference's restart epsilon (t16 + 1e-5) is dropped by
# one implementation and kept by the other.  tests/tools/outlier_arbiter.py arbitrates such rays hit by hit against a
# brute-force float64 restatement of the raygen loop (27 of 40 examined rays: permutations of hits 0.3..5e-6 m apart, 13:
# restart-epsilon edges); tests/tools/t_noise_probe.py measures the hit distance itself: 0.52 ulp median error for the HIP
# path, 0.65 ulp for a float32 Moeller-Trumbore evaluation.  Measured ratios to the floor: 1..3.5 (profiles/r02_parity.json).
# Round 4: the gates follow the measurements (profiles/r03_parity.md, r04: largest ratios 2.1 on the fraction, 3.1 on L2) instead of 4 / 8.
FLOOR_K = 2.5            # on the fraction of elements beyond the tolerance (a count of events: robust)
FLOOR_K_L2 = 4.0         # on the relative L2 error (dominated by the one or two largest events of an image: heavy-tailed)
OUT_TOL, GRAD_TOL = 1e-4, 1e-3            # BASELINE.json north_star: 1e-4 relative on rendered channels, 1e-3 on gradients
# Round 5: the PRIMARY gate is arbitrated by the fp64 oracle -- "the HIP path is no farther from the exact answer than
# the reference's own fp32 arithmetic": for every channel and every gradient, HIP-vs-fp64 must stay within F64_K x the fp32-oracle-vs-fp64
# statistic (the floor).  The fraction of elements beyond the tolerance is a count of events and gets a 3-sigma counting allowance.  Events
# are RAY events (a candidate on the edge of the restart epsilon, an order swap ...): a rendered channel counts them one by one, a gradient
# tensor counts every element of the ~32 Gaussians composited behind the event -- its counts come in clusters of 32 x (components per row),
# and the counting noise of a clustered count is sqrt(cluster x n).  (S200k, tests/tools/parity_events.py: 10 restart-epsilon rays for HIP, 7
# for the fp32 oracle -- that difference of three rays IS the 331-against-222 Gaussians of d_opacities: profiles/r05_parity_events_grads_s200k.md.)
# The relative L2 error is dominated by the one or two largest events of an image (heavy-tailed); (1.1, 1.25) is the claim itself, scenes on
# which the measured L2 ratio does not support 1.25 pass their own factor and say so (tests/test_config_shapes_gpu.py).
F64_K = (1.1, 1.25)
GRAD_CLUSTER = 16          # Gaussians whose rows one ray event moves: the hits BEHIND the event, half of the K = 30 composited per ray (round 6: 32 before)


def parity_stats(got, ref, rtol, floor=1e-3):
    """max / p99 of the element-wise relative error (relative to max(|ref|, floor * max|ref|)), the fraction of elements
    beyond rtol, and the relative L2 error."""
    a = np.asarray(got, np.float64).reshape(-1); b = np.asarray(ref, np.float64).reshape(-1)
    if a.size == 0:
        return {"max_rel": 0.0, "p99_rel": 0.0, "frac_gt_tol": 0.0, "rel_l2": 0.0, "tol": rtol, "n": 0}
    scale = max(np.abs(b).max(), 1e-300)
    e = np.abs(a - b) / np.maximum(np.abs(b), floor * scale)
    return {"max_rel": float(e.max()), "p99_rel": float(np.quantile(e, 0.99)), "frac_gt_tol": float((e > rtol).mean()),
            "rel_l2": float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)), "tol": rtol, "n": int(a.size)}


OUT_CHANNELS = ((0, "intensity"), (1, "rayhit"), (2, "raydrop"), (3, "depth"), (4, "weight"), (8, "transmittance"))


def parity_report(name, hip, f32, f64=None, grads=("means", "scales", "rotations", "opacities", "shs"), extra=None, k=FLOOR_K, check=True, f64_k=F64_K):
    """Compare hip = {"out", "accum", "grads": {...}} with the fp32 oracle results f32 = (fw, bw) and, when the fp64 oracle
    results are given, assert every statistic within k x the f32-vs-f64 floor (or within the north-star tolerance itself where
    the floor is below it).  Writes the table to gpurun_out/parity/<name>.json and returns it."""
    import json, os
    fw, bw = f32
    rows = {}

    def add(label, got, ref, tol, ref64=None):
        st = parity_stats(got, ref, tol)
        a_ = np.asarray(ref)
        st["row_width"] = int(np.prod(a_.shape[1:])) if (label.startswith("grad.") and a_.ndim > 1) else 1
        if ref64 is not None:
            fl = parity_stats(ref, ref64, tol)
            st["floor"] = {q: fl[q] for q in ("max_rel", "p99_rel", "frac_gt_tol", "rel_l2")}
            h64 = parity_stats(got, ref64, tol)                    # the HIP path against the fp64 oracle, for the record
            st["vs_f64"] = {q: h64[q] for q in ("max_rel", "p99_rel", "frac_gt_tol", "rel_l2")}
        rows[label] = st

    for c, cname in OUT_CHANNELS:
        add(f"out.{cname}", hip["out"][..., c], fw["out"][..., c], OUT_TOL, None if f64 is None else f64[0]["out"][..., c])
    add("accum", hip["accum"], fw["accum"], OUT_TOL, None if f64 is None else f64[0]["accum"])
    if bw is not None and "grads" in hip:
        for g in grads:
            add(f"grad.{g}", hip["grads"][g].reshape(bw[g].shape), bw[g], GRAD_TOL, None if f64 is None else f64[1][g])
    rep = {"name": name, "k_frac": k, "k_l2": k * (FLOOR_K_L2 / FLOOR_K), "f64_k_frac": f64_k[0] if f64_k else None, "f64_k_l2": f64_k[1] if f64_k else None,
           "asserted": bool(check), "rows": rows, **(extra or {})}
    # ratios of the fp64-arbitrated gate, for the record (worst row of each statistic)
    if f64 is not None:
        rf = [(st["vs_f64"]["frac_gt_tol"] * st["n"] + 1.0) / (st["floor"]["frac_gt_tol"] * st["n"] + 1.0) for st in rows.values() if "vs_f64" in st]
        rl = [st["vs_f64"]["rel_l2"] / max(st["floor"]["rel_l2"], 1e-30) for st in rows.values() if "vs_f64" in st and st["floor"]["rel_l2"] > st["tol"] * 1e-2]
        rep["f64_gate_worst"] = {"frac_ratio": max(rf) if rf else None, "l2_ratio": max(rl) if rl else None}
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name + ".json"), "w") as f:
            json.dump(rep, f, indent=1)
    except OSError:
        pass
    bad = []
    for label, st in rows.items():
        if "floor" not in st:
            continue
        tol = st["tol"]
        # a statistic passes when it is within k x the floor; where the floor itself is far below the north-star tolerance
        # (small scenes) being within that tolerance is enough
        if st["frac_gt_tol"] > max(k * st["floor"]["frac_gt_tol"], 1e-3 if tol == OUT_TOL else 2e-3):
            bad.append((label, "frac_gt_tol", st["frac_gt_tol"], st["floor"]["frac_gt_tol"]))
        if st["rel_l2"] > max(k * (FLOOR_K_L2 / FLOOR_K) * st["floor"]["rel_l2"], 2 * tol):
            bad.append((label, "rel_l2", st["rel_l2"], st["floor"]["rel_l2"]))
    assert not (check and bad), f"{name}: beyond {k} x the f32-vs-f64 floor: {bad}"
    # ---- the fp64-arbitrated gate: HIP against the fp64 oracle, bounded by the fp32 oracle against the fp64 oracle
    bad64 = []
    if f64_k:
        for label, st in rows.items():
            if "vs_f64" not in st:
                continue
            tol, n = st["tol"], max(st["n"], 1)
            ev_floor = st["floor"]["frac_gt_tol"] * n
            cluster = 1.0 if label.startswith("out.") else GRAD_CLUSTER * st.get("row_width", 1)
            allow = (f64_k[0] * ev_floor + 3.0 * np.sqrt(cluster * (ev_floor + 1.0)) + 2.0 * cluster) / n      # no absolute slack beyond two events
            if st["vs_f64"]["frac_gt_tol"] > allow:
                bad64.append((label, "frac_gt_tol", st["vs_f64"]["frac_gt_tol"], st["floor"]["frac_gt_tol"]))
            if st["vs_f64"]["rel_l2"] > max(f64_k[1] * st["floor"]["rel_l2"], 2 * tol):
                bad64.append((label, "rel_l2", st["vs_f64"]["rel_l2"], st["floor"]["rel_l2"]))
    assert not (check and bad64), (f"{name}: the HIP path is farther from the fp64 oracle than {f64_k} x the fp32 oracle is "
                                   f"(label, statistic, HIP-vs-f64, f32-vs-f64): {bad64}")
    return rep
