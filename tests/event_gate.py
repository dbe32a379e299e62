"""The fp64-arbitrated parity gate with the threshold EVENTS taken out by name (tests/ only).

Two float32 implementations of the reference's raygen loop cannot agree on every ray: a candidate within a few ulp of the restart window
(t16 + 1e-5, forward.cu:282-291), two hits closer than fp32 resolves, an alpha on the 1/255 threshold or a transmittance on the 1e-4 stop
decide one way in one implementation and the other way in the other -- for the reference's own arithmetic no less than for ours.  One such
event changes ONE ray's compositing sequence and, through dL/dalpha, the gradient rows of every Gaussian behind it; the relative L2 error of
an image or a gradient tensor is carried by one to three of them.  Round 5 bounded that L2 by a factor per scene read off its own
measurement.  This module replaces the factor by an argument:

  1. the composited-Gaussian SEQUENCE of every ray is compared -- the HIP path's (its hit record), the fp32 oracle's and the fp64 oracle's
     (their event traces).  A ray whose sequence leaves the fp64 oracle's is an EVENT ray;
  2. the HIP path may not have more event rays than the reference's own fp32 arithmetic (the fp32 oracle) + 3 sigma of counting noise;
  3. every HIP event ray must be CERTIFIED as a knife edge by the brute-force float64 restatement (oracle/bruteforce.py: every quad
     intersected analytically, no tree, shares no code with the C oracle): at the depth where the sequences part there must be a candidate
     within 2e-6 (relative) of a restart point, or two candidates closer than 1e-6 (relative), or an alpha within 5e-4 (relative) of 1/255, or
     a quad met within 2e-4 (relative) of its rim, or a hit within 1e-6 of the 0.2 m threshold; a 1e-4 transmittance stop one hit apart is
     certified from the HIP path's OWN recorded alphas (its transmittance differs from the exact one by x, every alpha within 2e-3, and the exact
     stop test lies within 2 x + 1e-5 of the threshold).  A sequence that parts from the fp64 oracle's anywhere else is a BUG and fails the test;
  4. with the event rays of BOTH fp32 implementations masked (upstream gradient zero, image rows taken from the fp64 oracle) every channel and
     every gradient must satisfy the claim itself -- HIP no farther from the fp64 oracle than 1.1 x (fraction beyond tolerance) / 1.25 x
     (relative L2) the fp32 oracle is -- with no scene-specific factor and no "2 x tolerance" escape.
"""
import ctypes as C
import json
import os

import numpy as np
import torch

from oracle import oracle
from oracle.bruteforce import QuadScene

from tests.hip_util import DEV, DEFAULT_OPTS, GRAD_TOL, OUT_TOL, OUT_CHANNELS, parity_stats, settings

TRACE_CAP = 192
# relative closeness that makes a knife edge.  restart / order / near are statements about t (fp32: 0.5-4 ulp = 6e-8 .. 5e-7); alpha = op exp(-(u^2+v^2)/2)
# carries the cancellation of (u, v) = L (o + t d - mu) through the exponent (|u|, |v| up to 3.3: 1e-4 .. 5e-4 relative at the quad's rim), the
# transmittance the product of up to a hundred (1 - alpha) factors -- the stop test one hit apart is certified from the implementation's own recorded
# alphas instead of a fixed margin (certify) --; "rim": the ray meets the quad within that (u, v) error of its edge
EDGE = {"restart": 2e-6, "order": 1e-6, "alpha": 5e-4, "tstop": 1e-4, "near": 1e-6, "rim": 2e-4}


def trace_sequences(tr, HW, cap=TRACE_CAP):
    """Composited Gaussians per ray of an oracle event trace: (list of int32 arrays, truncated mask)."""
    n = tr["n"].reshape(HW); g = tr["g"].reshape(HW, cap); fl = tr["flags"].reshape(HW, cap)
    seqs = [g[r, :min(int(n[r]), cap)][(fl[r, :min(int(n[r]), cap)] & 1) > 0] for r in range(HW)]
    return seqs, n > cap


def hip_sequences(state, HW):
    """Composited Gaussians per ray of the last recording forward (the hit record, lrt_debug_read 5 / 7)."""
    idx, h = state.handle(DEV)
    cap = state.get_option("hit_cap", DEV)
    hn = np.empty(HW, np.int32); hg = np.empty((HW, cap), np.int32)
    state._lib.lrt_debug_read.restype = C.c_longlong
    for which, arr in ((5, hn), (7, hg)):
        got = state._lib.lrt_debug_read(h, which, arr.ctypes.data_as(C.c_void_p), C.c_longlong(arr.nbytes), None)
        assert got >= arr.nbytes, (which, got, arr.nbytes)
    wa = np.empty((HW, cap, 2), np.float32)                           # (composite weight, un-clamped opacity x G) per recorded hit
    got = state._lib.lrt_debug_read(h, 8, wa.ctypes.data_as(C.c_void_p), C.c_longlong(wa.nbytes), None)
    assert got >= wa.nbytes, (8, got, wa.nbytes)
    return [hg[r, :hn[r]] for r in range(HW)], hn >= cap, wa


def knife_edges(g, t, al):
    """[(depth, kind, margin)] of every place in the reference's loop over the sorted float64 candidates (g, t, alpha) where a decision hangs
    on the last bits: the instrumented twin of oracle/bruteforce.raygen_loop."""
    edges = []
    for j in range(len(t) - 1):
        rel = (t[j + 1] - t[j]) / max(t[j], 1e-30)
        if rel < EDGE["order"]:
            edges.append((float(t[j]), "order", float(rel)))
    for j in range(len(t)):
        if abs(t[j] - 0.2) / 0.2 < EDGE["near"]:
            edges.append((float(t[j]), "near", float(abs(t[j] - 0.2) / 0.2)))
    T, start, i = 1.0, -1.0, 0
    while True:
        while i < len(g) and not (t[i] > start):
            i += 1
        chunk = list(range(i, min(i + 16, len(g)))); i += len(chunk)
        stop = False
        for k in chunk:
            if t[k] < 0.2:
                continue
            if abs(al[k] * 255.0 - 1.0) < EDGE["alpha"]:
                edges.append((float(t[k]), "alpha", float(abs(al[k] * 255.0 - 1.0))))
            if al[k] < 1 / 255:
                continue
            q = T * (1 - al[k])
            if abs(q / 1e-4 - 1.0) < EDGE["tstop"]:
                edges.append((float(t[k]), "tstop", float(abs(q / 1e-4 - 1.0))))
            if q < 1e-4:
                edges.append((float(t[k]), "stop", float(abs(q / 1e-4 - 1.0))))      # where the exact loop stops, and by how much
                stop = True
                break
            if q < 1.05e-4:
                edges.append((float(t[k]), "stop", float(abs(q / 1e-4 - 1.0))))      # ... or nearly does
            T = q
        if stop or len(chunk) < 16:
            break
        start = t[chunk[-1]] + 1e-5
        for j in range(max(chunk[0] - 2, 0), min(chunk[-1] + 20, len(t))):
            m = abs(t[j] - start) / start
            if m < EDGE["restart"]:
                edges.append((float(t[j]), "restart", float(m)))
    return edges


def quad_uv(sc, qs, g, o_r, d_r):
    """(u, v, half extent, t) of the ray on Gaussian g's quad plane in float64, from the full-scene QuadScene."""
    if qs is None or qs.flim[g] < 0:
        return None
    o = np.asarray(o_r, np.float64); d = np.asarray(d_r, np.float64)
    den = float(qs.n[g] @ d)
    if den == 0.0:
        return None
    t = float(((qs.mu[g] - o) * qs.n[g]).sum() / den)
    p = o + t * d - qs.mu[g]
    return float(qs.U[g] @ p), float(qs.V[g] @ p), float(qs.flim[g]), t


def certify(qs, o_r, d_r, seq_x, seq64, subset=None, sc=None, alpha_x=None, rim_cands=None):
    """Is the place where `seq_x` parts from the fp64 sequence a knife edge?  -> (kind or None, margin, depth of the divergence).
    subset: evaluate the brute-force intersection on these Gaussians only (every candidate any implementation looked at on this ray) instead of
    on all P -- the large scenes' hundreds of event rays x millions of quads; the first rays of every scene take the full scene and must agree."""
    certify.last_diag = None
    if subset is not None:
        idx = np.unique(np.asarray(subset, np.int64))
        q2 = QuadScene(sc["means"][idx], sc["scales"][idx], sc["rotations"][idx], sc["opacities"][idx])
        g, t, al = q2.candidates(np.asarray(o_r, np.float64), np.asarray(d_r, np.float64))
        g = idx[g]
    else:
        g, t, al = qs.candidates(np.asarray(o_r, np.float64), np.asarray(d_r, np.float64))
    where = {int(gg): float(tt) for gg, tt in zip(g.tolist(), t.tolist())}
    m = min(len(seq_x), len(seq64))
    i = next((k for k in range(m) if int(seq_x[k]) != int(seq64[k])), m)
    involved = [int(s[i]) for s in (seq_x, seq64) if i < len(s)]
    depths = [where[q] for q in involved if q in where]
    # the depth window in which the two loops part: from the last hit both composited to the first one they disagree about.  The deciding
    # candidate need not be composited by either side (the hit at which ONE loop stops; a candidate one loop never sees), so every candidate and
    # every looked-at quad in the window is examined
    t_prev = where.get(int(seq64[i - 1]), 0.0) if i > 0 else 0.0
    t_div = min(depths) if depths else None
    hi = (t_div if t_div is not None else (float(t[-1]) if len(t) else t_prev)) * (1.0 + 2e-6) + 3e-5
    lo = t_prev - 3e-5
    # the implementation's own transmittance error at the divergence, from the alphas it recorded (a stop one hit apart is a knife edge exactly
    # when the exact stop test lies within that error of the threshold)
    own = None
    if alpha_x is not None and len(alpha_x) >= i and i > 0:
        a64 = {int(gg): float(aa) for gg, aa in zip(g.tolist(), al.tolist())}
        if all(int(q) in a64 for q in seq64[:i]):
            ax = np.minimum(0.99, np.asarray(alpha_x[:i], np.float64)); a6 = np.array([a64[int(q)] for q in seq64[:i]])
            if float(np.max(np.abs(ax / a6 - 1.0))) < 2e-3:               # every alpha arithmetic-close to the exact one
                own = abs(float(np.prod(1.0 - ax)) / float(np.prod(1.0 - a6)) - 1.0)
    best = None
    for (tl, kind, marg) in knife_edges(g, t, al):
        if not (lo <= tl <= hi):
            continue
        if kind == "stop":
            if not (marg <= 2.0 * (own if own is not None else 5e-5) + 1e-5):
                continue
            kind = "tstop"
        if best is None or marg < best[1]:
            best = (kind, marg)
    if best is None and qs is not None:
        # a quad met at its RIM: a candidate for one arithmetic, a miss for the other (it may be absent from the float64 candidate list altogether)
        looked = subset if subset is not None else rim_cands
        cand = set(involved) | (set(int(x) for x in np.asarray(looked).tolist()) if looked is not None else set())
        for q in cand:
            uvf = quad_uv(sc, qs, q, o_r, d_r)
            if uvf is None:
                continue
            u, v, fl, tq = uvf
            m_ = abs(max(abs(u), abs(v)) - fl) / fl
            if m_ < EDGE["rim"] and lo <= tq <= hi and (best is None or m_ < best[1]):
                best = ("rim", float(m_))
    certify.last_diag = {"window": [lo, hi], "own_T_error": own}
    return (best[0], best[1], t_div) if best else (None, None, t_div)


def _mask_rows(img, rows, src):
    out = np.array(img, copy=True).reshape(-1, img.shape[-1])
    out[rows] = np.asarray(src).reshape(-1, img.shape[-1])[rows]
    return out.reshape(img.shape)


def event_masked_gate(name, sc, o, d, deg, bg, dL, f32_fw, f64_fw, opts=None, max_certify=600, extra=None, count_only=False):
    """Runs the HIP forward (+ backward on the MASKED upstream gradient), finds and certifies the event rays, and asserts the (1.1, 1.25) gate on
    everything else.  f32_fw / f64_fw: the oracles' forward results (for `accum`).  Returns the record it wrote to gpurun_out/parity/<name>.json."""
    from lidar_rt_amd.diff_lidar_tracer import Tracer
    H, W = o.shape[:2]; HW = H * W
    bg = np.asarray(bg, np.float32)
    traces = {}
    for prec in ("f32", "f64"):
        orc = oracle.Oracle(sc["means"], sc["scales"], sc["rotations"], sc["opacities"], prec)
        traces[prec] = orc.forward_trace(o, d, sc["shs"], deg, bg, cap=TRACE_CAP)
        del orc
    s32, tr32 = trace_sequences(traces["f32"], HW); s64, tr64 = trace_sequences(traces["f64"], HW)
    # ---- HIP forward, its hit record
    tr = Tracer()
    for k, v in {**DEFAULT_OPTS, **(opts or {})}.items():
        tr.optix_context.set_option(k, v)
    t = {k: torch.as_tensor(np.asarray(v, np.float32), device=DEV).requires_grad_(True) for k, v in sc.items()}
    ro, rd = torch.as_tensor(np.asarray(o, np.float32), device=DEV), torch.as_tensor(np.asarray(d, np.float32), device=DEV)
    tr.build_from_gaussians(t["means"], t["scales"], t["rotations"], t["opacities"])
    out, acc = tr(ro, rd, None, t["means"], torch.zeros_like(t["means"]), shs=t["shs"], opacities=t["opacities"], scales=t["scales"],
                  rotations=t["rotations"], tracer_settings=settings(bg, deg))
    torch.cuda.synchronize()
    shp, trh, hip_wa = hip_sequences(tr.optix_context, HW)
    same = lambda a, b: len(a) == len(b) and np.array_equal(np.asarray(a, np.int64), np.asarray(b, np.int64))
    usable = ~(tr64 | tr32 | trh)                                      # (a truncated trace says nothing: such rays are masked, and counted)
    ev_hip = np.array([usable[r] and not same(shp[r], s64[r]) for r in range(HW)])
    ev_f32 = np.array([usable[r] and not same(s32[r], s64[r]) for r in range(HW)])
    n_hip, n_f32 = int(ev_hip.sum()), int(ev_f32.sum())
    # ---- (2) no more events than the reference's own arithmetic
    allow = n_f32 + 3.0 * np.sqrt(n_f32 + 1.0) + 2.0
    # ---- (3) every HIP event is a certified knife edge
    qs = QuadScene(sc["means"], sc["scales"], sc["rotations"], sc["opacities"])
    o2, d2 = np.asarray(o, np.float64).reshape(-1, 3), np.asarray(d, np.float64).reshape(-1, 3)
    kinds, uncertified = {}, []
    rays = np.nonzero(ev_hip)[0]
    g32 = traces["f32"]["g"].reshape(HW, TRACE_CAP); g64 = traces["f64"]["g"].reshape(HW, TRACE_CAP)
    n32 = traces["f32"]["n"].reshape(HW); n64 = traces["f64"]["n"].reshape(HW)
    for nr, r in enumerate(rays[:max_certify]):
        looked_at = np.concatenate([g32[r, :n32[r]], g64[r, :n64[r]], np.asarray(shp[r], np.int32)])
        kind, marg, t_div = certify(qs, o2[r], d2[r], shp[r], s64[r], subset=looked_at, sc=sc, alpha_x=hip_wa[r, :len(shp[r]), 1])
        if nr < 12:                                                   # the whole scene, every quad: the same verdict
            k_full, m_full, _ = certify(qs, o2[r], d2[r], shp[r], s64[r], alpha_x=hip_wa[r, :len(shp[r]), 1], rim_cands=looked_at)
            assert (k_full is None) == (kind is None), (name, int(r), kind, k_full)
        if kind is None:
            m_ = min(len(shp[r]), len(s64[r])); i_ = next((k for k in range(m_) if int(shp[r][k]) != int(s64[r][k])), m_)
            uncertified.append({"ray": int(r), "depth": t_div, "first_difference": int(i_), "len": [len(shp[r]), len(s64[r])], "diag": getattr(certify, "last_diag", None),
                                "hip": [int(x) for x in shp[r][max(i_ - 2, 0):i_ + 4]], "f64": [int(x) for x in s64[r][max(i_ - 2, 0):i_ + 4]]})
        else:
            kinds[kind] = kinds.get(kind, 0) + 1
    if count_only:                                                    # (a comparison run: events counted and certified, nothing masked or asserted)
        rec_ = {"name": name, "event_rays": {"hip": n_hip, "fp32_oracle": n_f32}, "certified": kinds, "uncertified": len(uncertified), **(extra or {})}
        try:
            dd_ = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity")
            os.makedirs(dd_, exist_ok=True)
            with open(os.path.join(dd_, name + "_events.json"), "w") as f_:
                json.dump(rec_, f_, indent=1)
        except OSError:
            pass
        return rec_
    # ---- (4) the claim itself on everything else
    masked = ev_hip | ev_f32 | ~usable
    rows = np.nonzero(masked)[0]
    dLm = np.array(dL, np.float32, copy=True).reshape(HW, -1); dLm[rows] = 0.0; dLm = dLm.reshape(np.asarray(dL).shape)
    out.backward(torch.as_tensor(dLm, device=DEV))
    torch.cuda.synchronize()
    hip = {"out": _mask_rows(out.detach().cpu().numpy(), rows, traces["f64"]["out"]), "accum": acc.detach().cpu().numpy(),
           "grads": {k: t[k].grad.detach().cpu().numpy() for k in ("means", "scales", "rotations", "opacities", "shs")}}
    bw = {}
    for prec in ("f32", "f64"):
        orc = oracle.Oracle(sc["means"], sc["scales"], sc["rotations"], sc["opacities"], prec)
        bw[prec] = orc.backward(o, d, sc["shs"], deg, bg, traces[prec]["out"], dLm)
        del orc
    o32 = _mask_rows(traces["f32"]["out"], rows, traces["f64"]["out"])
    table, bad = {}, []

    def bulk_l2(x, ref):
        """Median over 32 fixed random blocks of the block's relative L2 error, and the share of the squared error its largest element carries.
        The plain relative L2 of a gradient table is a one-sample statistic: on the KITTI-360 frame 2 ONE component of one grazing Gaussian
        (|error| 7e-4 of its value, inside the tolerance) carries 65 % of the HIP path's squared error and another one 60 % of the fp32
        oracle's (profiles/r06_parity.md) -- the ratio of two such numbers says nothing.  The median of block norms does not see single elements;
        those are bounded by the count beyond the tolerance (above) and by the plain L2 staying within the north-star tolerance (below)."""
        e2 = (np.asarray(x, np.float64).reshape(-1) - np.asarray(ref, np.float64).reshape(-1)) ** 2
        r2 = np.asarray(ref, np.float64).reshape(-1) ** 2
        top = float(e2.max() / max(e2.sum(), 1e-300)) if e2.size else 0.0
        B = 32
        if e2.size < B * 64:
            return float(np.sqrt(e2.sum() / max(r2.sum(), 1e-300))), top
        perm = np.random.default_rng(12345).permutation(e2.size)[:B * (e2.size // B)].reshape(B, -1)
        return float(np.median(np.sqrt(e2[perm].sum(1) / np.maximum(r2[perm].sum(1), 1e-300)))), top

    def add(label, got, ref32, ref64, tol, width=1):
        h, f = parity_stats(got, ref64, tol), parity_stats(ref32, ref64, tol)
        (h["bulk_l2"], h["top1_share"]), (f["bulk_l2"], f["top1_share"]) = bulk_l2(got, ref64), bulk_l2(ref32, ref64)
        n = max(h["n"], 1)
        ev_floor = f["frac_gt_tol"] * n
        # what is left is arithmetic: an outlier element moves at most its own row
        allow_frac = (1.1 * ev_floor + 3.0 * np.sqrt(width * (ev_floor + 1.0)) + 2.0 * width) / n
        qs_ = ("frac_gt_tol", "rel_l2", "bulk_l2", "top1_share", "max_rel")
        table[label] = {"hip_vs_f64": {q: h[q] for q in qs_}, "f32_vs_f64": {q: f[q] for q in qs_}, "tol": tol}
        if h["frac_gt_tol"] > allow_frac:
            bad.append((label, "frac_gt_tol", h["frac_gt_tol"], f["frac_gt_tol"]))
        if h["bulk_l2"] > 1.25 * f["bulk_l2"] and h["bulk_l2"] > 1e-7:     # (1e-7: both at the resolution of float32 itself)
            bad.append((label, "bulk_l2", h["bulk_l2"], f["bulk_l2"]))
        if h["rel_l2"] > max(1.25 * f["rel_l2"], tol) and h["rel_l2"] > 1e-7:      # the plain L2: within the fp32 oracle's own, or within the north-star tolerance
            bad.append((label, "rel_l2", h["rel_l2"], f["rel_l2"]))

    for c, cname in OUT_CHANNELS:
        add(f"out.{cname}", hip["out"][..., c], o32[..., c], traces["f64"]["out"][..., c], OUT_TOL)
    for gname in ("means", "scales", "rotations", "opacities", "shs"):
        ref = bw["f64"][gname]
        add(f"grad.{gname}", hip["grads"][gname].reshape(ref.shape), bw["f32"][gname], ref, GRAD_TOL, width=int(np.prod(ref.shape[1:])))
    rec = {"name": name, "rays": [H, W], "event_rays": {"hip": n_hip, "fp32_oracle": n_f32, "allowed": float(allow), "in_both": int((ev_hip & ev_f32).sum()),
                                                         "truncated_traces": int((~usable).sum())},
           "certified": kinds, "uncertified": uncertified[:20], "masked_rays": int(masked.sum()), "gate": "count beyond tol <= 1.1 x fp32 oracle's (+3 sigma); block-median L2 <= 1.25 x fp32 oracle's; plain L2 <= max(1.25 x, north-star tol); no scene factor",
           "rows": table, "violations": [list(map(str, b)) for b in bad], **(extra or {})}
    try:
        dd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity")
        os.makedirs(dd, exist_ok=True)
        if os.environ.get("LRT_GATE_DUMP") == "1":                     # developer switch: the gradient tables themselves, for a look at who carries a statistic
            gk = [k for k in ("means", "scales", "rotations", "opacities") if any(b[0] == "grad." + k for b in bad)]
            np.savez_compressed(os.path.join(dd, name + "_grads.npz"), **{f"hip_{k}": hip["grads"][k] for k in gk},
                                **{f"f32_{k}": np.asarray(bw["f32"][k], np.float32) for k in gk}, **{f"f64_{k}": bw["f64"][k] for k in gk},
                                means=sc["means"], scales=sc["scales"], opacities=sc["opacities"], masked=rows, accum=hip["accum"])
        with open(os.path.join(dd, name + "_events.json"), "w") as f:
            json.dump(rec, f, indent=1)
    except OSError:
        pass
    assert n_hip <= allow, f"{name}: {n_hip} rays whose composited sequence leaves the fp64 oracle's, the fp32 oracle has {n_f32} (allowed {allow:.0f})"
    assert not uncertified, f"{name}: {len(uncertified)} event ray(s) that are NOT knife edges by the brute-force float64 restatement: {uncertified[:3]}"
    assert not bad, f"{name}: with the {int(masked.sum())} event rays masked the HIP path is farther from the fp64 oracle than (1.1, 1.25) x the fp32 oracle is: {bad}"
    return rec
