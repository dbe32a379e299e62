#!/usr/bin/env python3
"""Which EVENTS separate the HIP image from the fp64 oracle, and the fp32 oracle from the fp64 oracle?  

The parity tests bound the fraction of pixels beyond the north-star tolerance by k x the algorithm's own noise floor (fp32 oracle vs
fp64 oracle).  This tool explains the numbers: for every ray it compares the sequence of COMPOSITED Gaussians of an fp32
implementation (the HIP path: its hit record; the fp32 oracle: its event trace) with the fp64 oracle's and names the first event
where they part:
    order swap        two neighbouring hits change places (depths closer than fp32 resolves; only the colour channels see it)
    1/255 edge        a candidate whose alpha is within 2e-3 (relative) of the 1/255 threshold is composited by one, skipped by the other
    T-stop edge       one sequence is a proper prefix of the other: the 1e-4 transmittance test fired one hit apart
    restart epsilon   a candidate inside the +1e-5 restart window of a 16-chunk (forward.cu:282-291) is seen by one, dropped by the other
    other             none of the above at the first difference
Rays with IDENTICAL sequences differ by arithmetic only; those beyond tolerance are listed as "same sequence" (0.99 clamp edge when a
composited alpha lies within 1e-5 of the clamp).  Output: a table per implementation (rays per class, rays of the class whose
intensity / ray-drop / depth leave 1e-4) -> profiles/<tag>_parity_events.{json,md}.

    python tests/tools/parity_events.py [s1m|s200k|s10k] [tag]
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, REPO)
from lidar_rt_amd import scenes                      # noqa: E402
from lidar_rt_amd.parallel import HipBackend        # noqa: E402
from oracle import oracle                           # noqa: E402  (a checker under tests/: the only place besides smoke() and bench.py that may use the oracle)

wl = sys.argv[1] if len(sys.argv) > 1 else "s1m"
tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
if wl == "s1m": sc, o, d = scenes.s1m()
elif wl == "s200k": sc = scenes.make_scene(200_000, radius_scale=0.5); o, d = scenes.kitti_rays(32, 512)
else: sc, o, d = scenes.s10k()
H, W = o.shape[:2]; HW = H * W
bg = scenes.BG_DEFAULT
CAP = 192

tr = {}
for prec in ("f32", "f64"):
    orc = oracle.Oracle(sc["means"], sc["scales"], sc["rotations"], sc["opacities"], prec)
    tr[prec] = orc.forward_trace(o, d, sc["shs"], 3, bg, cap=CAP)
    del orc

dev = torch.device("cuda:0")
t = {k: torch.as_tensor(v, device=dev) for k, v in sc.items()}
be = HipBackend()
be.build(t["means"], t["scales"], t["rotations"], t["opacities"])
out, _ = be.forward(torch.as_tensor(o, device=dev), torch.as_tensor(d, device=dev), t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3,
                    torch.as_tensor(bg, device=dev))
torch.cuda.synchronize()
be.state.check(dev, wait=True)
hip_out = out.cpu().numpy().reshape(HW, 9).astype(np.float64)
idx, hd = be.state.handle(dev)
cap_h = be.state.get_option("hit_cap", dev)
hn = np.empty(HW, np.int32); hg = np.empty((HW, cap_h), np.int32)
be.state._lib.lrt_debug_read.restype = C.c_longlong
for which, arr in ((5, hn), (7, hg)):
    be.state._lib.lrt_debug_read(hd, which, arr.ctypes.data_as(C.c_void_p), C.c_longlong(arr.nbytes), None)

ref = tr["f64"]
ref_out = ref["out"].reshape(HW, 9).astype(np.float64)
n64 = ref["n"].reshape(HW); g64 = ref["g"].reshape(HW, CAP); a64 = ref["alpha"].reshape(HW, CAP); f64_ = ref["flags"].reshape(HW, CAP); t64 = ref["t"].reshape(HW, CAP)


def comp_seq_trace(T, r):
    n = min(int(T["n"].reshape(HW)[r]), CAP)
    fl = T["flags"].reshape(HW, CAP)[r, :n]
    return T["g"].reshape(HW, CAP)[r, :n][(fl & 1) > 0].tolist()


def classify(seq, r):
    """First event where `seq` (composited Gaussians of an fp32 implementation) parts from the fp64 oracle's sequence of ray r."""
    n = min(int(n64[r]), CAP)
    comp64 = g64[r, :n][(f64_[r, :n] & 1) > 0].tolist()
    if seq == comp64:
        near_clamp = bool((np.abs(a64[r, :n][(f64_[r, :n] & 1) > 0] - 0.99) < 1e-5).any())
        return "same sequence (0.99 clamp edge)" if near_clamp else "same sequence"
    if int(n64[r]) > CAP:
        return "other"                                                  # trace truncated
    m = min(len(seq), len(comp64))
    i = next((k for k in range(m) if seq[k] != comp64[k]), m)
    if i == m:
        return "T-stop edge"                                            # one is a proper prefix of the other
    a, b = seq[i], comp64[i]
    if i + 1 < len(seq) and i + 1 < len(comp64) and seq[i + 1] == b and comp64[i + 1] == a:
        return "order swap"
    cand = {int(g): k for k, g in enumerate(g64[r, :n].tolist())}      # position of every fp64 candidate
    def near_thr(g):
        k = cand.get(g)
        return k is not None and abs(float(a64[r, k]) * 255.0 - 1.0) < 2e-3
    extra_in_x = a not in comp64[i:i + 6]                               # the fp32 side composited a where fp64 did not (nearby)
    missing_in_x = b not in seq[i:i + 6]
    g_ = a if extra_in_x else (b if missing_in_x else None)
    if g_ is None:
        return "order swap"                                             # a wider permutation of the same hits
    if near_thr(g_):
        return "1/255 edge"
    k = cand.get(g_)
    if k is None:
        return "restart epsilon"                                        # fp64 never looked at it: dropped behind a chunk's restart
    # fp64 looked at it (and composited it) but the fp32 side did not: dropped on that side if it sits right behind a chunk end
    restarts = np.nonzero((f64_[r, :n] & 8) > 0)[0]
    for q in restarts:
        if q > 0 and abs(float(t64[r, k]) - float(t64[r, q - 1])) < 5e-5:
            return "restart epsilon"
    # the mirror case: the fp32 side restarted somewhere else (its chunk boundaries moved by an earlier event) -> other
    return "other"


RAY_CLASS = {}


def table(name, seq_of, out_x):
    scale = lambda c: np.maximum(np.abs(ref_out[:, c]), 1e-3 * np.abs(ref_out[:, c]).max())
    bad = {c: np.abs(out_x[:, c] - ref_out[:, c]) / scale(c) > 1e-4 for c in (0, 2, 3)}
    any_bad = bad[0] | bad[2] | bad[3]
    rows = {}
    per_ray = np.empty(HW, object)
    # candidates for a difference: every ray beyond tolerance + a sample of the others is not enough -- sequences can differ without a
    # visible error, so all rays are compared (vectorised pre-filter on length and content)
    for r in range(HW):
        seq = seq_of(r)
        n = min(int(n64[r]), CAP)
        comp64 = g64[r, :n][(f64_[r, :n] & 1) > 0]
        if len(seq) == len(comp64) and np.array_equal(np.asarray(seq, np.int32), comp64) and not any_bad[r]:
            kind = "same sequence, within 1e-4"
        else:
            kind = classify(list(seq), r)
        per_ray[r] = kind
        e = rows.setdefault(kind, {"rays": 0, "intensity": 0, "raydrop": 0, "depth": 0})
        e["rays"] += 1; e["intensity"] += int(bad[0][r]); e["raydrop"] += int(bad[2][r]); e["depth"] += int(bad[3][r])
    tot = {k: sum(v[k] for v in rows.values()) for k in ("rays", "intensity", "raydrop", "depth")}
    RAY_CLASS[name] = per_ray
    return {"implementation": name, "classes": rows, "total": tot, "frac_beyond_1e-4": {k: tot[k] / HW for k in ("intensity", "raydrop", "depth")}}


res = {"workload": wl, "rays": HW, "note": "first event where the composited-Gaussian sequence of an fp32 implementation parts from the fp64 oracle's; "
       "counts of rays, and of those rays whose intensity / ray-drop / depth leave 1e-4 relative (floor 1e-3 of the channel maximum)",
       "hip": table("HIP (k_fwd_cr4)", lambda r: hg[r, :min(int(hn[r]), cap_h)].tolist(), hip_out),
       "f32_oracle": table("fp32 oracle", lambda r: comp_seq_trace(tr["f32"], r), tr["f32"]["out"].reshape(HW, 9).astype(np.float64))}
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(REPO, "gpurun_out", f"{tag}_parity_events_{wl}.json"), "w"), indent=1)
order = ["same sequence, within 1e-4", "same sequence", "same sequence (0.99 clamp edge)", "order swap", "1/255 edge", "T-stop edge", "restart epsilon", "other"]
lines = [f"# Parity events on {wl} ({HW} rays): first event where an fp32 implementation's composited sequence parts from the fp64 oracle's\n",
         "| class | HIP rays | HIP beyond 1e-4 (intensity / ray-drop / depth) | fp32-oracle rays | fp32 oracle beyond 1e-4 (intensity / ray-drop / depth) |", "|---|---:|---:|---:|---:|"]
for k in order:
    a = res["hip"]["classes"].get(k, {"rays": 0, "intensity": 0, "raydrop": 0, "depth": 0}); b = res["f32_oracle"]["classes"].get(k, {"rays": 0, "intensity": 0, "raydrop": 0, "depth": 0})
    lines.append(f"| {k} | {a['rays']} | {a['intensity']} / {a['raydrop']} / {a['depth']} | {b['rays']} | {b['intensity']} / {b['raydrop']} / {b['depth']} |")
a, b = res["hip"]["total"], res["f32_oracle"]["total"]
lines.append(f"| **total** | {a['rays']} | {a['intensity']} / {a['raydrop']} / {a['depth']} | {b['rays']} | {b['intensity']} / {b['raydrop']} / {b['depth']} |")
open(os.path.join(REPO, "gpurun_out", f"{tag}_parity_events_{wl}.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

# ---------------------------------------------------------------------------------------------------------------------------------
# Gradients: which ray EVENT moved a Gaussian's gradient beyond 1e-3?  For every Gaussian whose d_opacity (and
# d_means) of an fp32 implementation leaves the fp64 oracle's by more than 1e-3 relative (floor 1e-3 of the tensor's maximum), the rays
# that composite it (fp64 trace) are looked up in the per-ray event classes above; the Gaussian is filed under the most consequential
# event among its rays (restart epsilon > T-stop edge > 1/255 edge > order swap > other > arithmetic only).
if os.environ.get("PARITY_GRADS", "1") == "1":
    dL = scenes.upstream_grad(H, W)
    grads = {}
    for prec in ("f32", "f64"):
        orc = oracle.Oracle(sc["means"], sc["scales"], sc["rotations"], sc["opacities"], prec)
        fw_ = orc.forward(o, d, sc["shs"], 3, bg)
        grads[prec] = orc.backward(o, d, sc["shs"], 3, bg, fw_["out"], dL)
        del orc
    gh = be.backward(torch.as_tensor(o, device=dev), torch.as_tensor(d, device=dev), t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3,
                     torch.as_tensor(bg, device=dev), out, torch.as_tensor(dL, device=dev))
    torch.cuda.synchronize()
    grads["hip"] = {k: v.cpu().numpy() for k, v in gh.items()}
    # Gaussian -> rays that composite it (fp64 trace)
    comp_mask = (f64_ & 1) > 0
    comp_mask &= np.arange(CAP)[None, :] < np.minimum(n64, CAP)[:, None]
    rr_, kk_ = np.nonzero(comp_mask)
    gg_ = g64[rr_, kk_]
    order_ = np.argsort(gg_, kind="stable"); gg_s, rr_s = gg_[order_], rr_[order_]
    P = sc["means"].shape[0]
    start_ = np.searchsorted(gg_s, np.arange(P + 1))
    SEV = ["restart epsilon", "T-stop edge", "1/255 edge", "order swap", "other", "same sequence (0.99 clamp edge)", "same sequence"]

    def grad_table(impl, ray_cls):
        sev_code = np.full(HW, len(SEV), np.int32)
        for i_, nm in enumerate(SEV):
            sev_code[np.asarray([c == nm for c in ray_cls])] = i_
        out_rows = {}
        for field in ("opacities", "means", "shs"):
            a_ = np.asarray(grads[impl][field], np.float64).reshape(P, -1); b_ = np.asarray(grads["f64"][field], np.float64).reshape(P, -1)
            scale_ = np.maximum(np.abs(b_), 1e-3 * np.abs(b_).max())
            badg = np.nonzero((np.abs(a_ - b_) / scale_ > 1e-3).any(1))[0]
            cnt = {nm: 0 for nm in SEV + ["arithmetic only (all its rays: same sequence, within 1e-4)"]}
            for g_ in badg:
                rays_ = rr_s[start_[g_]:start_[g_ + 1]]
                code = int(sev_code[rays_].min()) if len(rays_) else len(SEV)
                cnt[SEV[code] if code < len(SEV) else "arithmetic only (all its rays: same sequence, within 1e-4)"] += 1
            out_rows[field] = {"gaussians_beyond_1e-3": int(len(badg)), "touched_gaussians": int((start_[1:] > start_[:-1]).sum()), "by_event": cnt}
        return out_rows

    gres = {"workload": wl, "gaussians": int(P), "hip": grad_table("hip", RAY_CLASS["HIP (k_fwd_cr4)"]), "f32_oracle": grad_table("f32", RAY_CLASS["fp32 oracle"])}
    json.dump(gres, open(os.path.join(REPO, "gpurun_out", f"{tag}_parity_events_grads_{wl}.json"), "w"), indent=1)
    gl = [f"# Gradient parity events on {wl}: Gaussians whose gradient leaves the fp64 oracle's by more than 1e-3 (relative, floor 1e-3 of the tensor maximum), "
          "filed under the most consequential EVENT among the rays that composite them (events = first difference of a ray's composited sequence from the fp64 oracle's)\n"]
    for field in ("opacities", "means", "shs"):
        hh, ff = gres["hip"][field], gres["f32_oracle"][field]
        gl.append(f"## d_{field}: HIP {hh['gaussians_beyond_1e-3']} of {hh['touched_gaussians']} touched Gaussians beyond 1e-3; fp32 oracle {ff['gaussians_beyond_1e-3']}\n")
        gl.append("| event among the Gaussian's rays | HIP Gaussians | fp32-oracle Gaussians |"); gl.append("|---|---:|---:|")
        for nm in SEV + ["arithmetic only (all its rays: same sequence, within 1e-4)"]:
            gl.append(f"| {nm} | {hh['by_event'][nm]} | {ff['by_event'][nm]} |")
        gl.append("")
    open(os.path.join(REPO, "gpurun_out", f"{tag}_parity_events_grads_{wl}.md"), "w").write("\n".join(gl) + "\n")
    print("\n".join(gl))
