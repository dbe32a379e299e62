#!/usr/bin/env python3
"""profiles/<tag>_parity.md from the event-gate records the GPU suite leaves in gpurun_out/parity/*_events.json (tests/event_gate.py).
usage: python tools/parity_digest.py r06"""
import glob, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
out = [f"# {tag}: parity gates with the event rays certified and masked (`tests/event_gate.py`; written by the GPU suite into `gpurun_out/parity/*_events.json`, digest by `tools/parity_digest.py`)\n",
       "Per scene: rays whose composited sequence leaves the fp64 oracle's (HIP path / fp32 oracle / allowed = fp32's + 3 sigma + 2), how the HIP event rays were certified by the brute-force float64 "
       "restatement (at most 600 per scene are looked at), and -- with the event rays of both implementations masked -- the worst ratio HIP-vs-fp64 : fp32-oracle-vs-fp64 over the six rendered channels "
       "and five gradient tensors.  Gate: count beyond the north-star tolerance <= 1.1 x (+ 3 sigma), block-median L2 <= 1.25 x, plain L2 <= max(1.25 x, tolerance); no factor depends on the scene.\n",
       "| scene | rays | event rays HIP / fp32 oracle / allowed | certified as | uncertified | masked rays | worst count ratio | worst block-median L2 ratio | worst plain L2 ratio (row) | violations |",
       "|---|---|---|---|---:|---:|---:|---:|---|---|"]
for f in sorted(glob.glob(os.path.join(REPO, "gpurun_out", "parity", "*_events.json"))):
    d = json.load(open(f))
    if "rows" not in d:
        out.append(f"| {d['name']} (events counted only) | | {d['event_rays']['hip']} / {d['event_rays']['fp32_oracle']} | {d.get('certified')} | {d.get('uncertified')} | | | | | |")
        continue
    wc = wb = wl = 0.0; wlr = ""
    for k, r in d["rows"].items():
        h, f_ = r["hip_vs_f64"], r["f32_vs_f64"]
        if f_["frac_gt_tol"] > 0: wc = max(wc, h["frac_gt_tol"] / f_["frac_gt_tol"])
        if "bulk_l2" in h and f_["bulk_l2"] > 1e-7: wb = max(wb, h["bulk_l2"] / f_["bulk_l2"])
        if f_["rel_l2"] > 1e-7 and h["rel_l2"] / f_["rel_l2"] > wl:
            wl = h["rel_l2"] / f_["rel_l2"]; wlr = f"{k}: {h['rel_l2']:.2e} vs {f_['rel_l2']:.2e}, top-1 share {h.get('top1_share', 0):.2f} / {f_.get('top1_share', 0):.2f}"
    e = d["event_rays"]
    out.append(f"| {d['name']} | {d['rays'][0]}x{d['rays'][1]} | {e['hip']} / {e['fp32_oracle']} / {e['allowed']:.0f} | {d['certified']} | {len(d['uncertified'])} | {d['masked_rays']} | {wc:.2f} | {wb:.2f} | {wl:.2f} ({wlr}) | {len(d['violations'])} |")
open(os.path.join(REPO, "profiles", f"{tag}_parity.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out[3:]))
