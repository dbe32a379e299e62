#!/usr/bin/env python3
"""Merge gpurun_out/parity/*.json (written by the -m gpu parity tests on the GPU box) into profiles/<tag>_parity.json and
print the table of BASELINE.md section 6 (HIP vs fp32 oracle, with the fp32-vs-fp64 oracle floor beside every number)."""
import glob, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
reps = [json.load(open(f)) for f in sorted(glob.glob(os.path.join(REPO, "gpurun_out", "parity", "*.json")))]
out = {"what": "HIP path vs fp32 oracle per tensor: max / p99 relative error, fraction beyond the north-star tolerance (1e-4 outputs, "
               "1e-3 gradients), relative L2; 'floor' = the same statistics for the fp32 oracle against the fp64 oracle",
       "reports": reps}
json.dump(out, open(os.path.join(REPO, "profiles", f"{tag}_parity.json"), "w"), indent=1)
for r in reps:
    print(f"\n## {r['name']}: {r.get('config', '')}")
    print("| tensor | max rel | p99 rel | frac > tol | rel L2 | floor frac | floor L2 | hip-vs-f64 frac | hip-vs-f64 L2 |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|")
    for k, v in r["rows"].items():
        fl = v.get("floor", {})
        print(f"| {k} | {v['max_rel']:.2e} | {v['p99_rel']:.2e} | {v['frac_gt_tol']:.2e} | {v['rel_l2']:.2e} | "
              f"{fl.get('frac_gt_tol', float('nan')):.2e} | {fl.get('rel_l2', float('nan')):.2e} | "
              f"{v.get('vs_f64', {}).get('frac_gt_tol', float('nan')):.2e} | {v.get('vs_f64', {}).get('rel_l2', float('nan')):.2e} |")


# digest: per scene the worst output channel and the worst gradient (by fraction beyond the tolerance), floor beside it
def _worst(rows, prefix):
    best = None
    for k, v in rows.items():
        if k.startswith(prefix) and (best is None or v["frac_gt_tol"] > best[1]["frac_gt_tol"]): best = (k, v)
    return best
def _fmt(b):
    if not b: return "-"
    k, v = b; fl = v.get("floor", {})
    return (f"{k.split('.', 1)[1]}: frac {v['frac_gt_tol']:.1e} (floor {fl.get('frac_gt_tol', float('nan')):.1e}), "
            f"L2 {v['rel_l2']:.1e} (floor {fl.get('rel_l2', float('nan')):.1e}), p99 {v['p99_rel']:.1e}")
md = ["| scene | rays | Gaussians | worst output channel (tol 1e-4) | worst gradient (tol 1e-3) | bound (a): against the fp32 oracle | bound (b): arbitrated by the fp64 oracle; measured worst ratios (fraction / L2) |", "|---|---|---:|---|---|---|---|"]
for r in reps:
    md.append(f"| {r['name']} | {'x'.join(map(str, r['rays']))} | {r['gaussians']} | {_fmt(_worst(r['rows'], 'out'))} | {_fmt(_worst(r['rows'], 'grad'))} | "
              + (f"asserted: frac <= {r['k_frac']:g} x floor, L2 <= {r['k_l2']:g} x floor" if r["asserted"] else "recorded, not asserted against the floor") + " | "
              + ((f"asserted: HIP-vs-fp64 frac <= {r['f64_k_frac']:g} x (fp32-vs-fp64) + counting noise, L2 <= {r['f64_k_l2']:g} x" if r["asserted"] and r.get("f64_k_frac") else "recorded")
                 + (f"; measured {r['f64_gate_worst']['frac_ratio']:.2f} / " + (f"{r['f64_gate_worst']['l2_ratio']:.2f}" if r['f64_gate_worst'].get('l2_ratio') is not None else "-") if r.get("f64_gate_worst") else "")) + " |")
open(os.path.join(REPO, "profiles", f"{tag}_parity.md"), "w").write(
    f"# Parity digest `{tag}` (HIP path against the fp32 oracle; floor = fp32 oracle against fp64 oracle; full table: {tag}_parity.json)\n\n" + "\n".join(md) + "\n")
