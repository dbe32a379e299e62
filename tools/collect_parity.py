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
