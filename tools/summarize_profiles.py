#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/collect_profiles.sh (gpurun_out/prof_<tag>/) into the committed summaries
profiles/<tag>_kernel_stats.csv, profiles/<tag>_summary.md and profiles/pmc_traffic.json (read by bench.py).

Accounting rules (the r03 forward-region sum included the STATS instantiation of k_fwd_cr4, which runs once per
bench, not once per step):
  * everything is PER STEP: a kernel's bytes per step = its per-launch average x its launches per step (k_rs_pass: 3);
    launches per step = calls / steps, steps = the number of k_fwd_colour dispatches of the same run;
  * instantiations that do not belong to a step are EXCLUDED from every sum: k_fwd_cr4<.., true> (statistics build; the profiling
    runs pass --no-stats-step, so there is normally none) and one-off kernels (k_bounds: first build of a state only);
  * a region's total is the sum of its per-kernel rows by construction; the whole-step total is checked against the sum over ALL
    dispatches of the PMC run / steps (must agree within 2 %, else this script fails).
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(REPO, "gpurun_out", f"prof_{tag}")
dst = os.path.join(REPO, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", name)
    if len(name) > 90:
        m = re.search(r"(radix_sort_[a-z_]+|merge_sort_[a-z_]+|onesweep[a-z_]*|block_sort[a-z_]*|histogram[a-z_]*)", name)
        name = "rocprim::" + (m.group(1) if m else name[-60:])
    return name.replace("void ", "")


def is_stats_inst(k):                      # k_fwd_cr4<DEFER, NW, STATS = true>
    return re.match(r"k_fwd_cr4<[^>]*,\s*true>$", k) is not None


ONE_OFF = ("k_bounds", "k_cone_init")     # not part of a steady-state step


def region_of(k):
    if is_stats_inst(k) or any(k.startswith(o) for o in ONE_OFF):
        return None
    if k.startswith("k_fwd_"):
        return "forward"
    if k.startswith("k_bwd_") or k.startswith("k_bk_") or k.startswith("k_trace<true>"):
        return "backward"
    if k.startswith(("k_morton", "k_rs_", "k_make_", "k_level", "k_upper", "k_tree", "k_pack", "k_cone", "k_drift", "k_index", "rocprim::")):
        return "build"
    return "other"


rows = list(csv.DictReader(open(os.path.join(src, "stats", "s_kernel_stats.csv"))))
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
    f.write("kernel,calls,total_ms,avg_us,percent,min_us,max_us\n")
    for r in rows:
        f.write(f"\"{short(r['Name'])}\",{r['Calls']},{float(r['TotalDurationNs'])/1e6:.3f},{float(r['AverageNs'])/1e3:.2f},"
                f"{r['Percentage']},{float(r['MinNs'])/1e3:.2f},{float(r['MaxNs'])/1e3:.2f}\n")
stat = {short(r["Name"]): {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3} for r in rows}
COLOUR = next((k for k in stat if k.startswith("k_fwd_colour")), "k_fwd_colour")      # the colour pass runs once per step (a template since round 4: `k_fwd_colour<true, 4, 4>`)
steps_stat = stat.get(COLOUR, {}).get("calls", 0)

pmc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_l2"):
    p = os.path.join(src, sub, "p_counter_collection.csv")
    if not os.path.exists(p):
        continue
    for r in csv.DictReader(open(p)):
        k = short(r["Kernel_Name"])
        c = pmc[k][r["Counter_Name"]]
        c[0] += float(r["Counter_Value"]); c[1] += 1

bench = {}
try:
    bench = json.loads(open(os.path.join(src, "bench_under_trace.json")).read().strip().splitlines()[-1])
except Exception:
    pass


def avg(k, n):
    c = pmc[k]
    return c[n][0] / c[n][1] if n in c and c[n][1] else None


COLOUR_P = next((k for k in pmc if k.startswith("k_fwd_colour")), "k_fwd_colour")
steps_pmc = {n: pmc[COLOUR_P][n][1] for n in ("FETCH_SIZE", "WRITE_SIZE") if n in pmc[COLOUR_P]}
kernels = {}
for k in pmc:
    fs, ws = avg(k, "FETCH_SIZE"), avg(k, "WRITE_SIZE")
    if fs is None and ws is None:
        continue
    # launches per step, from the PMC run itself (falls back to the stats run)
    lps = None
    if steps_pmc.get("FETCH_SIZE") and "FETCH_SIZE" in pmc[k]:
        lps = pmc[k]["FETCH_SIZE"][1] / steps_pmc["FETCH_SIZE"]
    elif steps_stat and k in stat:
        lps = stat[k]["calls"] / steps_stat
    reg = region_of(k)
    if reg == "forward" and k.startswith("k_fwd_cr4"):
        lps = 1.0                                                    # the instrumented step (if any) runs the other instantiation
    ent = {"region": reg, "fetch_kb": fs, "write_kb": ws, "launches_per_step": lps,
           "corrected_bytes_per_launch": (2 * (fs or 0) + (ws or 0)) * 1024, "raw_bytes_per_launch": ((fs or 0) + (ws or 0)) * 1024,
           "avg_us": stat.get(k, {}).get("avg_us")}
    vi = avg(k, "SQ_INSTS_VALU")
    if vi and ent["avg_us"]:
        # share of the chip's VALU issue slots: a wave64 VALU instruction occupies a 16-lane SIMD for 4 cycles; 256 CUs x 4 SIMDs at 2.4 GHz
        ent["valu_insts_per_launch"] = vi
        ent["valu_issue_frac"] = vi * 4.0 / (ent["avg_us"] * 1e-6 * 1024 * 2.4e9)
    kernels[k] = ent

regions = {}
for reg in ("build", "forward", "backward", "other"):
    ks = [k for k, e in kernels.items() if e["region"] == reg and e["launches_per_step"]]
    regions[reg] = {"kernels": sorted(ks),
                    "corrected_bytes_per_step": sum(kernels[k]["corrected_bytes_per_launch"] * kernels[k]["launches_per_step"] for k in ks),
                    "raw_bytes_per_step": sum(kernels[k]["raw_bytes_per_launch"] * kernels[k]["launches_per_step"] for k in ks),
                    "us_per_step": sum((kernels[k]["avg_us"] or 0.0) * kernels[k]["launches_per_step"] for k in ks),
                    "launches_per_step": sum(kernels[k]["launches_per_step"] for k in ks)}
step = {q: sum(regions[r][q] for r in regions) for q in ("corrected_bytes_per_step", "raw_bytes_per_step", "us_per_step", "launches_per_step")}
# cross-check: every dispatch of the PMC runs that belongs to a step / steps
tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
for k, c in pmc.items():
    if region_of(k) is None:
        continue
    for n in tot:
        if n in c:
            tot[n] += c[n][0]
check = None
if steps_pmc.get("FETCH_SIZE") and steps_pmc.get("WRITE_SIZE"):
    check = (tot["FETCH_SIZE"] / steps_pmc["FETCH_SIZE"] + tot["WRITE_SIZE"] / steps_pmc["WRITE_SIZE"]) * 1024
    rel = abs(check - step["raw_bytes_per_step"]) / max(check, 1.0)
    # the one k_fwd_cr4<.., true> dispatch (if any) is excluded on both sides; a per-step kernel counted with a wrong multiplicity shows here
    assert rel < 0.02, f"per-kernel rows do not add up to the whole step: rows {step['raw_bytes_per_step']:.4g} B vs all dispatches {check:.4g} B ({100*rel:.1f} %)"
step["raw_bytes_per_step_all_dispatches"] = check

lines = [f"# rocprofv3 summary `{tag}` (MI355X, `python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-both-paths --no-stats-step --min-seconds 0`: the direct path only)\n",
         "Raw rocprofv3 outputs were written under `gpurun_out/prof_%s/` on the GPU box; this file is the committed digest "
         "(`tools/summarize_profiles.py`; bench.py's `roofline.per_kernel` reads `profiles/pmc_traffic.json`, written by the same run of the script).\n" % tag]
if bench:
    lines.append(f"bench line under `--kernel-trace --stats`: **{bench['value']/1e6:.2f} M rays/s**, {bench['ms_per_step']:.3f} ms/step; "
                 f"bench's own HIP-event averages: {bench['roofline']['avg_kernel_ms']}\n")
lines.append("## Kernel time (`--kernel-trace --stats`)\n")
lines.append("| kernel | calls | avg us | % of GPU time |\n|---|---:|---:|---:|")
for r in rows[:16]:
    lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {r['Percentage']} |")
lines.append("\n## PMC (separate `--pmc` passes; per-launch averages)\n")
lines.append("FETCH_SIZE / WRITE_SIZE are in KB.  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports half of the bytes of wide "
             "coalesced reads, so `corrected = (2 x FETCH_SIZE + WRITE_SIZE) x 1024`; `raw = (FETCH_SIZE + WRITE_SIZE) x 1024`.  Which one applies to "
             "which access pattern: `profiles/r04_fetch_calibration.md`.\n")
lines.append("| kernel | region | launches / step | FETCH_SIZE KB | WRITE_SIZE KB | corrected MB | raw MB | L2 hit % | SQ active % | SQ wait-any % | SQ issue-stall % | VALU insts/launch |\n|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
for k, e in sorted(kernels.items(), key=lambda kv: -(pmc[kv[0]].get("SQ_WAVE_CYCLES", [0, 1])[0])):
    hit, miss, wc = avg(k, "TCC_HIT_sum"), avg(k, "TCC_MISS_sum"), avg(k, "SQ_WAVE_CYCLES")
    f = lambda v, d=1: "-" if v is None else f"{v:.{d}f}"
    pct = lambda n: "-" if not wc or avg(k, n) is None else f"{100*avg(k, n)/wc:.0f}"
    l2 = "-" if hit is None or miss is None or hit + miss == 0 else f"{100*hit/(hit+miss):.0f}"
    lines.append(f"| `{k}` | {e['region'] or 'excluded'} | {f(e['launches_per_step'], 2)} | {f(e['fetch_kb'])} | {f(e['write_kb'])} | {e['corrected_bytes_per_launch']/1e6:.1f} | "
                 f"{e['raw_bytes_per_launch']/1e6:.1f} | {l2} | {pct('SQ_ACTIVE_INST_ANY')} | {pct('SQ_WAIT_ANY')} | {pct('SQ_WAIT_INST_ANY')} | {f(e.get('valu_insts_per_launch'), 0)} |")
lines.append("\n## Per step, by region (sum of the rows above x launches per step; kernel time from the stats run)\n")
lines.append("| region | kernels | launches | us | corrected MB | raw MB | corrected TB/s | raw TB/s |\n|---|---|---:|---:|---:|---:|---:|---:|")
for reg, e in list(regions.items()) + [("whole step", step)]:
    us = e["us_per_step"]
    bw = lambda b: "-" if not us else f"{b / (us * 1e-6) / 1e12:.2f}"
    lines.append(f"| {reg} | {', '.join('`%s`' % k for k in e.get('kernels', [])) or 'all of the above'} | {e['launches_per_step']:.0f} | {us:.1f} | "
                 f"{e['corrected_bytes_per_step']/1e6:.1f} | {e['raw_bytes_per_step']/1e6:.1f} | {bw(e['corrected_bytes_per_step'])} | {bw(e['raw_bytes_per_step'])} |")
if check is not None:
    lines.append(f"\nCross-check: all step dispatches of the PMC runs / steps = {check/1e6:.1f} MB raw per step (rows: {step['raw_bytes_per_step']/1e6:.1f} MB).")

sys.path.insert(0, REPO)
from lidar_rt_amd.build import source_hash
out = {"_meta": {"tag": tag, "csrc_sha": source_hash(), "note": "valid only for the kernel sources with this hash (lidar_rt_amd.build.source_hash); "
                 "bench.py reports counter traffic = null when the sources have changed since.  All *_per_step figures: per-launch average x launches per step; "
                 "k_fwd_cr4<.., true> (statistics instantiation) and one-off kernels are excluded."},
       "kernels": kernels, "regions": regions, "step": step}
json.dump(out, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
