#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/collect_profiles.sh (gpurun_out/prof_<tag>/) into the committed summaries
profiles/<tag>_kernel_stats.csv, profiles/<tag>_summary.md and profiles/pmc_traffic.json (read by bench.py)."""
import csv
import json
import os
import re
import sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
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


rows = list(csv.DictReader(open(os.path.join(src, "stats", "s_kernel_stats.csv"))))
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
    f.write("kernel,calls,total_ms,avg_us,percent,min_us,max_us\n")
    for r in rows:
        f.write(f"\"{short(r['Name'])}\",{r['Calls']},{float(r['TotalDurationNs'])/1e6:.3f},{float(r['AverageNs'])/1e3:.2f},"
                f"{r['Percentage']},{float(r['MinNs'])/1e3:.2f},{float(r['MaxNs'])/1e3:.2f}\n")

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

traffic = {}
lines = [f"# rocprofv3 summary `{tag}` (MI355X, `python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-both-paths --min-seconds 0`: the direct path only)\n",
         "Raw rocprofv3 outputs were written under `gpurun_out/prof_%s/` on the GPU box; this file is the committed digest.\n" % tag]
if bench:
    lines.append(f"bench line under `--kernel-trace --stats`: **{bench['value']/1e6:.2f} M rays/s**, {bench['ms_per_step']:.3f} ms/step; "
                 f"bench's own HIP-event averages: {bench['roofline']['avg_kernel_ms']}\n")
lines.append("## Kernel time (`--kernel-trace --stats`)\n")
lines.append("| kernel | calls | avg us | % of GPU time |\n|---|---:|---:|---:|")
for r in rows[:14]:
    lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {r['Percentage']} |")
lines.append("\n## PMC (separate `--pmc` passes; per-launch averages)\n")
lines.append("FETCH_SIZE / WRITE_SIZE are in KB; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports half of the bytes of wide "
             "coalesced reads, so `hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024` is used as the corrected traffic (the uncorrected sum is shown too).\n")
lines.append("| kernel | FETCH_SIZE KB | WRITE_SIZE KB | traffic MB (corrected) | traffic MB (raw) | L2 hit % | SQ active % | SQ wait-any % | SQ issue-stall % | VALU insts/launch |\n|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
for k, c in sorted(pmc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", [0, 1])[0]):
    if not any(s in k for s in ("k_", "radix", "sort", "onesweep")):
        continue
    def avg(n):
        return c[n][0] / c[n][1] if n in c and c[n][1] else None
    fs, ws = avg("FETCH_SIZE"), avg("WRITE_SIZE")
    hit, miss = avg("TCC_HIT_sum"), avg("TCC_MISS_sum")
    wc = avg("SQ_WAVE_CYCLES")
    corr = (2 * (fs or 0) + (ws or 0)) * 1024 / 1e6
    raw = ((fs or 0) + (ws or 0)) * 1024 / 1e6
    f = lambda v, d=1: "-" if v is None else f"{v:.{d}f}"
    pct = lambda n: "-" if not wc or avg(n) is None else f"{100*avg(n)/wc:.0f}"
    l2 = "-" if hit is None or miss is None or hit + miss == 0 else f"{100*hit/(hit+miss):.0f}"
    vi = avg("SQ_INSTS_VALU")
    lines.append(f"| `{k}` | {f(fs)} | {f(ws)} | {corr:.1f} | {raw:.1f} | {l2} | {pct('SQ_ACTIVE_INST_ANY')} | {pct('SQ_WAIT_ANY')} | {pct('SQ_WAIT_INST_ANY')} | {f(vi,0)} |")
    if fs is not None or ws is not None:
        traffic[k] = {"hbm_bytes_per_launch": corr * 1e6, "fetch_kb": fs, "write_kb": ws, "raw_bytes_per_launch": raw * 1e6}
        # share of the chip's VALU issue slots this kernel used: a wave64 VALU instruction occupies a 16-lane SIMD for 4 cycles;
        # 256 CUs x 4 SIMDs at 2.4 GHz (MI355X_MICROARCH.md)
        dur = next((float(r["AverageNs"]) * 1e-9 for r in rows if short(r["Name"]) == k), None)
        if vi and dur:
            traffic[k]["valu_insts_per_launch"] = vi
            traffic[k]["valu_issue_frac"] = vi * 4.0 / (dur * 1024 * 2.4e9)

# labels bench.py uses for the dominant (HIP-event timed) region: sums over the kernels of the region
fw = [traffic[k] for k in traffic if k.startswith("k_fwd_")]
if fw:
    traffic["forward (k_fwd_cr4 + k_fwd_colour)"] = {"hbm_bytes_per_launch": sum(x["hbm_bytes_per_launch"] for x in fw),
                                                      "raw_bytes_per_launch": sum(x["raw_bytes_per_launch"] for x in fw),
                                                      "valu_issue_frac_dominant_kernel": max((x.get("valu_issue_frac", 0.0) for x in fw), default=None)}
bw = [traffic[k] for k in traffic if k.startswith("k_bwd_") or k.startswith("k_bk_")]
if bw:
    traffic["backward (k_bk_count .. k_bwd_prep .. k_bk_sort + k_bwd_reduce4)"] = {"hbm_bytes_per_launch": sum(x["hbm_bytes_per_launch"] for x in bw),
                                                                                   "raw_bytes_per_launch": sum(x["raw_bytes_per_launch"] for x in bw)}
# whole step from the counters: all dispatches of the PMC runs, per step (= per k_fwd_cr4 dispatch)
tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}; steps_pmc = {"FETCH_SIZE": 0, "WRITE_SIZE": 0}
for k, c in pmc.items():
    for n in tot:
        if n in c:
            tot[n] += c[n][0]
            if k.startswith("k_fwd_cr4"): steps_pmc[n] += c[n][1]
if steps_pmc["FETCH_SIZE"] and steps_pmc["WRITE_SIZE"]:
    traffic["whole step (all kernels, per step)"] = {
        "hbm_bytes_per_launch": (2 * tot["FETCH_SIZE"] / steps_pmc["FETCH_SIZE"] + tot["WRITE_SIZE"] / steps_pmc["WRITE_SIZE"]) * 1024,
        "raw_bytes_per_launch": (tot["FETCH_SIZE"] / steps_pmc["FETCH_SIZE"] + tot["WRITE_SIZE"] / steps_pmc["WRITE_SIZE"]) * 1024}
sys.path.insert(0, REPO)
from lidar_rt_amd.build import source_hash
traffic["_meta"] = {"tag": tag, "csrc_sha": source_hash(), "note": "valid only for the kernel sources with this hash (lidar_rt_amd.build.source_hash); "
                    "bench.py reports traffic = null when the sources have changed since"}
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
