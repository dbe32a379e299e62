#!/bin/bash
# Run on the GPU box (via gpurun): FETCH_SIZE / WRITE_SIZE of the calibration kernels (tools/microbench/fetch_calib.hip) against their known byte counts.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/fetch_calib; rm -rf $OUT; mkdir -p $OUT
BIN=$R/tools/microbench/fetch_calib
[ -x $BIN ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $BIN $R/tools/microbench/fetch_calib.hip
$BIN > $OUT/timing.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_ATOMIC_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  d=$OUT/$(echo $c | tr ' ' '_')
  timeout -k 5 120 rocprofv3 --pmc $c --output-format csv -d $d -o p -- $BIN > /dev/null 2> $d.err || echo "pass '$c' failed: $(tail -2 $d.err)"
done
OUT=$OUT python3 - <<'PY'
import csv, glob, collections, os
OUT = os.environ["OUT"]
known = {"cal_stream16": (1073.74, 0), "cal_gather<1, 8>": (67.11, 0), "cal_gather<4, 4>": (268.44, 0), "cal_gather<12, 12>": (805.31, 0), "cal_write16": (0, 1073.74), "cal_scatter16": (0, 67.11), "cal_atomic": (0, 16.78)}
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(OUT + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        c = acc[k][r["Counter_Name"]]; c[0] += float(r["Counter_Value"]); c[1] += 1
lines = ["| kernel | known read MB | known write MB | FETCH_SIZE MB | FETCH/known | WRITE_SIZE MB | WRITE/known | other counters (per launch) |", "|---|---:|---:|---:|---:|---:|---:|---|"]
for k, (kr, kw) in known.items():
    a = acc.get(k, {})
    g = lambda n: (a[n][0] / a[n][1]) if n in a and a[n][1] else None
    fs, ws = g("FETCH_SIZE"), g("WRITE_SIZE")
    fm = None if fs is None else fs * 1024 / 1e6; wm = None if ws is None else ws * 1024 / 1e6
    oth = ", ".join(f"{n} {g(n):.4g}" for n in sorted(a) if n not in ("FETCH_SIZE", "WRITE_SIZE"))
    f = lambda v: "-" if v is None else f"{v:.1f}"
    r = lambda v, kn: "-" if (v is None or not kn) else f"{v/kn:.2f}"
    lines.append(f"| `{k}` | {kr:.1f} | {kw:.1f} | {f(fm)} | {r(fm, kr)} | {f(wm)} | {r(wm, kw)} | {oth} |")
open(OUT + "/table.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines)); print(open(OUT + "/timing.txt").read())
PY
