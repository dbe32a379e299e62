#!/bin/bash
# usage: tools/kernel_times.sh <tag> [bench args...]   (run on the GPU box; prints per-step kernel times from rocprofv3)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/$tag -- python /root/repo/bench.py --no-cpu-baseline --no-both-paths --steps 10 --warmup 2 --min-seconds 0 "$@" > /root/repo/gpurun_out/$tag.bench.log 2>&1
cd /root/repo
python - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/{sys.argv[1]}/*/*kernel_stats.csv")[0]
tot = 0
for r in csv.DictReader(open(f)):
    c = int(r["Calls"])
    if c >= 13:
        per = float(r["AverageNs"]) * c / 13 / 1e3; tot += per
        if per > 15: print(r["Name"][:48].ljust(48), c, round(float(r["AverageNs"]) / 1e3, 1), round(per, 1))
print("sum per step us", round(tot, 1))
PY
