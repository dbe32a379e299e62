#!/bin/bash
# Run on the GPU box (via gpurun): per-kernel average durations of a short bench run.  usage: tools/kstats.sh <tag> [bench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/ks_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-both-paths --no-vary --min-seconds 0 "$@" > $OUT/bench.json 2> $OUT/err.txt
find $OUT -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)
if not f: print("no stats file"); raise SystemExit
for r in list(csv.DictReader(open(f[0])))[:18]: print(r["Name"][:70].ljust(70), r["Calls"].rjust(5), ("%.1f" % (float(r["AverageNs"])/1000)).rjust(8), r["Percentage"])
PY
