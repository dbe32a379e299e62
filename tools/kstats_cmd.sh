#!/bin/bash
# Run on the GPU box (via gpurun): per-kernel average durations of an arbitrary command.  usage: tools/kstats_cmd.sh <tag> <command...>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/ks_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- "$@" > $OUT/cmd.out 2> $OUT/err.txt
find $OUT -name "*kernel_trace.csv" -delete
OUT=$OUT python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["OUT"] + "/**/*kernel_stats.csv", recursive=True)
if not f: print("no stats file"); raise SystemExit
for r in list(csv.DictReader(open(f[0])))[:28]: print(r["Name"][:70].ljust(70), r["Calls"].rjust(5), ("%.1f" % (float(r["AverageNs"]) / 1000)).rjust(8), r["Percentage"])
PY
tail -5 $OUT/cmd.out
