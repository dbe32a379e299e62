#!/bin/bash
# Run on the GPU box (via gpurun): PMC counters of one kernel of a short bench run.  usage: tools/kpmc.sh <kernel-substring> "<counters>" [bench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
K=$1; C=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/kpmc_tmp
rm -rf $OUT; mkdir -p $OUT
timeout -k 5 200 rocprofv3 --pmc $C --output-format csv -d $OUT -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-both-paths --min-seconds 0 "$@" > /dev/null 2> $OUT/err.txt
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
if not f: print("no counter file"); print(open("$OUT/err.txt").read()[-600:]); raise SystemExit
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    if "$K" in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"][:40]].add(r["Dispatch_Id"])
for k,v in acc.items():
    print(k, "dispatches", len(n[k]))
    for c,x in v.items(): print("   %-28s %.4g per launch" % (c, x/len(n[k])))
PY
