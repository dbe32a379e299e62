#!/bin/bash
# Everything the round-5 digest needs, in one GPU call (about 9 minutes): kernel stats + PMC passes of the bench command, the default bench line,
# the other sizes, one rank's step of an N-way split, all eight ranks with the forward's tile statistics, the pose-inside (near-ray) frame.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r06; mkdir -p $O
bash tools/collect_profiles.sh r06 > $O/collect.log 2>&1
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
for w in s10k s200k waymo4m; do python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --no-both-paths --min-seconds 1.5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d.get('value_varying') or {}
print(d['config']['workload'][:14], 'exact-window', round(d['value']), round(d['ms_per_step'],4), 'sustained', round(d['sustained']['value']), d['sustained']['phase_ms'], 'varying', round(v.get('value',0)))"; done > $O/other_sizes.txt
CULL=1 python tools/slab_timing.py 2>/dev/null | grep "^N=" > $O/slab_timing_cull.txt
WORKLOAD=waymo4m CULL=1 SLAB_OWNER=0 python tools/slab_timing.py 2>/dev/null | grep "^N=" > $O/slab_timing_waymo4m.txt
python tools/rank_profile.py 2>/dev/null | grep "rank" > $O/rank_profile_s1m.txt
WORKLOAD=waymo4m python tools/rank_profile.py 2>/dev/null | grep "rank" > $O/rank_profile_waymo4m.txt
python bench.py --no-cpu-baseline --no-both-paths --pose-inside --min-seconds 1.5 2>/dev/null > $O/bench_pose_inside.json
cat $O/other_sizes.txt; tail -2 $O/slab_timing_cull.txt; tail -2 $O/slab_timing_waymo4m.txt; head -c 400 $O/bench_n1.json
