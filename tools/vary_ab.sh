#!/bin/bash
# Run on the GPU box: the static and the varying window of bench.py once per library variant.  usage: tools/vary_ab.sh <tag|-> ...
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  if [ "$v" = "-" ]; then unset LRT_HIP_LIB; else export LRT_HIP_LIB=$R/lidar_rt_amd/csrc/liblrt_ab_$v.so; fi
  python $R/bench.py --no-cpu-baseline --no-both-paths --min-seconds 1.5 --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d['value_varying']
print('$v', 'static %.4f ms fwd %.4f' % (d['sustained']['ms_per_step'], d['sustained']['phase_ms']['fwd']), 'varying %.4f ms fwd %.4f' % (v['ms_per_step'], v['phase_ms']['fwd']))"
done
