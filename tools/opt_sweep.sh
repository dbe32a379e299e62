#!/bin/bash
# usage: tools/opt_sweep.sh name v1 v2 ...   -- bench S1M with library option name=v
n=$1; shift
for v in "$@"; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --min-seconds 0 --check-sum --opt $n=$v 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], sys.argv[2], 'rays/s', round(d['value']), 'fwd ms', round(d['phase_ms']['forward'], 4), 'bwd', round(d['phase_ms']['backward'], 4), d['hip_counters_per_step']['nodes_visited'], d['hip_counters_per_step']['prims_tested'], round(d['checksums']['out'], 2))" $n $v
done
