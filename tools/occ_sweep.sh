#!/bin/bash
# forward time against workgroups per CU (k_fwd_cr4): how much of the kernel is latency that more resident waves would hide
for v in 4 3 2 1; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --min-seconds 0 --opt wg4_per_cu=$v 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wg4_per_cu', sys.argv[1], 'rays/s', round(d['value']), 'forward ms', round(d['phase_ms']['forward'], 4))" $v
done
