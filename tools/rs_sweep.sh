#!/bin/bash
# developer tool: the three workloads with own_sort = 0 / 1 / 2
for w in s10k s200k s1m; do for v in 0 1 2; do
python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --min-seconds 0 --check-sum --opt own_sort=$v 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], 'own_sort', sys.argv[2], round(d['value']), {k: round(v, 4) for k, v in d['phase_ms'].items()}, round(d['checksums']['out'],2))" $w $v
done; done
