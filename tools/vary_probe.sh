for o in "" "--opt learn_slab=0"; do python bench.py --no-cpu-baseline --no-both-paths --min-seconds 1.5 --steps 100 $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d['value_varying']
print('$o', 'fixed', round(d['sustained']['value']/1e6,2), d['sustained']['phase_ms'], 'varying', round(v['value']/1e6,2), v['phase_ms'])"; done
