#!/bin/bash
# usage: tools/lib_sweep.sh name1 name2 ...   -- bench S1M with the A/B libraries lidar_rt_amd/csrc/liblrt_ab_<name>.so ("default" = the product library)
for n in "$@"; do
  lib=$PWD/lidar_rt_amd/csrc/liblrt_ab_$n.so; [ "$n" = default ] && lib=$PWD/lidar_rt_amd/csrc/liblrt_hip.so
  LRT_HIP_LIB=$lib python bench.py --steps 20 --warmup 3 --no-cpu-baseline --min-seconds 0 --check-sum 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], 'rays/s', round(d['value']), 'build', round(d['phase_ms']['build'], 4), 'fwd', round(d['phase_ms']['forward'], 4), 'bwd', round(d['phase_ms']['backward'], 4), round(d['checksums']['out'], 2))" $n
done
