#!/bin/bash
# Run on the GPU box: bench.py once per library variant.  usage: tools/ab_run.sh <tag|-> ...   ("-" = the default library; tag = liblrt_ab_<tag>.so)
# extra bench arguments through AB_ARGS
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  if [ "$v" = "-" ]; then unset LRT_HIP_LIB; else export LRT_HIP_LIB=$R/lidar_rt_amd/csrc/liblrt_ab_$v.so; fi
  python $R/bench.py --no-cpu-baseline --no-both-paths --no-vary --min-seconds 1.5 --steps 100 --check-sum $AB_ARGS 2>/dev/null | python $R/tools/benchline.py "$v"
done
