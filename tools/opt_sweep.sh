#!/bin/bash
# Run on the GPU box: bench.py once per value of one library option.  usage: tools/opt_sweep.sh <option> <v1> <v2> ...   (S1M, direct path)
O=$1; shift
for v in "$@"; do python bench.py --no-cpu-baseline --no-both-paths --min-seconds 1.5 --opt $O=$v 2>/dev/null | python tools/benchline.py "$O=$v"; done
