#!/bin/bash
# CPU only: compile lrt_kernels.hip for the device alone with extra -D flags and print the register / spill / LDS table of the kernels matching a pattern.
# usage: tools/kres.sh <pattern> [-DMACRO=VALUE ...]
set -e
pat=$1; shift
cd "$(dirname "$0")/../lidar_rt_amd/csrc"
out=$(mktemp /tmp/kres.XXXXXX.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fno-slp-vectorize -Wno-unused-value --cuda-device-only --no-gpu-bundle-output -c "$@" -o $out lrt_kernels.hip
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $out | python3 -c "
import re,sys,subprocess
txt=sys.stdin.read()
for b in txt.split('\n  - .agpr_count:')[1:]:
    g=lambda k:(re.search(r'\.'+k+r':\s+(\S+)',b) or [None,None])[1]
    n=subprocess.run(['c++filt',g('name')],capture_output=True,text=True).stdout.strip(); n=re.sub(r'\(.*','',n)
    if re.search(sys.argv[1],n): print('%-44s vgpr %4s sgpr %4s vspill %3s sspill %3s scratch %5s lds %6s'%(n[:44],g('vgpr_count'),g('sgpr_count'),g('vgpr_spill_count'),g('sgpr_spill_count'),g('private_segment_fixed_size'),g('group_segment_fixed_size')))
" "$pat"
rm -f $out
