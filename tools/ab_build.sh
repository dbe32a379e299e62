#!/bin/bash
# Build an A/B variant of the library next to the default one:  tools/ab_build.sh NAME [-DMACRO=VALUE ...]
# -> lidar_rt_amd/csrc/liblrt_ab_NAME.so, selected at run time with LRT_HIP_LIB=<path> (git-ignored, travels with gpurun).
set -e
cd "$(dirname "$0")/../lidar_rt_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -fno-slp-vectorize -Wno-unused-value "$@" \
    -o liblrt_ab_${name}.so lrt_kernels.hip lrt_chamfer.hip lrt_preprocess.hip
echo "built liblrt_ab_${name}.so"
