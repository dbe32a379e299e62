#!/bin/bash
# After `gpurun ... bash tools/round6_profiles.sh` (and the GPU suite) have merged their outputs into gpurun_out/: write the committed digests under profiles/.
set -eu
T=${1:-r06}
cd "$(dirname "$0")/.."
python tools/summarize_profiles.py $T | tail -3
for f in bench_n1.json other_sizes.txt rank_profile_s1m.txt rank_profile_waymo4m.txt slab_timing_cull.txt slab_timing_waymo4m.txt bench_pose_inside.json; do cp gpurun_out/$T/$f profiles/${T}_$f; done
python tools/write_resources_md.py $T
python tools/parity_digest.py $T | tail -9
