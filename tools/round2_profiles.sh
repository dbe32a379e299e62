#!/bin/bash
# Everything the round-2 profile digest needs, in one GPU call (about 3 minutes): kernel stats + PMC passes of the bench command,
# the per-segment cycle profile of k_fwd_cr4, the per-rank slab timings, the occupancy sweep, the default bench line.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash tools/collect_profiles.sh r02 > gpurun_out/collect_r02.log 2>&1
C4_PROF=1 LRT_HIP_LIB=$R/lidar_rt_amd/csrc/liblrt_ab_prof.so python tools/tile_profile.py > gpurun_out/r02_tile_segments.txt 2>&1
python tools/slab_timing.py > gpurun_out/r02_slab_timing.txt 2>&1
CULL=1 python tools/slab_timing.py 2>/dev/null | grep "^N=[48] rank" > gpurun_out/r02_slab_timing_cull.txt
bash tools/occ_sweep.sh > gpurun_out/r02_occ_sweep.txt 2>&1
python bench.py > gpurun_out/r02_bench_n1.json 2> gpurun_out/r02_bench_n1.err
for w in s10k s200k; do python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --min-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:6], round(d['value']), round(d['ms_per_step'], 4), d['phase_ms'])"; done > gpurun_out/r02_other_sizes.txt
python tools/big_scene_check.py > gpurun_out/r02_big_scene.txt 2>&1
tail -3 gpurun_out/r02_slab_timing.txt; cat gpurun_out/r02_other_sizes.txt; head -c 600 gpurun_out/r02_bench_n1.json
