#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes of the default bench command.
# Outputs land in gpurun_out/prof_<tag>/ ; tools/summarize_profiles.py turns them into profiles/*.md + pmc_traffic.json
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
BENCH="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-both-paths --no-stats-step --no-vary --no-deferred --min-seconds 0"
timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $BENCH > $OUT/bench_under_trace.json 2> $OUT/stats.err
SHORT="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-both-paths --no-stats-step --no-vary --no-deferred --min-seconds 0"
timeout -k 5 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $SHORT > /dev/null 2> $OUT/pmc_fetch.err
timeout -k 5 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $SHORT > /dev/null 2> $OUT/pmc_write.err
timeout -k 5 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -o p -- $SHORT > /dev/null 2> $OUT/pmc_sq.err
timeout -k 5 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -o p -- $SHORT > /dev/null 2> $OUT/pmc_l2.err
timeout -k 5 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_lds -o p -- $SHORT > /dev/null 2> $OUT/pmc_lds.err
find $OUT -name "*.csv" | head -30
# keep the merge small: drop the per-dispatch kernel trace of the stats run except the stats files
find $OUT -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT
