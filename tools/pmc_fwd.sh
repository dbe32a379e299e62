#!/bin/bash
# rocprofv3 PMC passes over tools/fwd_only.py (build + forward of S1M): what saturates in k_fwd_cr4?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_fwd; rm -rf $OUT; mkdir -p $OUT
run() { rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -o p -- python $R/tools/fwd_only.py > /dev/null 2> $OUT/$1.err; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU"
run b "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM"
run c "SQ_IFETCH SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/pmc_fwd/*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_fwd_cr4" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
v = {k: a[0] / a[1] for k, a in acc.items()}
for k in sorted(v): print(f"{k:28s} {v[k]:.4g}")
wc = v.get("SQ_WAVE_CYCLES", 0); bc = v.get("SQ_BUSY_CYCLES", 0)
print("per wave-cycle: VALU %.3f LDS %.3f SCA %.3f VMEM %.3f | wait_any %.3f wait_inst %.3f active %.3f" % tuple(v.get(k, 0) / wc for k in
      ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")))
PY
