#!/bin/bash
# usage (on the GPU box): tools/pmc_corridor.sh [group ...]   -> gpurun_out/pmc_corridor.txt
# SQ / LDS / TCP / TCC counters of the corridor kernels (tools/corridor_bench.py), one rocprofv3 pass per counter group
# (kernel-trace only).  Without arguments: the default groups below.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
if [ $# -eq 0 ]; then
  set -- "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
         "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" \
         "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"
fi
rm -rf $R/gpurun_out/pmc_corr_*
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_corr_$i -o p -- python $R/tools/corridor_bench.py 3 > /dev/null 2>$R/gpurun_out/pmc_corr_$i.err
done
python - <<PY > $R/gpurun_out/pmc_corridor.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_corr_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "corridor" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        print("   %-30s mean %.5g  (n=%d, min %.5g max %.5g)" % (c, sum(v) / len(v), len(v), min(v), max(v)))
PY
cat $R/gpurun_out/pmc_corridor.txt
