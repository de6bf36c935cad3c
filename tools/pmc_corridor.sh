#!/bin/bash
# usage (on the GPU box): tools/pmc_corridor.sh   -> gpurun_out/pmc_corridor.txt
# SQ / LDS / TCC counters of the corridor kernels (tools/corridor_bench.py), one rocprofv3 pass per counter group (kernel-trace only).
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_IFETCH" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
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
        print("   %-26s mean %.5g  (n=%d, min %.5g max %.5g)" % (c, sum(v) / len(v), len(v), min(v), max(v)))
PY
cat $R/gpurun_out/pmc_corridor.txt
