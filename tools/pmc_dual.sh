#!/bin/bash
# usage (on the GPU box): tools/pmc_dual.sh   -> gpurun_out/pmc_dual.txt
# What bounds the two dual preludes (corridor_dual_kernel on config 3, rows_dual_kernel on config 3 + K = 2 rows): issue slots, the LDS
# pipe of the CU, or waiting.  One rocprofv3 --pmc pass (kernel-trace only) per counter group around bench.py's kernels-only child run.
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/pmc_dual.txt
: > $out
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
G2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS"
G3="GRBM_GUI_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN"
i=0
for cfg in "--rows 2" "--rows 0"; do
  for grp in "$G1" "$G2" "$G3"; do
    i=$((i+1))
    echo "== bench.py --config 3 $cfg --inner : $grp" >> $out
    timeout 240 bash $R/tools/pmc_cmd.sh dual$i "$grp" dual_kernel -- python $R/bench.py --config 3 $cfg --steps 2 --warmup 1 --inner --repeats 1 --graph 0 --pipelined-streams 0 --no-allgather >> $out 2>&1
  done
done
cat $out
