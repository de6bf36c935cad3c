#!/bin/bash
# usage (on the GPU box): tools/collect_profiles.sh <round-tag>     -> gpurun_out/<tag>/...   (copy what is to be judged into profiles/)
# The bench line (with its other_configs sub-record), the rocprofv3 --kernel-trace --stats summary of the SAME command, the PMC counters of
# the headline kernel, every other BASELINE config as its own full bench line with kernel stats and HBM traffic, the A/B of the two
# general-rows kernels, the soaks, the per-wave timeline of the headline kernel.
#
# REFUSES to run when libuavqp.so was not built from the sources next to it (summaries that predate the last kernel commits):
# the library reports the hash of its sources (uavqp_version()), `make src-hash` recomputes it from the files on this box.  The hash goes
# into <out>/MANIFEST.txt; `git log -1 --format=%h` of the commit whose tree has that hash is what the summaries belong to.
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
LIB_HASH=$(python - <<PY
import ctypes, re, sys
sys.path.insert(0, "$R")
import uav_motion_planning_amd as U
l = U.lib(); l.uavqp_version.restype = ctypes.c_char_p
m = re.search(rb"src ([0-9a-f]+)", l.uavqp_version())
print(m.group(1).decode() if m else "none")
PY
)
SRC_HASH=$(make -s -C $R/uav_motion_planning_amd/csrc src-hash)
if [ "$LIB_HASH" != "$SRC_HASH" ]; then
  echo "collect_profiles.sh: libuavqp.so was built from sources with hash $LIB_HASH, the sources here hash to $SRC_HASH -- rebuild (make -C uav_motion_planning_amd/csrc) first" >&2
  exit 1
fi
mkdir -p $O
{
  echo "round tag: $TAG"
  echo "library source hash (uavqp_version): $LIB_HASH"
  echo "collected: $(date -u +%Y-%m-%dT%H:%M:%SZ) on $(hostname)"
  /opt/rocm/bin/rocminfo 2>/dev/null | grep -m1 "Marketing Name" | sed 's/^ *//'
  echo "commands: see tools/collect_profiles.sh at the commit whose csrc/ hashes to the value above"
} > $O/MANIFEST.txt
cd /tmp; export TMPDIR=/tmp
python $R/bench.py > $O/bench4096.json 2> $O/bench4096.err
python $R/bench.py --steps 20 --warmup 5 > $O/bench4096_steps20.json 2>> $O/bench4096.err          # the driver's invocation
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof4096 -o b -- python $R/bench.py --cpu-sample 0 --no-traffic --no-fp64 --pipelined-streams 0 --no-other-configs > $O/bench4096_under_prof.json 2>> $O/bench4096.err
cp $(find $O/prof4096 -name "*kernel_stats.csv" | head -1) $O/bench4096_kernel_stats.csv
for B in 65536 1048576; do
  python $R/bench.py --batch $B --steps 50 --cpu-sample 0 --pipelined-streams 0 > $O/bench_$B.json 2>> $O/bench4096.err
done
timeout 600 python $R/bench.py --config 3 --steps 10 --warmup 3 --cpu-sample 512 > $O/bench_config3.json 2>> $O/bench4096.err
timeout 600 python $R/bench.py --config 3 --rows 2 --steps 3 --warmup 1 --cpu-sample 0 > $O/bench_config3_rows2.json 2>> $O/bench4096.err
timeout 600 python $R/bench.py --config 4 --steps 50 --cpu-sample 0 > $O/bench_config4.json 2>> $O/bench4096.err
timeout 600 python $R/bench.py --config 5 --steps 10 --cpu-sample 0 > $O/bench_config5.json 2>> $O/bench4096.err
# rocprofv3 --kernel-trace --stats of every other config (the same commands as --inner child runs: kernels only)
for C in "3 0" "3 2" "4 0" "5 0"; do
  set -- $C
  N=config$1; [ "$2" != "0" ] && N=${N}_rows$2
  ( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$N -o c -- python $R/bench.py --config $1 --rows $2 --steps 10 --warmup 3 --inner --repeats 1 --cpu-sample 0 > /dev/null 2>> $O/bench4096.err )
  cp $(find $O/prof_$N -name "*kernel_stats.csv" | head -1) $O/bench_${N}_kernel_stats.csv
done
python $R/tools/bench_configs.py > $O/other_configs.jsonl 2> $O/other_configs.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_configs -o c -- python $R/tools/bench_configs.py > /dev/null 2>> $O/other_configs.err
cp $(find $O/prof_configs -name "*kernel_stats.csv" | head -1) $O/other_configs_kernel_stats.csv
$R/tools/pmc_configs.sh > /dev/null 2>&1; cp $R/gpurun_out/pmc_configs.txt $O/pmc_other_configs.txt
$R/tools/pmc.sh s4k "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" -- --no-traffic --pipelined-streams 0 --graph 0 --steps 50 --repeats 1 --no-other-configs > $O/pmc_bench4096.txt 2>&1
$R/tools/pmc.sh l1m "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU" -- --batch 1048576 --no-traffic --pipelined-streams 0 --graph 0 --steps 10 --repeats 1 > $O/pmc_bench1m.txt 2>&1
cd $R
python tools/rows_ab.py 65536 > $O/rows_ab.jsonl 2>> $O/other_configs.err
python tools/rows_ab.py 64 small >> $O/rows_ab.jsonl 2>> $O/other_configs.err
python tools/config1_latency.py > $O/config1_latency.txt 2>> $O/other_configs.err
make -s -C tools/ubench axis_latency -B > /dev/null 2>&1   # (always against the library of this tree)
[ -x tools/ubench/axis_latency ] && ( echo "C ABI, no Python (tools/ubench/axis_latency.cpp):"; tools/ubench/axis_latency 3000 ) >> $O/config1_latency.txt 2>&1
# soaks: equality / corridor, general rows (every unsolved draw cross-checked against the OSQP port: exit code 1 on a feasible one), pipeline
( python tools/soak.py 300 95; python tools/soak_rows.py 150 95; python tools/soak_rows.py 150 96; python tools/soak_aux.py 100 95 ) > $O/soak.txt 2>&1
# round 6: the rows soak also at the facade's eps_prim_inf = 1e-3 and with the reason codes of a -DUAVQP_ROWS_REASON build (tools/soak_rows_r6.sh; fresh seeds)
tools/soak_rows_r6.sh $O/soak_rows 150 97 98 > $O/soak_rows_r6.txt 2>&1
( python tools/soak_pipeline.py 80 15 ) > $O/soak_pipeline.txt 2>&1
if [ -x tools/ubench/tw/h_16 ]; then
  ( cd tools/ubench/tw; for v in h_16 h_base h_t32; do echo "== $v 4096"; ./$v 4096; done; echo "== h_base 8192"; ./h_base 8192; echo "== h_t32 65536"; ./h_t32 65536 12 ) > $O/headline_timeline.txt 2>&1
fi
rm -rf $O/prof4096 $O/prof_config* $O/prof_configs
ls -la $O
