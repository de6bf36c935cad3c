#!/bin/bash
# usage (on the GPU box): tools/g2_probe_ab.sh [out.log]  -- config 4 (32 768 ragged trajectories) with the probe builds of solve_generic2_kernel:
#   -DG2_NO_WS   the HBM workspace round trip of the forward-sweep records compiled out,  -DG2_NO_OUT  the coefficient stores compiled out,  both.
# (Results of the probe builds are wrong by construction; the step and the kernel's average are what is read.)  VERDICT r5 item 4: what a kernel
# WITHOUT workspace records -- the register-resident bodies the item proposes -- could gain at the very most.
# Build:  make -C uav_motion_planning_amd/csrc EXTRA=-DG2_NO_WS OUT=../../tools/ubench/libuavqp_g2nows.so   (… -DG2_NO_OUT … g2noout.so, both … g2none.so)
OUT=${1:-gpurun_out/g2_probe_ab.log}
: > $OUT
for rep in 1 2; do
  for L in uav_motion_planning_amd/libuavqp.so tools/ubench/libuavqp_g2nows.so tools/ubench/libuavqp_g2noout.so tools/ubench/libuavqp_g2none.so; do
    [ -f $L ] || continue
    UAVQP_LIB_PATH=$(realpath $L) python bench.py --config 4 --steps 50 --warmup 10 --no-fp64 --kernels-only --cpu-sample 0 --no-traffic 2>/dev/null | TAG="$L" python -c '
import json, os, sys
x = json.loads(sys.stdin.read().strip().split("\n")[-1])
print(os.environ["TAG"], "step", round(1e3 * x["ms_per_step"], 2), "us", [(k["kernel"][:40], round(k["avg_us"], 1)) for k in x.get("kernels", [])[:3]])' >> $OUT
  done
done
cat $OUT
