#!/bin/bash
# usage (on the GPU box): tools/g2_fused_ab.sh <old libuavqp.so>  -- config 4 with the dealing as window_sort_kernel's own launch (a library from before the
# fused ranking) against the waves ranking their own window (round 6), one box, alternating; as ONE hipGraph of K steps (bench.py's default for config 4)
# and with eager launches (--graph 0: what a caller of the C ABI gets).
OLD=${1:-tools/ubench/libuavqp_head.so}
for G in 1 0; do for rep in 1 2; do for L in $OLD uav_motion_planning_amd/libuavqp.so; do
UAVQP_LIB_PATH=$(realpath $L) python bench.py --config 4 --steps 50 --warmup 10 --graph $G --no-fp64 --kernels-only --cpu-sample 0 --no-traffic 2>/dev/null | TAG="graph=$G $L" python -c '
import json, os, sys
x = json.loads(sys.stdin.read().strip().split("\n")[-1])
print(os.environ["TAG"], "step", round(1e3 * x["ms_per_step"], 2), "us", x["parity"]["within_tolerance"], [(k["kernel"][:34], round(k["avg_us"], 1)) for k in x.get("kernels", [])[:3]])'
done; done; done
