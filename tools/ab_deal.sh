#!/bin/bash
# usage (on the GPU box): tools/ab_deal.sh [out.log]   -- A/B of the dealing of the two dual preludes on one box: UAVQP_DEAL_TICKETS=0 (round-robin) against 1
# (tickets, the default) on config 3, config 5 and config 3 + K = 2 rows; per run the step and the four longest kernels.
OUT=${1:-gpurun_out/ab_deal.log}
: > $OUT
for t in 0 1 0 1; do
  for c in "3 0" "5 0" "3 2"; do set -- $c
    UAVQP_DEAL_TICKETS=$t python bench.py --config $1 --rows $2 --steps 6 --warmup 2 --no-fp64 --kernels-only --cpu-sample 0 --no-traffic 2>/dev/null | TAG="tickets=$t config $1 rows $2" python -c '
import json, os, sys
x = json.loads(sys.stdin.read().strip().split("\n")[-1])
print(os.environ["TAG"], round(x["ms_per_step"], 4), x["parity"]["within_tolerance"], [(k["kernel"][:34], round(k["avg_us"], 1)) for k in x.get("kernels", [])[:4]])' >> $OUT
  done
done
cat $OUT
