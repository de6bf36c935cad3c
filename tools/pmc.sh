#!/bin/bash
# usage (on the GPU box): tools/pmc.sh <outname> "<counters>" -- <bench args>
# Collects PMC counters for the solve kernels in their own rocprofv3 pass (kernel-trace only).
name=$1; ctrs=$2; shift 3
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$name -o $name -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --no-fp64 "$@" > /dev/null 2>&1
python - <<PY
import csv,collections,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_$name/*counter_collection.csv")[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"][:60]
    if "uavqp" not in k: continue
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    n[(k,r["Counter_Name"])]+=1
for k,d in acc.items():
    print(k)
    for c,v in d.items(): print("   %-28s %.4g per-dispatch" % (c, v/n[(k,c)]))
PY
