#!/bin/bash
# usage (on the GPU box): tools/pmc_cmd.sh <outname> "<counters>" <kernel substring> -- <command ...>
# One rocprofv3 --pmc pass (kernel-trace only) around any command; prints the per-dispatch means of the counters for the kernels
# whose name contains the substring.
name=$1; ctrs=$2; pat=$3; shift 4
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $root/gpurun_out/pmc_$name -o $name -- "$@" > /dev/null 2>&1
python - <<PY
import csv,collections,glob
f=glob.glob("$root/gpurun_out/pmc_$name/**/*counter_collection.csv", recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"][:70]
    if "$pat" not in k: continue
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    n[(k,r["Counter_Name"])]+=1
for k,d in acc.items():
    print(k)
    for c,v in d.items(): print("   %-28s %.5g per-dispatch (%d)" % (c, v/n[(k,c)], n[(k,c)]))
PY
