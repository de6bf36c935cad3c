#!/bin/bash
# usage (on the GPU box): tools/pmc_traffic.sh <tag> -- <bench.py arguments>   -> gpurun_out/pmc_traffic_<tag>.txt
# HBM traffic PER KERNEL of one bench.py child run (--inner: kernels only): FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (kernel-trace
# only), KiB per dispatch averaged per kernel; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md, WRITE_SIZE as is (as tools/pmc_configs.sh).
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift; [ "$1" = "--" ] && shift
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_tr_${TAG}_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_tr_${TAG}_$c -o p -- python $R/bench.py "$@" --inner --repeats 1 --graph 0 --steps 2 --warmup 1 > /dev/null 2>&1
done
python - <<PY > $R/gpurun_out/pmc_traffic_$TAG.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$R/gpurun_out/pmc_tr_${TAG}_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if "uavqp" in r["Kernel_Name"] and r["Counter_Name"] == c:
                acc[r["Kernel_Name"][:70]][c].append(float(r["Counter_Value"]))
print("bench.py $* --inner: kernel | dispatches | fetch MB (corrected x2) | write MB | total MB per dispatch")
tot = 0.0
for k, d in sorted(acc.items()):
    f = 2 * 1024 * sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"])) / 1e6
    w = 1024 * sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"])) / 1e6
    tot += f + w
    print("%-70s | %3d | %10.2f | %10.2f | %10.2f" % (k, len(d["FETCH_SIZE"]), f, w, f + w))
print("sum over the kernels (one dispatch each): %.2f MB" % tot)
PY
rm -rf $R/gpurun_out/pmc_tr_${TAG}_FETCH_SIZE $R/gpurun_out/pmc_tr_${TAG}_WRITE_SIZE
cat $R/gpurun_out/pmc_traffic_$TAG.txt
