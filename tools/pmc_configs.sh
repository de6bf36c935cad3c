#!/bin/bash
# usage (on the GPU box): tools/pmc_configs.sh   -> gpurun_out/pmc_configs.txt
# HBM traffic of every kernel of tools/bench_configs.py: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes
# (kernel-trace only), KiB per dispatch averaged per kernel; FETCH_SIZE is doubled per the gfx950 note in
# MI355X_MICROARCH.md (128-B requests tallied as 64 B), WRITE_SIZE is taken as is.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_cfg_$c -o p -- python $R/tools/bench_configs.py > /dev/null 2>&1
done
python - <<PY > $R/gpurun_out/pmc_configs.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$R/gpurun_out/pmc_cfg_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if "uavqp" in r["Kernel_Name"] and r["Counter_Name"] == c:
                acc[r["Kernel_Name"][:70]][c].append(float(r["Counter_Value"]))
print("kernel | dispatches | fetch MB (corrected x2) | write MB | total MB per dispatch")
for k, d in sorted(acc.items()):
    f = 2 * 1024 * sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"])) / 1e6
    w = 1024 * sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"])) / 1e6
    print("%-70s | %3d | %10.2f | %10.2f | %10.2f" % (k, len(d["FETCH_SIZE"]), f, w, f + w))
PY
cat $R/gpurun_out/pmc_configs.txt
