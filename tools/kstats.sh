#!/bin/bash
# usage (on the GPU box): tools/kstats.sh <name> -- <command ...>   -> per-kernel calls / average us of the command (rocprofv3 --kernel-trace --stats)
name=$1; shift 2
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
rm -rf $root/gpurun_out/ks_$name
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/ks_$name -o $name -- "$@" > /dev/null 2>&1
f=$(find $root/gpurun_out/ks_$name -name "*kernel_stats.csv" | head -1)
test -n "$f" || { echo "no kernel_stats.csv"; exit 1; }
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-72s calls %5s  avg %9.1f us  %5.1f %%" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
