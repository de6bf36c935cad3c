#!/bin/bash
# usage (on the GPU box): tools/ktrace.sh <name> <last-N> -- <command ...>   -> the last N kernel dispatches of the command in time order
# (duration in us, gap to the previous dispatch's end, kernel name) from rocprofv3 --kernel-trace: what a multi-kernel call spends where
name=$1; last=$2; shift 3
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
rm -rf $root/gpurun_out/kt_$name
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/kt_$name -o $name -- "$@" > /dev/null 2>&1
f=$(find $root/gpurun_out/kt_$name -name "*kernel_trace.csv" | head -1)
test -n "$f" || { echo "no kernel_trace.csv"; exit 1; }
python - "$f" "$last" <<'PY'
import csv, sys, re
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
rows = rows[-int(sys.argv[2]):]
prev = rows[0][0]
for s, e, nm in rows:
    m = re.search(r"(uavqp::)?([A-Za-z0-9_]+)(<[^>]*>)?", nm.replace("void ", ""))
    print("%8.1f us  gap %6.1f  %s" % ((e - s) / 1e3, (s - prev) / 1e3, (m.group(2) + (m.group(3) or "")) if m else nm[:50]))
    prev = e
print("span %.1f us" % ((rows[-1][1] - rows[0][0]) / 1e3))
PY
