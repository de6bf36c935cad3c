#!/bin/bash
# usage (GPU box): tools/ab_libs.sh <libA.so> <libB.so> <kernel-name regex> -- <bench.py arguments>
# A/B of two builds of libuavqp.so on ONE box (boxes differ by a few percent): rocprofv3 --kernel-trace --stats of the same bench.py command
# with each library (UAVQP_LIB_PATH), alternating A B A B; prints calls and average duration of the kernels that match.
# Build the older side from a commit:  git archive <commit> uav_motion_planning_amd/csrc include | tar -x -C /tmp/old && (cd /tmp/old/uav_motion_planning_amd/csrc
#   && hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DUAVQP_SRC_HASH=\"old\" -o <repo>/tools/ubench/libuavqp_head.so uavqp.hip)
A=$1; B=$2; PAT=$3; shift 4
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do
  for L in "$A" "$B"; do
    rm -rf /tmp/kt_ab
    UAVQP_LIB_PATH=$(realpath $R/$L 2>/dev/null || echo $L) timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_ab -o x -- python $R/bench.py "$@" --inner --repeats 1 --cpu-sample 0 > /dev/null 2>&1
    echo "== $L"; python3 -c "
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r['Name']): print('   %-90s calls %4s  avg %10.1f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3))
" $(find /tmp/kt_ab -name "*kernel_stats.csv" | head -1) "$PAT" 
  done
done
