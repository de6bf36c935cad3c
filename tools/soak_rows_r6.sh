#!/bin/bash
# Round 6: the rows soak three ways on the same seeds -- (a) as in round 5 (eps_prim_inf 1e-9: every certificate
# counts), (b) at the facade's / the reference's eps_prim_inf = 1e-3 (minimum_control.cpp:161) with the verdicts tallied only, (c) the
# one-lane kernel of a -DUAVQP_ROWS_REASON build: the iteration count of a problem that ends undecided carries 1000 x its reason code
# (1: the direction solve itself inconsistent; 2: the direction did not vanish and nothing blocks; 3: certificate below the margin;
#  4: singular after the restart; 5: iteration cap).   usage: tools/soak_rows_r6.sh <out-dir> [draws] [seeds...]
set -u
OUT=${1:-gpurun_out/soak_r6}; DRAWS=${2:-150}; shift 2 2>/dev/null || true
SEEDS=${*:-"95 96"}
mkdir -p "$OUT"
R=$(cd "$(dirname "$0")/.." && pwd)
for s in $SEEDS; do
  python3 $R/tools/soak_rows.py $DRAWS $s > $OUT/rows_eps1e-9_seed$s.txt 2>&1; echo "rc=$?" >> $OUT/rows_eps1e-9_seed$s.txt
  cp $R/gpurun_out/soak_rows_unsolved_seed$s.npy $OUT/unsolved_eps1e-9_seed$s.npy 2>/dev/null
  UAVQP_SOAK_EPS=1e-3 UAVQP_SOAK_LENIENT=1 python3 $R/tools/soak_rows.py $DRAWS $s > $OUT/rows_eps1e-3_seed$s.txt 2>&1; echo "rc=$?" >> $OUT/rows_eps1e-3_seed$s.txt
  cp $R/gpurun_out/soak_rows_unsolved_seed$s.npy $OUT/unsolved_eps1e-3_seed$s.npy 2>/dev/null
  if [ -f $R/tools/ubench/libuavqp_reason.so ]; then
    UAVQP_LIB_PATH=$R/tools/ubench/libuavqp_reason.so UAVQP_SOAK_LANES=1 UAVQP_SOAK_LENIENT=1 python3 $R/tools/soak_rows.py $DRAWS $s > $OUT/rows_reason_seed$s.txt 2>&1; echo "rc=$?" >> $OUT/rows_reason_seed$s.txt
    cp $R/gpurun_out/soak_rows_unsolved_seed$s.npy $OUT/unsolved_reason_seed$s.npy 2>/dev/null
  fi
done
tail -n 3 $OUT/*.txt
