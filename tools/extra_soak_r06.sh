mkdir -p gpurun_out/r06x
( for s in 101 102 103 104 105 106; do
    UAVQP_SOAK_EPS=1e-3 UAVQP_SOAK_LENIENT=1 python3 tools/soak_rows.py 150 $s 2>&1 | tail -2; echo "rc=$? (seed $s, eps_prim_inf 1e-3)"
  done ) > gpurun_out/r06x/soak_rows_eps1e-3_extra.txt 2>&1
( python tools/soak.py 300 97; python tools/soak.py 300 98; python tools/soak_aux.py 100 97; python tools/soak_pipeline.py 80 17; python tools/soak_tiles.py 2>&1 | tail -3 ) > gpurun_out/r06x/soak_extra.txt 2>&1
tail -n 4 gpurun_out/r06x/soak_rows_eps1e-3_extra.txt gpurun_out/r06x/soak_extra.txt
