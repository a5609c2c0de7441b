#!/usr/bin/env bash
# M_p in steps of 16 up to 336: new library against libgapro_hip_prev.so at off-grid and on-grid sizes
mkdir -p gpurun_out/p17; o=gpurun_out/p17/pad16.txt; : > $o
python -m pytest tests/test_fit_gpu.py tests/test_svgp_kat.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do for lib in "" libgapro_hip_prev.so; do
  echo "== rep $rep lib ${lib:-default}" >> $o
  python tools/bench_fit.py --sizes 32,48,64,96,128,160,208,256,320,384 --fits 1024 --reps 3 ${lib:+--lib $lib} >> $o 2>&1
done; done
grep -E "^==|M=" $o | awk '{ if ($1=="==") print; else print $1,$2,$9,$10,$11,$12,$13,$14 }'
