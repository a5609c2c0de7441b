#!/usr/bin/env bash
# one workgroup per CU with 64 x 64 wave tiles (variant build, flag 64) against two per CU with 32 x 32 (default)
mkdir -p gpurun_out/p19; o=gpurun_out/p19/tu4.txt; : > $o
for rep in 1 2; do
  echo "== rep $rep default (two per CU)" >> $o
  python tools/bench_fit.py --sizes 192,224,256 --fits 512 --reps 3 >> $o 2>&1
  echo "== rep $rep default lib, one per CU (flag 64)" >> $o
  python tools/bench_fit.py --sizes 192,224,256 --fits 512 --reps 3 --flags 64 >> $o 2>&1
  echo "== rep $rep tu4 lib, one per CU (flag 64)" >> $o
  python tools/bench_fit.py --sizes 192,224,256 --fits 512 --reps 3 --flags 64 --lib libgapro_hip_tu4.so >> $o 2>&1
done
grep -E "^==|M=" $o | awk '{ if ($1=="==") print; else print $1,$2,$9,$10,$11,$12,$13,$14 }'
