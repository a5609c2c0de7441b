#!/usr/bin/env bash
mkdir -p gpurun_out/p21; o=gpurun_out/p21/ab.txt; : > $o
for rep in 1 2; do for lib in "" libgapro_hip_prev.so; do
  echo "== rep $rep lib ${lib:-default}" >> $o
  python tools/bench_fit.py --sizes ${1:-144,160,192,208,224,256,304,384} --fits 512 --reps 3 ${lib:+--lib $lib} >> $o 2>&1
done; done
grep -E "^==|M=" $o | awk '{ if ($1=="==") print; else print $1,$2,$9,$10,$11,$12,$13,$14 }'
