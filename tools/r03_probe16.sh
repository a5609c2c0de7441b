#!/usr/bin/env bash
# copy-free product forms (KMIN, M_p <= 256): new library against libgapro_hip_prev.so
mkdir -p gpurun_out/p16; o=gpurun_out/p16/km.txt; : > $o
python -m pytest tests/test_fit_gpu.py tests/test_svgp_kat.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do for lib in "" libgapro_hip_prev.so; do
  echo "== rep $rep lib ${lib:-default}" >> $o
  python tools/bench_fit.py --sizes 144,160,192,224,256,288,320,384,448 --fits 512 --reps 3 ${lib:+--lib $lib} >> $o 2>&1
done; done
grep -E "^==|M=" $o | awk '{ if ($1=="==") print; else print $1,$2,$9,$10,$11,$12,$13,$14 }'
