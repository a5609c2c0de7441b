#!/usr/bin/env bash
# A/B of a fit-kernel change: libgapro_hip.so (new) against libgapro_hip_prev.so (GAPRO_VARIANT=prev build of the parent
# commit), all kernel families.  usage: tools/r03_assoc_ab.sh [tests]   (tests: pytest selection, default the fit tests)
set -u
mkdir -p gpurun_out/assoc
out=gpurun_out/assoc/ab.txt
: > $out
python -m pytest ${1:-tests/test_fit_gpu.py tests/test_svgp_kat.py} -m gpu -x -q > gpurun_out/assoc/pytest.txt 2>&1
tail -3 gpurun_out/assoc/pytest.txt
for rep in 1 2; do
  for lib in "" libgapro_hip_prev.so; do
    echo "== rep $rep lib ${lib:-default}" >> $out
    python tools/bench_fit.py --sizes 32,64,96,128 --fits 2048 --reps 3 ${lib:+--lib $lib} >> $out 2>&1
    python tools/bench_fit.py --sizes 160,256,320,384,448 --fits 512 --reps 3 ${lib:+--lib $lib} >> $out 2>&1
    python tools/bench_fit.py --sizes 512,1024,2048 --fits 32 --reps 2 ${lib:+--lib $lib} >> $out 2>&1
  done
done
grep -E "^==|M=" $out | awk '{ if ($1=="==") print; else print $1,$2,$9,$10,$13,$14 }'
