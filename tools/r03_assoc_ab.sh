#!/usr/bin/env bash
# A/B of the G_Kzz re-association (round 3): libgapro_hip.so (new) against libgapro_hip_oldassoc.so, all kernel families.
set -u
mkdir -p gpurun_out/assoc
out=gpurun_out/assoc/ab.txt
: > $out
python -m pytest tests -m gpu -x -q > gpurun_out/assoc/pytest.txt 2>&1
tail -3 gpurun_out/assoc/pytest.txt
for rep in 1 2; do
  for lib in "" libgapro_hip_oldassoc.so; do
    echo "== rep $rep lib ${lib:-default}" >> $out
    python tools/bench_fit.py --sizes 32,64,96,128 --fits 2048 --reps 3 ${lib:+--lib $lib} >> $out 2>&1
    python tools/bench_fit.py --sizes 512,1024,2048 --fits 32 --reps 2 ${lib:+--lib $lib} >> $out 2>&1
  done
done
grep -E "^==|M=|TFLOP" $out | head -80
