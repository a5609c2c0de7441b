#!/usr/bin/env bash
R=$PWD; O=$R/gpurun_out/r03p4; mkdir -p $O
for v in noload nomfma; do
python tools/bench_fit.py --profile --lib libgapro_hip_$v.so --sizes 256,320 --fits 256 --reps 1 > $O/prof_$v.log 2>&1
python tools/bench_fit.py --profile --lib libgapro_hip_$v.so --sizes 256 --fits 512 --reps 1 >> $O/prof_$v.log 2>&1
done
python tools/bench_fit.py --profile --sizes 256 --fits 512 --reps 1 > $O/prof_new512.log 2>&1
python tools/bench_fit.py --profile --sizes 256 --fits 512 --reps 1 --flags 8192 > $O/prof_old512.log 2>&1
cat $O/prof_noload.log $O/prof_nomfma.log $O/prof_new512.log $O/prof_old512.log
