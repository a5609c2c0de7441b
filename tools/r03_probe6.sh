#!/usr/bin/env bash
R=$PWD; O=$R/gpurun_out/r03p6; mkdir -p $O
python tools/bench_fit.py --profile --sizes 256 --fits 256 --reps 1 > $O/p_new.log 2>&1
python tools/bench_fit.py --profile --lib libgapro_hip_m16.so --sizes 256 --fits 256 --reps 1 > $O/p_m16.log 2>&1
python tools/bench_fit.py --profile --lib libgapro_hip_noload.so --sizes 256 --fits 256 --reps 1 > $O/p_noload.log 2>&1
python tools/bench_fit.py --profile --sizes 256 --fits 64 --reps 1 > $O/p_new64.log 2>&1
python tools/bench_fit.py --profile --sizes 256 --fits 64 --reps 1 --flags 8192 > $O/p_old64.log 2>&1
cat $O/p_new.log $O/p_m16.log $O/p_noload.log $O/p_new64.log $O/p_old64.log
