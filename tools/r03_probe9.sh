#!/usr/bin/env bash
R=$PWD; O=$R/gpurun_out/r03p9; mkdir -p $O
python tools/bench_fit.py --profile --lib libgapro_hip_noload.so --sizes 256,384 --fits 256 --reps 1 > $O/p_noload.log 2>&1
python tools/bench_fit.py --profile --sizes 256,384 --fits 64 --reps 1 > $O/p_new64.log 2>&1
cat $O/p_noload.log $O/p_new64.log
