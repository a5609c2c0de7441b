#!/usr/bin/env bash
R=$PWD; O=$R/gpurun_out/r03p7; mkdir -p $O
python tools/bench_fit.py --profile --sizes 256 --fits 256 --reps 1 > $O/p_new.log 2>&1
python tools/bench_fit.py --profile --sizes 256 --fits 64 --reps 1 >> $O/p_new.log 2>&1
cat $O/p_new.log
