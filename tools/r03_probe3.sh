#!/usr/bin/env bash
R=$PWD; O=$R/gpurun_out/r03p3; mkdir -p $O
GAPRO_FIT_FLAGS=8192 python tools/ab_bitwise.py --out $O/old.npz > $O/ab.log 2>&1
python tools/ab_bitwise.py --out $O/new.npz >> $O/ab.log 2>&1
python tools/ab_bitwise.py --compare $O/old.npz $O/new.npz >> $O/ab.log 2>&1
cat $O/ab.log
python tools/bench_fit.py --profile --sizes 200,256,320,448 --fits 256 --reps 2 > $O/prof_new.log 2>&1
python tools/bench_fit.py --profile --sizes 256,320 --fits 256 --reps 2 --flags 8192 > $O/prof_old.log 2>&1
cat $O/prof_new.log $O/prof_old.log
rm -f $O/*.npz
