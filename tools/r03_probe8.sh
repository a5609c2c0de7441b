#!/usr/bin/env bash
R=$PWD; O=$R/gpurun_out/r03p8; mkdir -p $O
GAPRO_FIT_FLAGS=8192 python tools/ab_bitwise.py --out $O/old.npz > $O/ab.log 2>&1
python tools/ab_bitwise.py --out $O/new.npz >> $O/ab.log 2>&1
python tools/ab_bitwise.py --compare $O/old.npz $O/new.npz >> $O/ab.log 2>&1
tail -1 $O/ab.log
python tools/bench_fit.py --profile --sizes 256,384 --fits 256 --reps 1 > $O/p_new.log 2>&1
python tools/bench_fit.py --profile --sizes 256 --fits 512 --reps 1 >> $O/p_new.log 2>&1
python tools/bench_fit.py --sizes 144,200,256 --fits 512 --reps 2 >> $O/p_new.log 2>&1
python tools/bench_fit.py --sizes 320,384,448 --fits 256 --reps 2 >> $O/p_new.log 2>&1
cat $O/p_new.log
rm -f $O/*.npz
