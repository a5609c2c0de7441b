#!/usr/bin/env bash
R=$PWD; O=$R/gpurun_out/r03p11; mkdir -p $O
python tools/ab_bitwise.py --out $O/old.npz > $O/ab.log 2>&1
GAPRO_FIT_FLAGS=65536 python tools/ab_bitwise.py --out $O/new.npz >> $O/ab.log 2>&1
python tools/ab_bitwise.py --compare $O/old.npz $O/new.npz >> $O/ab.log 2>&1
tail -1 $O/ab.log
python tools/bench_fit.py --sizes 288,320,384,448,496 --fits 256 --reps 2 --flags 65536 > $O/new.log 2>&1
python tools/bench_fit.py --sizes 288,320,384,448,496 --fits 256 --reps 2 > $O/old.log 2>&1
cat $O/new.log $O/old.log
rm -f $O/*.npz
