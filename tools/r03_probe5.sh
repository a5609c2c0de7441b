#!/usr/bin/env bash
R=$PWD; O=$R/gpurun_out/r03p5; mkdir -p $O
timeout 600 python -m pytest tests/test_fit_gpu.py -m gpu -x -q -k "lane_maps" > $O/lanemap.log 2>&1; tail -3 $O/lanemap.log
GAPRO_FIT_FLAGS=8192 python tools/ab_bitwise.py --out $O/old.npz > $O/ab.log 2>&1
python tools/ab_bitwise.py --out $O/new.npz >> $O/ab.log 2>&1
python tools/ab_bitwise.py --compare $O/old.npz $O/new.npz >> $O/ab.log 2>&1
tail -2 $O/ab.log
python tools/bench_fit.py --sizes 144,200,256 --fits 512 --reps 2 > $O/new512.log 2>&1
python tools/bench_fit.py --sizes 256,320,384,448 --fits 256 --reps 2 > $O/new256.log 2>&1
python tools/bench_fit.py --sizes 256,320,384,448 --fits 256 --reps 2 --flags 8192 > $O/old256.log 2>&1
cat $O/new512.log $O/new256.log $O/old256.log
rm -f $O/*.npz
