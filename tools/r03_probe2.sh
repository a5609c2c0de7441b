#!/usr/bin/env bash
# round 3, probe 2: workgroup-tiled products (gemm_wg) against the per-wave products of round 2 (debug bit 13): bits, tests, speed
R=$PWD; O=$R/gpurun_out/r03p2; mkdir -p $O
GAPRO_FIT_FLAGS=8192 python tools/ab_bitwise.py --out $O/old.npz > $O/ab.log 2>&1
python tools/ab_bitwise.py --out $O/new.npz >> $O/ab.log 2>&1
python tools/ab_bitwise.py --compare $O/old.npz $O/new.npz >> $O/ab.log 2>&1
cat $O/ab.log
timeout 900 python -m pytest tests/test_fit_gpu.py tests/test_svgp_kat.py -m gpu -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
python tools/bench_fit.py --sizes 144,200,256,320,384,448 --fits 512 --reps 2 > $O/new512.log 2>&1
python tools/bench_fit.py --sizes 144,200,256,320,384,448 --fits 512 --reps 2 --flags 8192 > $O/old512.log 2>&1
cat $O/new512.log $O/old512.log
rm -f $O/*.npz
