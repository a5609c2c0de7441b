#!/usr/bin/env bash
# phase shares (prof build): usage tools/r03_probe14.sh [sizes] [fits]
mkdir -p gpurun_out/p14; o=gpurun_out/p14/phases.txt; : > $o
python tools/bench_fit.py --sizes ${1:-160,256,320,384,448} --fits ${2:-512} --reps 2 --profile >> $o 2>&1
cat $o
