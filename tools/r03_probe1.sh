#!/usr/bin/env bash
# round 3, probe 1: staged-kernel phase shares, and the spill A/B (the <4> build against the <2> build at the same M)
R=$PWD; O=$R/gpurun_out/r03p1; mkdir -p $O
python tools/bench_fit.py --profile --sizes 200,256,320,384,448 --fits 256 --reps 2 > $O/prof256.log 2>&1
python tools/bench_fit.py --sizes 200,256 --fits 512 --reps 2 > $O/b512_wps4.log 2>&1
python tools/bench_fit.py --sizes 200,256 --fits 512 --reps 2 --flags 64 > $O/b512_wps2.log 2>&1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc4_$c -o f --output-format csv -- python3 $R/tools/bench_fit.py --sizes 256 --fits 512 --reps 1 > $O/pmc4_$c.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc2_$c -o f --output-format csv -- python3 $R/tools/bench_fit.py --sizes 256 --fits 512 --reps 1 --flags 64 > $O/pmc2_$c.log 2>&1
done
cd $R
python tools/pmc_summary.py $O/pmc4_FETCH_SIZE $O/pmc4_WRITE_SIZE $O/pmc2_FETCH_SIZE $O/pmc2_WRITE_SIZE > $O/pmc_summary.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +8M -delete
cat $O/prof256.log $O/b512_wps4.log $O/b512_wps2.log; grep k_svgp $O/pmc_summary.txt
