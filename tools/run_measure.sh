#!/usr/bin/env bash
# One measurement pass on the GPU box (run through gpurun): bench line, per-stage times, rocprofv3 kernel stats, PMC
# passes (FETCH_SIZE and WRITE_SIZE separately) over the same bench command, the PMC calibration on known byte counts,
# and (round 3) the A/B of the staged kernel's products, the matrix-core loop benchmark and the host ceiling.
# Outputs under gpurun_out/$TAG; copy what is to be judged to profiles/ (tools/make_profile_summary.py).
TAG=${1:-meas}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
( time timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -3 $O/bench.time; tail -c 600 $O/bench.json; echo
LIGHT="--no-cpu-baseline --no-fixed-line --no-driver-line --no-extra-lines"
timeout 400 python bench.py $LIGHT --stage-times --steps 5 --warmup 1 > $O/bench_stages.json 2> $O/bench_stages.err; tail -2 $O/bench_stages.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o bench --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 $LIGHT > $O/stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o bench --output-format csv -- python3 $R/bench.py --steps 5 --warmup 0 $LIGHT > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o bench --output-format csv -- python3 $R/bench.py --steps 5 --warmup 0 $LIGHT > $O/pmc_write.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/calib_fetch -o calib --output-format csv -- python3 $R/tools/pmc_calib.py > $O/calib_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/calib_write -o calib --output-format csv -- python3 $R/tools/pmc_calib.py > $O/calib_write.log 2>&1
# staged kernel, per-wave products (default) against the workgroup-tiled ones (experiment bit 13): bytes and time
for c in FETCH_SIZE WRITE_SIZE; do for f in 0 131072; do for m in 256 384; do
timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/ab_${c}_${f}_${m} -o f --output-format csv -- python3 $R/tools/bench_fit.py --sizes $m --fits 256 --reps 1 --flags $f > $O/ab_${c}_${f}_${m}.log 2>&1
done; done; done
cd $R
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/calib_fetch $O/calib_write > $O/pmc_summary.txt 2>&1
python tools/pmc_summary.py $O/ab_* > $O/ab_pmc_summary.txt 2>&1
grep -v "rocclr\|k_scan\|k_flags\|k_rank\|at::native" $O/pmc_summary.txt | cut -c1-160
grep "k_svgp" $O/ab_pmc_summary.txt | cut -c1-160; grep -h "^M=" $O/ab_FETCH_SIZE_*.log
python tools/wgloop_peak.py > $O/wgloop_peak.txt 2>&1; grep -v amdgpu $O/wgloop_peak.txt | tail -30
python tools/host_ceiling.py --workers 1,2,4,8 --json $O/host_ceiling.json > $O/host_ceiling.txt 2>&1; grep "^workers" $O/host_ceiling.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +8M -delete
head -8 $O/stats/bench_kernel_stats.csv | cut -c1-200
