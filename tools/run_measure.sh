#!/usr/bin/env bash
# One measurement pass on the GPU box (run through gpurun): bench line, per-stage times, rocprofv3 kernel
# stats, PMC passes (FETCH_SIZE and WRITE_SIZE separately) over the same bench command, and the PMC
# calibration on known byte counts.  Outputs under gpurun_out/$TAG; copy what is to be judged to profiles/.
TAG=${1:-meas}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cat $O/bench.json
LIGHT="--no-cpu-baseline --no-fixed-line --no-driver-line"
timeout 300 python bench.py $LIGHT --stage-times --steps 4 > $O/bench_stages.json 2> $O/bench_stages.err; tail -2 $O/bench_stages.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats -o bench --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 $LIGHT > $O/stats.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o bench --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 $LIGHT > $O/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o bench --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 $LIGHT > $O/pmc_write.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/calib_fetch -o calib --output-format csv -- python3 $R/tools/pmc_calib.py > $O/calib_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/calib_write -o calib --output-format csv -- python3 $R/tools/pmc_calib.py > $O/calib_write.log 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/calib_fetch $O/calib_write > $O/pmc_summary.txt 2>&1
grep -v "rocclr\|k_scan\|k_flags\|k_rank\|at::native" $O/pmc_summary.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +8M -delete
head -6 $O/stats/bench_kernel_stats.csv | cut -c1-200
