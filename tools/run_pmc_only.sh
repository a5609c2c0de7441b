#!/usr/bin/env bash
# The part of tools/run_measure.sh that the bench line's `roofline.traffic` depends on, for a kernel build whose only
# change since the full pass does not alter the kernels' behaviour: bench line, rocprofv3 kernel stats, the two PMC
# passes and their calibration, the kernel build id.  Outputs under gpurun_out/$TAG; tools/make_profile_summary.py as usual.
TAG=${1:-pmc}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
python -c "from gapro_amd._lib import source_build_id; print(source_build_id())" > $O/build_id.txt
( time timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -3 $O/bench.time
LIGHT="--no-cpu-baseline --no-fixed-line --no-driver-line --no-extra-lines"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o bench --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 $LIGHT > $O/stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o bench --output-format csv -- python3 $R/bench.py --steps 5 --warmup 0 $LIGHT > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o bench --output-format csv -- python3 $R/bench.py --steps 5 --warmup 0 $LIGHT > $O/pmc_write.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/calib_fetch -o calib --output-format csv -- python3 $R/tools/pmc_calib.py > $O/calib_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/calib_write -o calib --output-format csv -- python3 $R/tools/pmc_calib.py > $O/calib_write.log 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/calib_fetch $O/calib_write > $O/pmc_summary.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +8M -delete
python tools/show_bench.py $O/bench.json | head -12
