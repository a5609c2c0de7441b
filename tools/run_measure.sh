mkdir -p gpurun_out/r01b
R=$PWD
timeout 600 python bench.py > gpurun_out/r01b/bench.json 2> gpurun_out/r01b/bench.err; tail -3 gpurun_out/r01b/bench.err; cat gpurun_out/r01b/bench.json
timeout 300 python bench.py --no-cpu-baseline --stage-times --steps 5 > gpurun_out/r01b/bench_stages.json 2> gpurun_out/r01b/bench_stages.err; tail -3 gpurun_out/r01b/bench_stages.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01b/stats -o bench --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r01b/stats.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r01b/pmc_fetch -o bench --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r01b/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/r01b/pmc_write -o bench --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r01b/pmc_write.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r01b/calib_fetch -o calib --output-format csv -- python3 $R/tools/pmc_calib.py > $R/gpurun_out/r01b/calib_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/r01b/calib_write -o calib --output-format csv -- python3 $R/tools/pmc_calib.py > $R/gpurun_out/r01b/calib_write.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/r01b/pmc_fetch gpurun_out/r01b/pmc_write gpurun_out/r01b/calib_fetch gpurun_out/r01b/calib_write > gpurun_out/r01b/pmc_summary.txt 2>&1
cat gpurun_out/r01b/pmc_summary.txt | grep -v "rocclr\|k_scan\|k_flags\|k_rank"
find gpurun_out/r01b -name "*kernel_stats.csv" | head; find gpurun_out/r01b -name "*kernel_trace.csv" -delete; find gpurun_out/r01b -name "*counter_collection.csv" -size +8M -delete
tail -3 gpurun_out/r01b/stats.log
