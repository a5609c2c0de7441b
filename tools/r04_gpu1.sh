set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
df -h /tmp /dev/shm . > gpurun_out/r04a/df.txt 2>&1
nproc >> gpurun_out/r04a/df.txt; free -g >> gpurun_out/r04a/df.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04a/pytest.log
tail -5 gpurun_out/r04a/pytest.log
timeout 600 python tools/bench_driver.py --scenes 1536 --unique 24 --procs -1 --threads -1 --batch 32 > gpurun_out/r04a/driver_native.log 2>&1
tail -6 gpurun_out/r04a/driver_native.log
GAPRO_NATIVE_PTH=0 timeout 600 python tools/bench_driver.py --scenes 1536 --unique 24 --procs 16 --threads 4 --batch 32 > gpurun_out/r04a/driver_torch.log 2>&1
tail -6 gpurun_out/r04a/driver_torch.log
for t in 4 8 24; do timeout 600 python tools/bench_driver.py --scenes 1536 --unique 24 --procs 0 --threads $t --batch 32 > gpurun_out/r04a/driver_native_t$t.log 2>&1; tail -4 gpurun_out/r04a/driver_native_t$t.log; done
