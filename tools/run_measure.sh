#!/usr/bin/env bash
# One measurement pass on the GPU box (run through gpurun): the bench line (with the gen_ps farm, the host ceiling and
# the CPU baseline), per-stage times, rocprofv3 kernel stats, PMC passes (FETCH_SIZE and WRITE_SIZE separately) over the
# same bench command with the calibration on known byte counts, and (round 4) the per-phase tables of the staged kernel
# with matrix-pipe busy shares, the per-phase profiles of the cluster / strip / small kernels and the CU-time split of one
# launch of the train-split mix per kernel (diagnostic builds: GAPRO_BUILD_PROFILE=1 and the profsplit variant).
# Outputs under gpurun_out/$TAG; copy what is to be judged to profiles/ (tools/make_profile_summary.py).
# Needs the two diagnostic libraries in the tree (they are not kept there between passes -- they ride along in every push
# to the GPU box otherwise):
#   GAPRO_BUILD_PROFILE=1 GAPRO_VARIANT=profsplit GAPRO_VARIANT_FLAGS="-DGAPRO_PROFILE -DGAPRO_PROFILE_SPLIT" bash gapro_amd/csrc/build.sh
TAG=${1:-meas}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
python -c "from gapro_amd._lib import source_build_id; print(source_build_id())" > $O/build_id.txt
( time timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -3 $O/bench.time; tail -c 600 $O/bench.json; echo
LIGHT="--no-cpu-baseline --no-fixed-line --no-driver-line --no-extra-lines"
GAPRO_DUMP_FIT_M=$O/train_split_fit_m.npy timeout 400 python bench.py $LIGHT --stage-times --steps 5 --warmup 1 > $O/bench_stages.json 2> $O/bench_stages.err; tail -2 $O/bench_stages.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o bench --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 $LIGHT > $O/stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o bench --output-format csv -- python3 $R/bench.py --steps 5 --warmup 0 $LIGHT > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o bench --output-format csv -- python3 $R/bench.py --steps 5 --warmup 0 $LIGHT > $O/pmc_write.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/calib_fetch -o calib --output-format csv -- python3 $R/tools/pmc_calib.py > $O/calib_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/calib_write -o calib --output-format csv -- python3 $R/tools/pmc_calib.py > $O/calib_write.log 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/calib_fetch $O/calib_write > $O/pmc_summary.txt 2>&1
grep -v "rocclr\|k_scan\|k_flags\|k_rank\|at::native" $O/pmc_summary.txt | cut -c1-160
# per-phase tables (staged kernel) and profiles (cluster / strip / small kernels), CU-time split of a launch
python tools/phase_table.py --sizes 160,256,320,384,448 --fits 512 --lib libgapro_hip_profsplit.so > $O/phase_tables.md 2>&1
python tools/phase_table.py --sizes 256 --fits 256 --lib libgapro_hip_profsplit.so >> $O/phase_tables.md 2>&1
python tools/bench_fit.py --profile --sizes 544,640,768,1024 --fits 16 --reps 1 > $O/cluster_prof.log 2>&1
python tools/bench_fit.py --profile --sizes 64,80,96,128 --fits 512 --reps 1 > $O/strip_prof.log 2>&1
python tools/bench_fit.py --profile --sizes 16,32,48 --fits 2048 --reps 1 > $O/wave_phases.txt 2>&1
python tools/fit_timeline.py --fit-m $O/train_split_fit_m.npy > $O/fit_timeline.txt 2>&1; grep -v amdgpu $O/fit_timeline.txt | head -12
python tools/bench_fit.py --sizes 16,32,48 --fits 4096 --reps 2 > $O/fit_sizes.log 2>&1
python tools/bench_fit.py --sizes 64,80,96,128,160,200,256,320,384,448 --fits 512 --reps 2 >> $O/fit_sizes.log 2>&1
# round 6: the deep-feature workflow's small fits (wave-per-fit at M_p <= 32) and their neighbours on the workgroup kernels
python tools/bench_fit.py --d 32 --sizes 16,32 --fits 4096 --reps 2 >> $O/fit_sizes.log 2>&1
python tools/bench_fit.py --d 32 --sizes 48,64,128 --fits 512 --reps 2 >> $O/fit_sizes.log 2>&1; grep "^M=" $O/fit_sizes.log
# round 6: the reproducibility probe and the conditioning figure over the S3DIS-shaped scene's 66 fits
python tools/cond_survey.py > $O/cond_survey.txt 2>&1
# round 6: HBM-side traffic of the two-per-CU staged build on its own (profiles/r06_spill_traffic.md), after the step function
cd /tmp
for SZ in 160 200 256; do
  for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VMEM_WR; do
    timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/staged_m${SZ}_$C -o fit --output-format csv -- python3 $R/tools/bench_fit.py --sizes $SZ --fits 512 --reps 1 > $O/staged_m${SZ}_$C.log 2>&1
  done
done
cd $R
for SZ in 160 200 256; do echo "== M=$SZ"; grep "^M=" $O/staged_m${SZ}_FETCH_SIZE.log; python tools/pmc_summary.py $O/staged_m${SZ}_FETCH_SIZE $O/staged_m${SZ}_WRITE_SIZE $O/staged_m${SZ}_SQ_INSTS_VMEM_WR | grep "k_svgp"; done > $O/staged_traffic.txt 2>&1
python tools/mfma_peak.py > $O/wgloop_peak.txt 2>&1
python tools/host_ceiling.py --workers 1,2,4,8 --json $O/host_ceiling.json > $O/host_ceiling.txt 2>&1; grep "^workers" $O/host_ceiling.txt
python tools/host_probe.py > $O/host_probe.txt 2>&1
# SQ counters of every fit kernel of the final build: matrix-pipe busy, issue stalls, LDS conflicts
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT -d $O/sq -o bench --output-format csv -- python3 $R/bench.py $LIGHT --steps 2 --warmup 1 > $O/sq.log 2>&1
cd $R
python tools/pmc_summary.py $O/sq 2>&1 | grep "k_svgp" > $O/sq_counters.txt; cut -c1-150 $O/sq_counters.txt | head -60
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +8M -delete
head -8 $O/stats/bench_kernel_stats.csv | cut -c1-200
