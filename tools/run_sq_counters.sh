#!/usr/bin/env bash
# SQ / L2 counter passes over the bench's stream workload (run through gpurun; --pmc passes carry --kernel-trace only).
TAG=${1:-sq}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
LIGHT="--no-cpu-baseline --no-fixed-line --no-driver-line --steps 2 --warmup 1"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT -d $O/sq -o bench --output-format csv -- python3 $R/bench.py $LIGHT > $O/sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O/tcc -o bench --output-format csv -- python3 $R/bench.py $LIGHT > $O/tcc.log 2>&1
cd $R
python tools/pmc_summary.py $O/sq $O/tcc > $O/summary.txt 2>&1
grep "k_svgp" $O/summary.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +8M -delete
