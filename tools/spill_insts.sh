#!/usr/bin/env bash
# Companion of tools/spill_traffic.sh: VMEM instruction counts of the two staged builds (SQ_INSTS_VMEM_WR / _RD count
# global AND scratch accesses; a workspace store moves 8 B per lane, a spill store 4 B per lane), so that
# WRITE bytes = 512 N_global + 256 N_spill and SQ_INSTS_VMEM_WR = N_global + N_spill separate the two.
TAG=${1:-spill_insts}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for SZ in 160 200 256; do
  for FL in 0 64; do
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_INSTS_VALU SQ_WAVES -d $O/m${SZ}_f${FL} -o fit --output-format csv -- \
      python3 $R/tools/bench_fit.py --sizes $SZ --fits 512 --reps 1 --flags $FL > $O/m${SZ}_f${FL}.log 2>&1
  done
done
cd $R
for SZ in 160 200 256; do
  for FL in 0 64; do
    echo "== M=$SZ flags=$FL"; python tools/pmc_summary.py $O/m${SZ}_f${FL} | grep "k_svgp"
  done
done > $O/spill_insts.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
cut -c1-170 $O/spill_insts.txt
