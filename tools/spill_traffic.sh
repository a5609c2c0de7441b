#!/usr/bin/env bash
# how much of the staged kernels' HBM-side traffic is register spills?  The two-per-CU build
# k_svgp_fit<4, true> (128 VGPRs: 2439 spill instructions, 1960 B of scratch per lane) and the one-per-CU build
# k_svgp_fit<2, true> (256 VGPRs: 1011 spill instructions; debug bit 6 routes every staged fit there) run the SAME
# products on the same fits; FETCH_SIZE / WRITE_SIZE per fit of both, separate --pmc passes, calibrated in the same
# call on known byte counts (tools/pmc_calib.py).  Output: gpurun_out/$TAG/spill_traffic.txt (tools/spill_traffic.py
# turns it into profiles/r06_spill_traffic.md).
TAG=${1:-spill}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "scratch|SQ_INSTS_(VMEM|FLAT|SMEM|LDS)|TCC_EA|TCP_TCC" | cut -c1-200 | head -60 > $O/avail_counters.txt
for SZ in 160 200 256; do
  for FL in 0 64; do
    for C in FETCH_SIZE WRITE_SIZE; do
      timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/m${SZ}_f${FL}_$C -o fit --output-format csv -- \
        python3 $R/tools/bench_fit.py --sizes $SZ --fits 512 --reps 1 --flags $FL > $O/m${SZ}_f${FL}_$C.log 2>&1
    done
  done
done
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/calib_fetch -o calib --output-format csv -- python3 $R/tools/pmc_calib.py > $O/calib_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/calib_write -o calib --output-format csv -- python3 $R/tools/pmc_calib.py > $O/calib_write.log 2>&1
cd $R
for SZ in 160 200 256; do
  for FL in 0 64; do
    echo "== M=$SZ flags=$FL"; grep "^M=" $O/m${SZ}_f${FL}_FETCH_SIZE.log
    python tools/pmc_summary.py $O/m${SZ}_f${FL}_FETCH_SIZE $O/m${SZ}_f${FL}_WRITE_SIZE | grep "k_svgp"
  done
done > $O/spill_traffic.txt 2>&1
echo "== calibration" >> $O/spill_traffic.txt
python tools/pmc_summary.py $O/calib_fetch $O/calib_write | grep -i "stream\|calib" >> $O/spill_traffic.txt
find $O -name "*kernel_trace.csv" -delete
cat $O/spill_traffic.txt | cut -c1-170
