#!/usr/bin/env bash
# Headline A/B on ONE box: bench.py with libgapro_hip.so (new) and libgapro_hip_prev.so (a GAPRO_VARIANT=prev build of
# an earlier commit) swapped in turn, new / prev / new / prev -- the first run on a fresh box is slow in wall time.
set -u
mkdir -p gpurun_out/ab
cp gapro_amd/libgapro_hip.so /tmp/new.so
cp gapro_amd/libgapro_hip_prev.so /tmp/prev.so
L="--no-cpu-baseline --no-extra-lines --no-fixed-line --no-driver-line"
for rep in 1 2; do
  for v in new prev; do
    cp /tmp/$v.so gapro_amd/libgapro_hip.so
    python bench.py $L > gpurun_out/ab/${v}_$rep.json 2> gpurun_out/ab/${v}_$rep.err
    python tools/show_bench.py gpurun_out/ab/${v}_$rep.json | head -6 | sed "s/^/$v $rep: /"
  done
done
cp /tmp/new.so gapro_amd/libgapro_hip.so
