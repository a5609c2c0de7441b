#!/usr/bin/env bash
# fit launches serialised by an event (default) against the old behaviour (GAPRO_OVERLAP_FITS=1), the driver's arguments
mkdir -p gpurun_out/p20
L="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-lines --no-fixed-line --no-driver-line"
for rep in 1 2; do
  python3 bench.py $L > gpurun_out/p20/ser$rep.json 2> gpurun_out/p20/ser$rep.err
  python tools/show_bench.py gpurun_out/p20/ser$rep.json | head -6 | sed "s/^/serial $rep: /"
  GAPRO_OVERLAP_FITS=1 python3 bench.py $L > gpurun_out/p20/ovl$rep.json 2> gpurun_out/p20/ovl$rep.err
  python tools/show_bench.py gpurun_out/p20/ovl$rep.json | head -6 | sed "s/^/overlap $rep: /"
done
