#!/usr/bin/env bash
# why is the FIRST bench run on a fresh box slower in wall time per step than the launch (and later runs are not)?
mkdir -p gpurun_out/p18
L="--no-cpu-baseline --no-extra-lines --no-fixed-line --no-driver-line"
for i in 1 2 3; do
  python bench.py $L > gpurun_out/p18/run$i.json 2> gpurun_out/p18/run$i.err
  python tools/show_bench.py gpurun_out/p18/run$i.json | head -2 | sed "s/^/run $i: /"
done
python bench.py $L --warmup 6 > gpurun_out/p18/run4.json 2> gpurun_out/p18/run4.err
python tools/show_bench.py gpurun_out/p18/run4.json | head -2 | sed "s/^/run 4 (warmup 6): /"
