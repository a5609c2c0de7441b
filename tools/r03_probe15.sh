#!/usr/bin/env bash
mkdir -p gpurun_out/p15; o=gpurun_out/p15/timeline.txt; : > $o
python tools/fit_timeline.py --bins 20 >> $o 2>&1
for rep in 1 2; do for lib in libgapro_hip.so libgapro_hip_prev.so; do
  echo "== $lib" >> $o; python tools/fit_timeline.py --lib $lib --reps 5 >> $o 2>&1
done; done
cat $o
