#!/bin/bash
# Mid-size fits (M = 256 .. 480) on the multi-workgroup kernel instead of the staged one: per-size rates by work unit
# (GAPRO_CLUSTER_UNIT: a fit of M_p gets the next power of two >= (M_p / unit)^3 workgroups).
for M in 256 320 384 448; do
  echo "== M=$M staged:"; python tools/bench_fit.py --sizes $M --fits 512 --reps 2 2>&1 | grep "^M="
  for U in 384 256 192 160; do
    for F in 64 128; do
      echo "== M=$M cluster unit $U fits $F:"; GAPRO_CLUSTER_UNIT=$U python tools/bench_fit.py --cluster-all --sizes $M --fits $F --reps 2 2>&1 | grep "^M=" | cut -c1-120
    done
  done
done
