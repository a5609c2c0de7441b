#!/usr/bin/env bash
R=$PWD; O=$R/gpurun_out/r03p13; mkdir -p $O
L="--no-cpu-baseline --no-fixed-line --no-driver-line --no-extra-lines --steps 6 --warmup 2"
for mp in 0 262144 0 262144; do
GAPRO_FIT_FLAGS=$mp timeout 600 python bench.py $L > $O/bench_$mp.json 2> $O/bench_$mp.err
python - <<PY
import json
d=json.loads(open("$O/bench_$mp.json").read().strip().splitlines()[-1])
k=d["fit_launch"]["kernels"]
print("flags $mp: %.1f scenes/s  launch %.1f ms  frac %.4f | staged %.0f strip %.0f small %.0f cluster %.0f ms" % (d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], k["staged"]["avg_ms"], k["strip"]["avg_ms"], k["small"]["avg_ms"], k["cluster"]["avg_ms"]))
PY
done
