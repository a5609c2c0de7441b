#!/usr/bin/env bash
R=$PWD; O=$R/gpurun_out/r03p12; mkdir -p $O
L="--no-cpu-baseline --no-fixed-line --no-driver-line --no-extra-lines --steps 6 --warmup 2"
for f in 0 131072 8192 0 131072 8192; do
GAPRO_FIT_FLAGS=$f timeout 600 python bench.py $L > $O/bench_$f.json 2> $O/bench_$f.err
python - <<PY
import json
d=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
print("flags $f: value %.1f scenes/s  ms/step %.1f  roofline %s" % (d["value"], d["ms_per_step"], {k:d["roofline"].get(k) for k in ("achieved","frac","avg_launch_ms")}))
PY
done
