#!/bin/bash
# cluster size policy sweep on the bench stream
LIGHT="--no-cpu-baseline --no-fixed-line --no-driver-line --steps 4 --warmup 1"
mkdir -p gpurun_out/expcl
run() { tag=$1; shift; env "$@" timeout 300 python bench.py $LIGHT > gpurun_out/expcl/$tag.json 2> gpurun_out/expcl/$tag.err; python - <<PY
import json
b=json.load(open("gpurun_out/expcl/$tag.json"))
k=b["fit_launch"]["kernels"]
print("$tag", "value %.1f"%b["value"], "launch %.0f ms"%b["fit_launch"]["avg_ms_first_start_to_last_end"], {n:round(v.get("ms",0)) for n,v in k.items()})
PY
}
run base GAPRO_DUMP_FIT_M=gpurun_out/expcl/fit_m.npy
run u384ceil GAPRO_CLUSTER_ROUND=ceil
run u448ceil GAPRO_CLUSTER_ROUND=ceil GAPRO_CLUSTER_UNIT=448
run u512ceil GAPRO_CLUSTER_ROUND=ceil GAPRO_CLUSTER_UNIT=512
run u512pow2 GAPRO_CLUSTER_UNIT=512
run u320ceil GAPRO_CLUSTER_ROUND=ceil GAPRO_CLUSTER_UNIT=320
run u448ceil_min512 GAPRO_CLUSTER_ROUND=ceil GAPRO_CLUSTER_UNIT=448 GAPRO_CLUSTER_MIN_MP=512
run u640ceil GAPRO_CLUSTER_ROUND=ceil GAPRO_CLUSTER_UNIT=640
