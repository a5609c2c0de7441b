#!/bin/bash
# A/B of fit routing flags on the headline workload (short bench runs, alternating)
for rep in 1 2; do
for fl in 0 8192 24576; do
  GAPRO_FIT_FLAGS=$fl python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fixed-line --no-extra-lines --no-driver-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flags $fl rep $rep: %.1f scenes/s  launch %.1f ms  frac %.4f' % (d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
done; done
