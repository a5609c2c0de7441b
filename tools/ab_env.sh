#!/bin/bash
# A/B of environment settings on the headline workload (short bench runs, alternating): tools/ab_env.sh "A=1" "B=2 C=3" ...
for rep in 1 2; do
for setting in "$@"; do
  env $setting python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fixed-line --no-extra-lines --no-driver-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['fit_launch']['kernels']; print('[$setting] rep $rep: %.1f scenes/s  launch %.1f ms  frac %.4f  (cluster %.0f staged %.0f strip %.0f small %.0f wave %.0f ms)' % (d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], k['cluster']['avg_ms'], k['staged']['avg_ms'], k['strip']['avg_ms'], k['small']['avg_ms'], k['wave']['avg_ms']))"
done; done
