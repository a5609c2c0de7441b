#!/bin/bash
# tools/ab_env.sh with the default number of timed steps (16): A/Bs of the pipeline ORDER need the steady state to dominate
for rep in 1 2; do
for setting in "$@"; do
  env $setting python bench.py --no-cpu-baseline --no-fixed-line --no-extra-lines --no-driver-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$setting] rep $rep: %.1f scenes/s  ms/step %.1f  launch %.1f ms  (step - launch %.1f ms)  frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['ms_per_step'] - d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
done; done
