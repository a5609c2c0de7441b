#!/usr/bin/env python3
"""Print the main figures of a bench.py JSON line (diagnostic)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.1f scenes/s  ms/step %.1f  fits/step %s" % (d["value"], d["ms_per_step"], d["config"].get("gp_fits_per_step_per_gpu")))
r = d["roofline"]
print("launch: %.2f TF/s (%.1f%% of %.1f)  avg %.1f ms" % (r["achieved"], 100 * r["frac"], r["peak"], r["avg_launch_ms"]))
for k, v in d["fit_launch"]["kernels"].items():
    print("  %-8s %8.1f ms  %6.2f TFLOP  %6.2f TF/s" % (k, v["avg_ms"], v["flops"] / 1e12, v["tflops"]))
print("m hist", d.get("fit_m_hist"))
p = d["partition"]
print("partition ms", {k: round(v, 2) for k, v in p["ms"].items()}, "GB/s %.0f" % p["GB/s"], "pool only %.0f" % p["pool_only_GB/s"])
for k in ("fixed_size_line", "gen_ps_disk_inclusive", "cpu_baseline", "peak_measured"):
    if d.get(k):
        print(k, {a: (round(b, 2) if isinstance(b, float) else b) for a, b in d[k].items() if not isinstance(b, (dict, str))})
