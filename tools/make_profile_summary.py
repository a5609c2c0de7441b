#!/usr/bin/env python3
"""Turn one tools/run_measure.sh output directory into the committed profile files:

    python tools/make_profile_summary.py gpurun_out/<tag> <tag> [--scenes-per-step 64 --points 150000 --feat-dim 6]

writes profiles/<tag>_bench.json, <tag>_kernel_stats.csv, <tag>_pmc_summary.txt, <tag>_pmc_traffic.json (HBM bytes
per launch of the fit kernels, corrected with the calibration run of the same pass) and <tag>_summary.md.
"""
import argparse
import csv
import json
import os
import re
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(path):
    acc = defaultdict(list)
    if not os.path.exists(path):
        return acc
    with open(path) as fh:
        for row in csv.DictReader(fh):
            acc[(row["Kernel_Name"], row["Counter_Name"])].append(float(row["Counter_Value"]))
    return acc


def mean_of(acc, key_substr, counter):
    """Mean counter value per launch of the kernels matching key_substr; kernels with different names that match
    (k_svgp_fit<2> and k_svgp_fit<4>: the two staged launches of one fit batch) are summed, each at its own mean."""
    per = {k: vs for (k, c), vs in acc.items() if key_substr in k and c == counter and vs}
    if not per:
        return None, 0
    return sum(sum(vs) / len(vs) for vs in per.values()), max(len(vs) for vs in per.values())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("tag")
    ap.add_argument("--scenes-per-step", type=int, default=256)
    ap.add_argument("--points", type=int, default=150000)
    ap.add_argument("--feat-dim", type=int, default=6)
    ap.add_argument("--workload", default="stream")
    ap.add_argument("--distinct", type=int, default=64)
    ap.add_argument("--build-id", default="", help="kernel build id of the pass when the box did not record it")
    a = ap.parse_args()
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    src = a.src
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(out, a.tag + "_bench.json"))
    shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"), os.path.join(out, a.tag + "_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "pmc_summary.txt"), os.path.join(out, a.tag + "_pmc_summary.txt"))
    with open(os.path.join(src, "bench.json")) as fh:
        bench = json.loads(fh.read().strip().splitlines()[-1])
    # round 4: per-phase tables with matrix-pipe busy shares, per-kernel CU-time split of one launch, per-phase profiles
    # of the cluster / strip / small kernels, per-size rates, the host ceiling
    for name, dst in (("phase_tables.md", "_phase_tables.md"), ("fit_timeline.txt", "_fit_timeline.txt"),
                      ("cluster_prof.log", "_cluster_phases.txt"), ("strip_prof.log", "_strip_phases.txt"),
                      ("fit_sizes.log", "_fit_sizes.txt"), ("host_ceiling.json", "_host_ceiling.json"),
                      ("wgloop_peak.txt", "_wgloop_peak.txt"), ("kernel_resources.txt", "_kernel_resources.txt"),
                      ("sq_counters.txt", "_sq_counters.txt"), ("wave_phases.txt", "_wave_phases.txt"),
                      ("cond_survey.txt", "_cond_survey.txt"), ("staged_traffic.txt", "_staged_traffic.txt")):
        pth = os.path.join(src, name)
        if os.path.exists(pth):
            with open(pth) as fh:
                txt = "".join(ln for ln in fh if "amdgpu.ids" not in ln)
            with open(os.path.join(out, a.tag + dst), "w") as fh:
                fh.write(txt)

    # calibration: known byte counts -> bytes per counter unit
    n_bytes = 8 * (1 << 26)
    cf = counters(os.path.join(src, "calib_fetch", "calib_counter_collection.csv"))
    cw = counters(os.path.join(src, "calib_write", "calib_counter_collection.csv"))
    f_mean, f_n = mean_of(cf, "k_stream_calib", "FETCH_SIZE")   # every launch reads n_bytes
    w_sum = sum(v for (k, c), vs in cw.items() if "k_stream_calib" in k and c == "WRITE_SIZE" for v in vs)
    w_launches = sum(len(vs) for (k, c), vs in cw.items() if "k_stream_calib" in k and c == "WRITE_SIZE")
    fetch_unit = n_bytes / f_mean if f_mean else None            # bytes per FETCH_SIZE unit in this access pattern
    write_unit = n_bytes * (w_launches / 2) / w_sum if w_sum else None  # half of the launches (mode 1) write n_bytes
    pf = counters(os.path.join(src, "pmc_fetch", "bench_counter_collection.csv"))
    pw = counters(os.path.join(src, "pmc_write", "bench_counter_collection.csv"))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from gapro_amd._lib import source_build_id

    # the kernels these counters belong to: bench.py attaches the figure only to lines of the same build
    bid_file = os.path.join(src, "build_id.txt")  # written on the box by tools/run_measure.sh
    traffic = {"build": (open(bid_file).read().strip() if os.path.exists(bid_file) else None) or a.build_id
                        or source_build_id(),
               "workload": {"name": a.workload, "scenes_per_step": a.scenes_per_step, "points": a.points,
                            "feat_dim": a.feat_dim, "distinct": a.distinct},
               "calibration": {"kernel": "k_stream_calib (one double per lane, grid-stride, 512 MiB per pass)",
                               "bytes_per_FETCH_SIZE_unit": fetch_unit, "bytes_per_WRITE_SIZE_unit": write_unit,
                               "note": "guide: FETCH_SIZE is in KiB and reports half of a coalesced streaming read "
                                       "on gfx950 (x2048 B per unit expected); WRITE_SIZE is KiB, exact (x1024)"}}
    for key, sub in (("cluster_kernel", "k_svgp_fit_cluster"), ("wave_kernel", "k_svgp_fit_wave"), ("strip_kernel", "k_svgp_fit_strip<"),
                     ("small_strip_kernel", "k_svgp_fit_strip256<"), ("staged_kernel", "k_svgp_fit<"),
                     ("generic_kernel", "k_svgp_fit_large"), ("pool_kernel", "k_pool("), ("stats_kernel", "k_stats("),
                     ("broadcast_kernel", "k_broadcast(")):
        fm, fn = mean_of(pf, sub, "FETCH_SIZE")
        wm, wn = mean_of(pw, sub, "WRITE_SIZE")
        if fm is None or wm is None or not fetch_unit or not write_unit:
            continue
        traffic[key] = {"launches_profiled": fn, "FETCH_SIZE_mean": fm, "WRITE_SIZE_mean": wm,
                        "read_bytes_per_launch": fm * fetch_unit, "written_bytes_per_launch": wm * write_unit,
                        "hbm_bytes_per_launch": fm * fetch_unit + wm * write_unit}
    fit_keys = [k for k in ("cluster_kernel", "wave_kernel", "strip_kernel", "small_strip_kernel", "staged_kernel", "generic_kernel")
                if k in traffic]
    if fit_keys:  # one gapro_svgp_fit_batch launch = its fit kernels side by side
        traffic["fit_launch"] = {
            "kernels": fit_keys,
            "read_bytes_per_launch": sum(traffic[k]["read_bytes_per_launch"] for k in fit_keys),
            "written_bytes_per_launch": sum(traffic[k]["written_bytes_per_launch"] for k in fit_keys),
            "hbm_bytes_per_launch": sum(traffic[k]["hbm_bytes_per_launch"] for k in fit_keys)}
    with open(os.path.join(out, a.tag + "_pmc_traffic.json"), "w") as fh:
        json.dump(traffic, fh, indent=1)

    rows = []
    with open(os.path.join(src, "stats", "bench_kernel_stats.csv")) as fh:
        for r in csv.DictReader(fh):
            name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"]).split("(")[0]
            rows.append((name, int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6,
                         float(r["Percentage"])))
    rl = bench["roofline"]
    fl = bench.get("fit_launch", {})
    md = ["# rocprofv3 summary, %s" % a.tag, "",
          "Commands (on the MI355X box, see tools/run_measure.sh):", "",
          "    python bench.py                                   -> %s_bench.json" % a.tag,
          "    L=\"--no-cpu-baseline --no-fixed-line --no-driver-line --no-extra-lines\"",
          "    rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1 $L   -> %s_kernel_stats.csv" % a.tag,
          "    rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 bench.py --steps 5 --warmup 0 $L",
          "    rocprofv3 --kernel-trace --pmc WRITE_SIZE -- (same)            -> %s_pmc_summary.txt, %s_pmc_traffic.json" % (a.tag, a.tag),
          "    rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 tools/pmc_calib.py   (calibration, same file)", "",
          "Workload: %s" % bench["config"]["workload"], "",
          "| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
    for name, calls, avg, tot, pct in rows[:14]:
        md.append("| %s | %d | %.1f | %.2f | %.2f |" % (name, calls, avg, tot, pct))
    md += ["",
           "bench.py (un-profiled run): %.1f scenes/s, %.1f ms per step of %d scenes; fit launch (%s): first kernel "
           "start -> last kernel end %.2f ms by HIP events on the kernels' streams, %.3g algorithmic FLOP per launch -> "
           "%.2f TFLOP/s = %.1f %% of the %.1f TFLOP/s FP64 MFMA datasheet peak (%.1f %% of the %.1f TFLOP/s that "
           "v_mfma_f64_16x16x4 sustains on this box in a loop shaped like the kernels' products, fragments re-read from "
           "LDS: tools/mfma_peak.py, key f64_16x16x4_lds_loop; the register-constant chains of rounds 1-2 read %.1f)."
           % (bench["value"], bench["ms_per_step"], bench["config"]["scenes_per_step_per_gpu"],
              ", ".join("%s %.0f ms" % (k, v["avg_ms"]) for k, v in fl.get("kernels", {}).items() if v["flops"] > 0),
              rl["avg_launch_ms"], rl["flops_per_launch"], rl["achieved"], 100 * rl["frac"], rl["peak"],
              100 * rl["achieved"] / max((bench.get("peak_measured") or {}).get("f64_16x16x4_lds_loop", rl["peak"]), 1e-9),
              (bench.get("peak_measured") or {}).get("f64_16x16x4_lds_loop", float("nan")),
              (bench.get("peak_measured") or {}).get("f64_16x16x4", float("nan"))), ""]
    if "fit_launch" in traffic:
        t = traffic["fit_launch"]
        ms = rl["avg_launch_ms"]
        md += ["HBM-side traffic of one fit launch (PMC, separate FETCH_SIZE / WRITE_SIZE passes, corrected by the "
               "calibration pass of the same run: %.0f B per FETCH_SIZE unit, %.0f B per WRITE_SIZE unit): %.1f GB read + "
               "%.1f GB written = %.2f TB/s over the %.1f ms launch (HBM peak 8 TB/s, ~6.3 achievable); algorithmic "
               "FLOP / HBM byte = %.2f.  Per kernel: %s."
               % (fetch_unit, write_unit, t["read_bytes_per_launch"] / 1e9, t["written_bytes_per_launch"] / 1e9,
                  t["hbm_bytes_per_launch"] / (ms * 1e-3) / 1e12, ms, rl["flops_per_launch"] / t["hbm_bytes_per_launch"],
                  "; ".join("%s %.1f GB" % (k, traffic[k]["hbm_bytes_per_launch"] / 1e9) for k in t["kernels"])), ""]
    if "pool_kernel" in traffic:
        pts = bench["config"].get("points_per_step_per_gpu", 0)
        md += ["Partition: k_pool moves %.2f GB per launch by the counters for %.2f GB algorithmic (68 B / point)."
               % (traffic["pool_kernel"]["hbm_bytes_per_launch"] / 1e9, 68.0 * pts / 1e9), ""]
    with open(os.path.join(out, a.tag + "_summary.md"), "w") as fh:
        fh.write("\n".join(md))
    print("\n".join(md))


if __name__ == "__main__":
    main()
