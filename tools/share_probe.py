#!/usr/bin/env python3
"""Where does the wall clock of a SHORT gen_ps job go?  Writes --scenes train-split-shaped scenes as files (once), then
runs `python -m gapro_amd.gen_ps --devices 0` over them --reps times and prints, per run: wall clock of the command,
the worker's own start-up / clock lines, and what is left (interpreter start + exit).  Run on the GPU box.
    python tools/share_probe.py [--scenes 151] [--reps 3] [--exit-fast]
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _write(task):
    import bench
    seed, root = task
    bench.build_scene_inputs(seed, 150000, 6, "stream", (root, "scene%04d_00" % seed))
    return seed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=151)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--env", default="", help="extra environment, K=V,K=V")
    args = ap.parse_args()
    import concurrent.futures as cf
    td = tempfile.mkdtemp(prefix="gapro_share_")
    root = os.path.join(td, "dataset", "scannetv2")
    with cf.ProcessPoolExecutor(max_workers=min(32, os.cpu_count() or 4)) as ex:
        list(ex.map(_write, [(8 * s, root) for s in range(args.scenes)]))
    env = dict(os.environ, GAPRO_DRIVER_TIMES="1")
    for kv in args.env.split(","):
        if "=" in kv:
            k, v = kv.split("=", 1)
            env[k] = v
    try:
        for r in range(args.reps):
            save = os.path.join(td, "labels%d" % r)
            t = time.time()
            p = subprocess.run([sys.executable, "-m", "gapro_amd.gen_ps", "--save_folder", save, "--data_root", root],
                               cwd=ROOT, capture_output=True, text=True, env=dict(env, GAPRO_T0=repr(t)))
            wall = time.time() - t
            txt = p.stdout
            m1 = re.search(r"(\d+) scenes written, (\d+) skipped/failed, ([\d.]+) s", txt)
            m2 = re.search(r"start-up ([\d.]+) s .*first batch out after ([\d.]+) s", txt)
            m3 = re.search(r"teardown after the last file: ([\d.]+) s", txt)
            worker = float(m1.group(3)) if m1 else float("nan")
            start = float(m2.group(1)) if m2 else float("nan")
            print("run %d: rc %d  wall %.2f s = start-up %.2f + worker clock %.2f + teardown %.2f + rest %.2f  (first batch "
                  "after %.2f s; %s scenes)" % (r, p.returncode, wall, start, worker, float(m3.group(1)) if m3 else 0.0,
                                               wall - start - worker - (float(m3.group(1)) if m3 else 0.0),
                                               float(m2.group(2)) if m2 else float("nan"), m1.group(1) if m1 else "?"))
            for ln in txt.splitlines():
                if "scenes of a batch taken" in ln or "D launched" in ln or "finish" in ln and "batch" in ln or "since the parent" in ln:
                    print("      " + ln.strip())
    finally:
        shutil.rmtree(td, ignore_errors=True)


if __name__ == "__main__":
    main()
