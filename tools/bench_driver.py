#!/usr/bin/env python3
"""End-to-end rate of the gen_ps driver on a synthetic on-disk dataset (disk read + host preprocessing + PCIe
upload + generation + file writes), for the "PCIe-inclusive" figure of DESIGN.md.

    python tools/bench_driver.py [--scenes 96] [--points 150000] [--batch 32] [--threads 8]
    python tools/bench_driver.py --workers 1,2,4,8 --host-only --raw-cache     # host ceiling of the multi-GPU farm

`--workers W,...`: for every W, gen_ps --devices 0,0,..,0 (W workers on GPU 0: what is measured is the HOST side of
a W-GPU farm on one box; with --host-only the generation itself is replaced by zero outputs so that the one shared GPU
is not what limits the run).  Prints one "farm" line per W and source (.pth files / raw cache): delivered scenes/s =
scenes / slowest worker's own clock (start-up excluded), start-up seconds apart.
"""
import argparse
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=512)
    ap.add_argument("--points", type=int, default=150000)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--unique", type=int, default=32, help="distinct scenes on disk; the rest are links")
    ap.add_argument("--procs", type=int, default=0, help="loader processes (0 = threads)")
    ap.add_argument("--raw-cache", action="store_true",
                    help="also time a run over the opt-in raw scene cache (gen_ps --raw_cache), written by the warm run")
    ap.add_argument("--workers", default="", help="comma-separated worker counts: farm runs with --devices 0,0,...")
    ap.add_argument("--host-only", action="store_true", help="GAPRO_DRIVER_HOST_ONLY=1: no generation (host ceiling)")
    args = ap.parse_args()
    from gapro_amd import gen_ps
    from gapro_amd.synth import make_scene, write_scannet_layout

    root = tempfile.mkdtemp(prefix="gapro_ds_")
    try:
        data = os.path.join(root, "dataset", "scannetv2")
        t0 = time.time()
        uniq = min(args.scenes, args.unique)
        for i in range(uniq):
            sc = make_scene(seed=i % 8, n_points=args.points, n_objects=25, with_walls_json=False,
                            scan_name="scene%04d_00" % i)
            write_scannet_layout(sc, data)
        for i in range(uniq, args.scenes):  # further scenes: links to the unique ones (page-cache resident anyway)
            src, dst = "scene%04d_00" % (i % uniq), "scene%04d_00" % i
            os.symlink(os.path.join(data, "train", src + "_inst_nostuff.pth"),
                       os.path.join(data, "train", dst + "_inst_nostuff.pth"))
            os.symlink(os.path.join(data, "superpoints", src + ".pth"), os.path.join(data, "superpoints", dst + ".pth"))
            os.makedirs(os.path.join(data, "scans_transform", dst))
            os.symlink(os.path.join(data, "scans_transform", src, src + ".txt"),
                       os.path.join(data, "scans_transform", dst, dst + ".txt"))
        print("dataset of %d scenes written in %.1f s" % (args.scenes, time.time() - t0), flush=True)
        # each run is a child process (the loader pool must start before the process touches the GPU); a first
        # run warms the page cache and the code-object cache, the second one is timed on fresh outputs
        import subprocess

        def run(save, cache=None, workers=1, procs=None, host_only=False):
            cmd = [sys.executable, "-m", "gapro_amd.gen_ps", "--save_folder", save, "--data_root", data,
                   "--batch_scenes", str(args.batch), "--loader_threads", str(args.threads), "--loader_procs",
                   str(args.procs if procs is None else procs), "--devices", ",".join(["0"] * workers),
                   "--raw_cache", cache if cache else "none"]
            env = dict(os.environ)
            if host_only:
                env["GAPRO_DRIVER_HOST_ONLY"] = "1"
            t = time.time()
            out = subprocess.run(cmd, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                 capture_output=True, text=True, check=True, env=env).stdout
            lines = out.strip().splitlines()
            return time.time() - t, "\n".join(l for l in lines if l.startswith("[gen_ps]"))

        def farm(workers, cache, tag):
            import re

            wall, txt = run(os.path.join(root, "farm_%s_%d" % (tag, workers)), cache, workers, procs=-1,
                            host_only=args.host_only)
            done = [(int(a), float(b)) for a, b in re.findall(r"(\d+) scenes written, \d+ skipped/failed, ([\d.]+) s", txt)]
            start = [float(a) for a in re.findall(r"start-up ([\d.]+) s", txt)]
            loaders = re.findall(r"(\d+) loader processes", txt)
            n = sum(d for d, _ in done)
            slow = max((t for _, t in done), default=0.0)
            print("farm %s workers %d: %d scenes, %.1f scenes/s delivered (slowest worker %.1f s), start-up %.1f s, "
                  "wall %.1f s, loaders per worker %s%s" % (tag, workers, n, n / slow if slow > 0 else 0.0, slow,
                                                           max(start, default=0.0), wall, loaders[0] if loaders else "?",
                                                           ", host-only" if args.host_only else ""), flush=True)

        cache = os.path.join(root, "rawcache") if args.raw_cache else None
        run(os.path.join(root, "warm"), cache)  # also writes the raw cache when asked
        if args.workers:
            for w in [int(x) for x in args.workers.split(",") if x]:
                farm(w, None, "pth")
                if cache:
                    farm(w, cache, "raw")
            return
        dt, line = run(os.path.join(root, "labels"))
        print(line)
        if cache:
            dtc, linec = run(os.path.join(root, "labels_cached"), cache)
            print("raw cache: " + linec)
        print("driver process wall time %.2f s (includes interpreter start and library load)" % dt)
        print("loaders: %s" % ("%d processes" % args.procs if args.procs else "%d threads" % args.threads))
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
