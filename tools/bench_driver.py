#!/usr/bin/env python3
"""End-to-end rate of the gen_ps driver on a synthetic on-disk dataset (disk read + host preprocessing + PCIe
upload + generation + file writes), for the "PCIe-inclusive" figure of DESIGN.md.

    python tools/bench_driver.py [--scenes 96] [--points 150000] [--batch 32] [--threads 8]
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

        def run(save, cache=None):
            cmd = [sys.executable, "-m", "gapro_amd.gen_ps", "--save_folder", save, "--data_root", data,
                   "--batch_scenes", str(args.batch), "--loader_threads", str(args.threads), "--loader_procs",
                   str(args.procs)] + (["--raw_cache", cache] if cache else [])
            t = time.time()
            out = subprocess.run(cmd, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                 capture_output=True, text=True, check=True).stdout
            lines = out.strip().splitlines()
            return time.time() - t, "\n".join(l for l in lines if l.startswith("[gen_ps]"))

        cache = os.path.join(root, "rawcache") if args.raw_cache else None
        run(os.path.join(root, "warm"), cache)  # also writes the raw cache when asked
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
