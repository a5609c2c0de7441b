#!/usr/bin/env python3
"""What the HOST of this box can deliver to W GPU workers of the gen_ps farm -- with no GPU involved.

    python tools/host_ceiling.py [--workers 1,2,4,8] [--scenes 1024] [--points 150000] [--torch-io]

One MI355X takes ~340 scenes/s with resident inputs; eight workers on one host share its cores.  The host work per
scene is the read (scene tuple + superpoint ids), the preprocessing of read_scene (features, axis alignment), one pass
over the arrays (the copy into the pinned staging buffer), and the write of the 5-tuple.  Rounds 2-3 measured this with
an emulation of the loader-process pool and found the host capped at 270 .. 350 scenes/s by unpickling.  Round 4: the
driver itself has a GPU-less mode -- `gen_ps --devices 0,..,W-1 --dry_run` runs W real worker processes (claim queue,
loader threads, native reader / writer, result files; all-zero stand-in outputs) -- and this tool times exactly that,
so the figure is the product's own host code, not a model of it.  Round 5: those workers run the library's batch
feeder in its host-only mode (the same C++ loader / writer threads as with a GPU, pageable instead of pinned staging).

It deliberately has no GPU in it: on the 1-GPU bench box eight real workers would time-share ONE device.
Prints per W: delivered scenes/s (all workers together; slowest worker's clock, start-up apart) and the projected
farm rate min(W x gpu, host).
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(workers, data, out_root, native=True, tag="", threads=-1):
    save = os.path.join(out_root, "w%d%s" % (workers, tag))
    cmd = [sys.executable, "-m", "gapro_amd.gen_ps", "--save_folder", save, "--data_root", data, "--batch_scenes", "32",
           "--devices", ",".join(str(d) for d in range(workers)), "--raw_cache", "none", "--dry_run", "--loader_threads", str(threads)]
    env = dict(os.environ, GAPRO_NATIVE_PTH="1" if native else "0")
    t = time.time()
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, env=env)
    wall = time.time() - t
    done = [(int(a), float(b)) for a, b in re.findall(r"(\d+) scenes written, \d+ skipped/failed, ([\d.]+) s", r.stdout)]
    start = [float(a) for a in re.findall(r"start-up ([\d.]+) s", r.stdout)]
    thr = re.findall(r"(\d+) loader threads", r.stdout)
    n = sum(d for d, _ in done)
    slow = max((t for _, t in done), default=0.0)
    shutil.rmtree(save + ".DRY_RUN", ignore_errors=True)
    shutil.rmtree(save, ignore_errors=True)
    row = {"workers": workers, "loader_threads_per_worker": int(thr[0]) if thr else None, "source": "pth",
           "file_io": "native feeder (gapro_feed_*)",
           "scenes_per_s": round(n / slow, 1) if slow > 0 else 0.0, "slowest_worker_s": round(slow, 2),
           "startup_s": round(max(start, default=0.0), 2), "wall_s": round(wall, 2), "scenes": n}
    if r.returncode != 0:
        row["error"] = r.stderr[-400:]
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", default="1,2,4,8")
    ap.add_argument("--scenes", type=int, default=1024)
    ap.add_argument("--points", type=int, default=150000)
    ap.add_argument("--unique", type=int, default=16, help="scenes generated")
    ap.add_argument("--distinct", type=int, default=512, help="scenes with files of their own (copies of the generated ones)")
    ap.add_argument("--gpu-rate", type=float, default=340.0, help="scenes/s one GPU takes (resident inputs)")
    ap.add_argument("--threads", type=int, default=-1, help="loader threads per worker (-1 = gen_ps's own choice)")
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    from gapro_amd.synth import make_scene, write_scannet_layout

    root = tempfile.mkdtemp(prefix="gapro_hc_")
    try:
        data = os.path.join(root, "dataset", "scannetv2")
        uniq = min(args.unique, args.scenes)
        for i in range(uniq):
            sc = make_scene(seed=i % 8, n_points=args.points, n_objects=25, with_walls_json=False,
                            scan_name="scene%04d_00" % i)
            write_scannet_layout(sc, data)
        # Further scenes are real COPIES (own inodes, own page-cache pages) up to --distinct, links beyond: with 16
        # files behind 1024 names, 64 .. 128 reader threads take and drop references on the same few thousand page
        # structs -- cache-line ping-pong across both sockets that a dataset of distinct files does not have (round 4:
        # 4 workers x 16 threads read 985 scenes/s from 16 linked files)
        link = os.symlink
        for i in range(uniq, args.scenes):
            base = i % uniq if i < args.distinct else uniq + (i - uniq) % max(1, min(args.distinct, args.scenes) - uniq)
            src, dst = "scene%04d_00" % base, "scene%04d_00" % i
            make = shutil.copyfile if i < args.distinct else link
            make(os.path.join(data, "train", src + "_inst_nostuff.pth"),
                 os.path.join(data, "train", dst + "_inst_nostuff.pth"))
            make(os.path.join(data, "superpoints", src + ".pth"), os.path.join(data, "superpoints", dst + ".pth"))
            os.makedirs(os.path.join(data, "scans_transform", dst))
            os.symlink(os.path.join(data, "scans_transform", src, src + ".txt"),
                       os.path.join(data, "scans_transform", dst, dst + ".txt"))
        measure(1, data, root, True, "warm")  # page cache, code objects
        rows = []
        for w in [int(x) for x in args.workers.split(",") if x]:
            for native in (True,):
                r = measure(w, data, root, native, "" if native else "t", args.threads)
                r["projected_farm_scenes_per_s"] = round(min(w * args.gpu_rate, r["scenes_per_s"]), 1)
                rows.append(r)
                print("workers %d x %s loader threads, %s: host delivers %7.1f scenes/s -> farm of %d GPUs: "
                      "min(%d x %.0f, host) = %.0f" % (w, r["loader_threads_per_worker"], r["file_io"], r["scenes_per_s"],
                                                       w, w, args.gpu_rate, r["projected_farm_scenes_per_s"]), flush=True)
        out = {"host": {"cpus": os.cpu_count()}, "points_per_scene": args.points, "scenes": args.scenes,
               "gpu_rate_assumed": args.gpu_rate, "rows": rows,
               "how": "gen_ps --devices 0..W-1 --dry_run: the driver's own worker processes without a GPU"}
        if args.json:
            with open(args.json, "w") as f:
                json.dump(out, f, indent=1)
        print("JSON " + json.dumps(out))
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
