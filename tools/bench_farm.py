#!/usr/bin/env python3
"""The product path at farm scale: `gen_ps --devices d0,d1,..` over an on-disk dataset of ScanNet-layout files.

    python tools/bench_farm.py --data_root <dir>/dataset/scannetv2 --devices 0[,1,..] [--passes 2] [--batch 32]

What BASELINE.json's north_star names -- "full ScanNetV2-train pseudo-label generation at 8xMI355X" -- is this command:
files in (`torch.save`d scene tuples + superpoint ids + alignment + plane quads), files out (5-tuples), one worker
process per GPU sharing the scene list through the claim queue, no collective.  bench.py writes the dataset (its 1201
log-normal-sized stream scenes, one file set per scene) while it generates the resident copies, and calls this tool
after the timed region; the result is the `gen_ps_farm` key of the bench line (never `value`).

Every pass writes into a fresh label folder.  Pass 1 reads the `.pth` files as the dataset writer left them (page cache
warm: the files were just written; a cold first pass additionally pays the disk), pass 2 the same again.  Rate = scenes
/ the slowest worker's own clock (generator ready -> last file written); start-up (interpreter, library load) is
reported apart, as is the wall clock of the whole command.  --share K adds the `share_1ofK` pass: ONE worker over every
K-th scene of the cost-sorted list -- what one GPU of a K-GPU node gets from the claim queue -- with the wall clock from
process start: the job a node really runs per GPU is that short.  Prints one line starting with "JSON ".
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


def one_pass(data_root, save, devices, batch, extra, files=None):
    """files: run over this subset of the scene list (a directory of links is built for it)"""
    if files is not None:
        data_root = subset_root(data_root, files, save + ".subset")
    cmd = [sys.executable, "-m", "gapro_amd.gen_ps", "--save_folder", save, "--data_root", data_root, "--batch_scenes",
           str(batch), "--devices", devices] + extra
    t = time.time()
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, env=dict(os.environ, GAPRO_DRIVER_TIMES="1"))
    wall = time.time() - t
    txt = r.stdout
    for ln in txt.splitlines():  # the workers' own summary and stage timeline, for whoever reads the bench's stderr
        if ln.startswith("[gen_ps]"):
            print("    " + ln, file=sys.stderr)
    done = [(int(a), int(b), float(c)) for a, b, c in
            re.findall(r"(\d+) scenes written, (\d+) skipped/failed, ([\d.]+) s", txt)]
    start = [float(a) for a in re.findall(r"start-up ([\d.]+) s", txt)]
    first = [float(a) for a in re.findall(r"first batch out after ([\d.]+) s", txt)]
    steady = [(float(a), int(b), float(c)) for a, b, c in
              re.findall(r"steady state ([\d.]+) scenes/s \((\d+) scenes in the ([\d.]+) s", txt)]
    io = re.findall(r"(\d+) loader threads, (\d+) loader processes, ([^,\n]+) file I/O", txt)
    n = sum(d for d, _, _ in done)
    slow = max((t for _, _, t in done), default=0.0)
    out = dict(exit_status=r.returncode, scenes=n, failed=sum(f for _, f, _ in done),
               scenes_per_s=round(n / slow, 2) if slow > 0 else 0.0, slowest_worker_s=round(slow, 2),
               wall_s=round(wall, 2), startup_s=round(max(start, default=0.0), 2),
               first_batch_s=round(max(first, default=0.0), 2),
               per_worker_scenes=[d for d, _, _ in done], per_worker_s=[t for _, _, t in done],
               written_files=sum(len([f for f in os.listdir(d) if f.endswith(".pth")])
                                 for d in (save, save + ".DRY_RUN") if os.path.isdir(d)))
    if steady:  # all workers' scenes after their first batch / the slowest worker's time after its first batch
        out["steady_scenes_per_s"] = round(sum(b for _, b, _ in steady) / max(max(c for _, _, c in steady), 1e-3), 2)
    if io:
        out["loader_threads"], out["loader_processes"], out["file_io"] = int(io[0][0]), int(io[0][1]), io[0][2].strip()
    if r.returncode not in (0, 3):
        out["error"] = (r.stderr or txt)[-600:]
    return out


def subset_root(data_root, files, dst):
    """A dataset root holding links to the given scenes only (train/, superpoints/, scans_transform/, scannet_planes/)."""
    for sub in ("train", "superpoints", "scans_transform", "scannet_planes"):
        os.makedirs(os.path.join(dst, sub), exist_ok=True)
    for f in files:
        scan = f[:12]
        os.symlink(os.path.realpath(os.path.join(data_root, "train", f)), os.path.join(dst, "train", f))
        os.symlink(os.path.realpath(os.path.join(data_root, "superpoints", scan + ".pth")),
                   os.path.join(dst, "superpoints", scan + ".pth"))
        os.makedirs(os.path.join(dst, "scans_transform", scan), exist_ok=True)
        os.symlink(os.path.realpath(os.path.join(data_root, "scans_transform", scan, scan + ".txt")),
                   os.path.join(dst, "scans_transform", scan, scan + ".txt"))
        pj = os.path.join(data_root, "scannet_planes", scan + ".json")
        if os.path.exists(pj):
            os.symlink(os.path.realpath(pj), os.path.join(dst, "scannet_planes", scan + ".json"))
    return dst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data_root", required=True)
    ap.add_argument("--devices", default="0")
    ap.add_argument("--passes", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--share", type=int, default=0, help="K: a further pass of ONE worker over every K-th scene of the "
                    "cost-sorted list (the share of one GPU of a K-GPU node)")
    ap.add_argument("--gen-ps-args", default="", help="extra arguments for gen_ps, space separated")
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()
    train = os.path.join(args.data_root, "train")
    files = sorted(f for f in os.listdir(train) if f.endswith("_inst_nostuff.pth"))
    sizes = [os.path.getsize(os.path.realpath(os.path.join(train, f))) for f in files]
    distinct = len({os.path.realpath(os.path.join(train, f)) for f in files})
    out_root = tempfile.mkdtemp(prefix="gapro_farm_")
    extra = [a for a in args.gen_ps_args.split(" ") if a]
    res = dict(workers=len(args.devices.split(",")), devices=args.devices, scene_files=len(files),
               distinct_scene_files=distinct, scene_file_MB=dict(min=round(min(sizes) / 1e6, 1),
                                                                 median=round(sorted(sizes)[len(sizes) // 2] / 1e6, 1),
                                                                 max=round(max(sizes) / 1e6, 1),
                                                                 total_GB=round(sum(sizes) / 1e9, 2)),
               batch_scenes=args.batch, passes=[])
    try:
        for k in range(args.passes):
            p = one_pass(args.data_root, os.path.join(out_root, "labels%d" % k), args.devices, args.batch, extra)
            p["source"] = ".pth files (pass %d%s)" % (k + 1, ": page cache as the dataset writer left it" if k == 0 else "")
            res["passes"].append(p)
            print("pass %d: %s" % (k + 1, json.dumps(p)), file=sys.stderr, flush=True)
        if args.share > 1:
            # the claim queue hands out the cost-sorted list (largest file first); one of K equally fast workers ends
            # up with about every K-th scene of it
            order = [f for _, f in sorted(zip(sizes, files), reverse=True)]
            mine = order[::args.share]
            p = one_pass(args.data_root, os.path.join(out_root, "labels_share"), args.devices.split(",")[0], args.batch,
                         extra, files=mine)
            p["what"] = ("ONE worker over every %d-th scene of the cost-sorted list (%d scenes): the job of one GPU of a "
                         "%d-GPU node; wall_s is the whole command from process start" % (args.share, len(mine), args.share))
            p["projected_node_scenes_per_s"] = round(len(files) / p["wall_s"], 1) if p["wall_s"] > 0 else None
            res["share_1of%d" % args.share] = p
            print("share 1/%d: %s" % (args.share, json.dumps(p)), file=sys.stderr, flush=True)
    finally:
        if not args.keep:
            shutil.rmtree(out_root, ignore_errors=True)
    print("JSON " + json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
