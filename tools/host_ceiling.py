#!/usr/bin/env python3
"""What the HOST of this box can deliver to W GPU workers of the gen_ps farm -- with no GPU involved.

    python tools/host_ceiling.py [--workers 1,2,4,8] [--scenes 1024] [--points 150000]

VERDICT r02 item 3: one MI355X takes ~300 scenes/s; eight workers on one host share its cores.  The host work per
scene is (a) the read -- `torch.load` of the ScanNet tuple + superpoints (unpickling: ~40 ms of one core), or a raw
cache hit (memory-mapped flat file: a memcpy) -- and (b) the write of the 5-tuple (`torch.save`: pickling + file
write); both run in the workers' loader processes.  The main thread's own share (upload calls, job assembly: 1-2 ms per
scene) and PCIe (12 MB per scene, one link per GPU) are small next to them.  This tool puts exactly that load on the
host: W workers x the loader count gen_ps would choose for W workers on this host (`--loader_procs -1`), every loader
task = read one scene as production does (`_read_scene_shm` or a raw-cache read + one pass over the mapped arrays) and
write one 5-tuple of the right sizes (`_save_arrays`).  It deliberately has no GPU in it: on the 1-GPU bench box eight
workers would time-share ONE device (measured: 8 x gen_ps --devices 0,...,0 deliver 57 .. 91 scenes/s in all, against
183 .. 320 for one worker -- process switches on the device, not the host).

Prints per W and source: delivered scenes/s (all workers together), the projected farm rate min(W x gpu, host).
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _task(fn, data_root, raw_cache, out_dir, k):
    """One scene's host work inside a loader process: read, then write a 5-tuple of the right sizes."""
    import numpy as np

    from gapro_amd import gen_ps

    if raw_cache:
        sc = gen_ps.read_raw_cache(gen_ps.raw_cache_path(raw_cache, fn), gen_ps._source_stamp(fn, data_root))
        n = int(len(sc["spp"]))
        # the upload's pass over the mapped pages (page cache -> pinned staging buffer)
        for key in ("coords_float", "mask_feats", "spp", "semantic_label", "instance_label"):
            np.array(sc[key])  # a copy = the memcpy into the staging buffer
    else:
        from multiprocessing import shared_memory

        msg = gen_ps._read_scene_shm(fn, data_root)
        n = [int(shape[0]) for (k, _, shape, _) in msg["layout"] if k == "spp"][0] if msg.get("layout") else \
            int(len(msg["arrays"]["spp"]))
        if msg.get("shm"):
            s = shared_memory.SharedMemory(name=msg["shm"])
            s.close()
            s.unlink()
    arrays = (np.zeros(n, np.int32), np.zeros(n, np.int32), np.ones(n, np.float32),
              np.full(max(1, n // 50), -100.0, np.float32), np.full(max(1, n // 50), -100.0, np.float32))
    gen_ps._save_arrays(os.path.join(out_dir, "s%06d.pth" % k), arrays, None)
    return 1


def loaders_for(workers):
    phys = max(1, (os.cpu_count() or 2) // 2)
    return min(16, max(2, phys // (2 * workers)), (os.cpu_count() or 1) // 4)


def measure(workers, fns, data, raw, out_root):
    import multiprocessing as mp

    from gapro_amd.gen_ps import _loader_init

    n_load = loaders_for(workers)
    pools = [mp.get_context("spawn").Pool(n_load, initializer=_loader_init) for _ in range(workers)]
    outs = []
    for w in range(workers):
        d = os.path.join(out_root, "w%d_%d_%s" % (workers, w, "raw" if raw else "pth"))
        os.makedirs(d)
        outs.append(d)
    for w, p in enumerate(pools):  # warm: interpreters up, page cache hot
        p.starmap(_task, [(fns[i % len(fns)], data, raw, outs[w], 10**6 + i) for i in range(2 * n_load)])
    shares = [fns[w::workers] for w in range(workers)]
    walls = [0.0] * workers

    def drive(w):
        t = time.time()
        pools[w].starmap(_task, [(fn, data, raw, outs[w], i) for i, fn in enumerate(shares[w])], chunksize=1)
        walls[w] = time.time() - t

    ths = [threading.Thread(target=drive, args=(w,)) for w in range(workers)]
    t0 = time.time()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    wall = time.time() - t0
    for p in pools:
        p.close()
        p.join()
    for d in outs:
        shutil.rmtree(d, ignore_errors=True)
    return {"workers": workers, "loaders_per_worker": n_load, "source": "raw" if raw else "pth",
            "scenes_per_s": round(len(fns) / wall, 1), "wall_s": round(wall, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", default="1,2,4,8")
    ap.add_argument("--scenes", type=int, default=1024)
    ap.add_argument("--points", type=int, default=150000)
    ap.add_argument("--gpu-rate", type=float, default=300.0, help="scenes/s one GPU takes (resident inputs)")
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    from gapro_amd import gen_ps
    from gapro_amd.synth import make_scene, write_scannet_layout

    root = tempfile.mkdtemp(prefix="gapro_hc_")
    try:
        data = os.path.join(root, "d")
        cache = os.path.join(root, "raw")
        os.makedirs(cache)
        uniq = 16
        for i in range(uniq):
            sc = make_scene(seed=i % 8, n_points=args.points, n_objects=25, with_walls_json=False,
                            scan_name="scene%04d_00" % i)
            write_scannet_layout(sc, data)
            fn = os.path.join(data, "train", "scene%04d_00_inst_nostuff.pth" % i)
            gen_ps.write_raw_cache(gen_ps.raw_cache_path(cache, fn), gen_ps.read_scene(fn, data), gen_ps._source_stamp(fn, data))
        fns = [os.path.join(data, "train", "scene%04d_00_inst_nostuff.pth" % (i % uniq)) for i in range(args.scenes)]
        rows = []
        for w in [int(x) for x in args.workers.split(",") if x]:
            for raw in (None, cache):
                r = measure(w, fns, data, raw, root)
                r["projected_farm_scenes_per_s"] = round(min(w * args.gpu_rate, r["scenes_per_s"]), 1)
                rows.append(r)
                print("workers %d x %2d loaders, %s: host delivers %7.1f scenes/s -> farm of %d GPUs: min(%d x %.0f, host) = %.0f"
                      % (w, r["loaders_per_worker"], r["source"], r["scenes_per_s"], w, w, args.gpu_rate,
                         r["projected_farm_scenes_per_s"]), flush=True)
        out = {"host": {"cpus": os.cpu_count()}, "points_per_scene": args.points, "scenes": args.scenes,
               "gpu_rate_assumed": args.gpu_rate, "rows": rows}
        if args.json:
            with open(args.json, "w") as f:
                json.dump(out, f, indent=1)
        print("JSON " + json.dumps(out))
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
