#!/usr/bin/env python3
"""Benchmark of the pseudo-label hot path (BASELINE.json metric: scenes/sec pseudo-label generation).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scenes-per-step B] [--points P]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path -- gen_pseudo_label_gaussian_process of reference
gapro/gen_ps_utils.py:293-482, i.e. partition + static schedule + every GP fit + merge + broadcast --
over one batch of B synthetic ScanNet-shaped scenes whose inputs (coords f64, features f32,
superpoint ids i64) are already resident in HBM.  Scenes are independent, so with N GPUs every rank
runs its own batch (weak scaling, no data-path collective; the only collective is the MAX of the
wall time).  Rank 0 prints ONE JSON line.

`roofline` is for the dominant kernel, the batched SVGP fit: achieved = algorithmic FLOPs of one
launch (F_fit = I(8.33 M^3 + 12 D M^2) + M^3/3 + 2 M^2 T + 2 D (M^2 + M T), summed over
the fits the kernel processes) / the kernel's duration measured with HIP events that the library
records on the stream the kernel is launched on (its own CU-masked fit stream).
`cpu_baseline` times the CPU oracle (torch float64 autograd restatement of the reference algorithm,
kind "port") on this box's host cores on one scene of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix (public datasheet); the local guide lists no FP64 row


def build_scene_inputs(seed, n_points, feat_dim):
    from gapro_amd.gen_ps_utils import getInstanceInfo
    from gapro_amd.synth import make_scene

    sc = make_scene(seed=seed, n_points=n_points, n_objects=25, with_walls_json=False)
    xyz = sc.aligned_xyz()
    _, cls, box, vol, _ = getInstanceInfo(xyz, sc.inst, sc.sem)
    if feat_dim == 6:
        feats = sc.default_feats().astype(np.float32)
    else:
        rng = np.random.default_rng(seed + 7)
        proj = rng.standard_normal((6, feat_dim)) / np.sqrt(6.0)
        feats = (sc.default_feats() @ proj).astype(np.float32)
    return dict(coords_float=xyz, mask_feats=feats, spp=sc.spp, instance_cls=cls.astype(np.int64),
                instance_box=box.astype(np.float32), instance_box_volume=vol.astype(np.float32), wall_box=[],
                wall_box_volume=[], instance_classes=18, ground_h=0.1, thresh_spp_occu=0.999)


def lib_route(m, feat_dim):
    from gapro_amd import _lib

    return int(_lib.load().gapro_fit_route(m, feat_dim))


def pmc_traffic(args, scenes_per_step):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/*_pmc_traffic.json,
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over this same command, corrected as calibrated there); None when
    the profiled workload is not the one being run."""
    import glob

    best = None
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        try:
            with open(fn) as fh:
                rec = json.load(fh)
        except (OSError, ValueError):
            continue
        w = rec.get("workload", {})
        if (w.get("scenes_per_step"), w.get("points"), w.get("feat_dim")) == (scenes_per_step, args.points, args.feat_dim):
            best = rec
    return None if best is None else float(best["strip_kernel"]["hbm_bytes_per_launch"])


_CPU_JOBS = None


def _cpu_worker_init(jobs=None):
    import torch

    torch.set_num_threads(1)
    global _CPU_JOBS
    _CPU_JOBS = jobs


def _cpu_worker_warm(_):
    # one tiny fit per worker before the clock starts: the first autograd / LAPACK call of a process pays library
    # initialisation (hundreds of ms with a hundred processes starting at once), which is harness cost
    from oracle.svgp_oracle import fit_gp_spp_oracle

    rng = np.random.default_rng(0)
    feats = rng.standard_normal((12, 6)).astype(np.float32)
    fit_gp_spp_oracle(feats, np.arange(0, 4), np.arange(4, 9), np.arange(9, 12), 2, impl="autograd", dtype="f64")
    time.sleep(0.2)
    return os.getpid()


def _cpu_worker_run(idxs):
    """Run the fits with these indices (into the job list every worker received at start-up)."""
    from oracle.svgp_oracle import fit_gp_spp_oracle

    for i in idxs:
        feats_spp, b1, b2, it = _CPU_JOBS[i]
        fit_gp_spp_oracle(feats_spp, b1, b2, it, 50, impl="autograd", dtype="f64")
    return len(idxs)


def cpu_baseline(scene_kws, target_s=6.0, max_workers=128):
    """Time the CPU oracle on the GP fits of a few scenes of the workload, throughput-style, the way one would
    batch the reference on a many-core host: one single-threaded worker process per physical core (up to
    `max_workers`, BLAS / OpenMP pinned to one thread).  Every scene's partition + schedule is timed on one core.
    The workers receive the fit list once; the clock runs over R passes of it in small mixed chunks (R chosen from a
    first timed pass so that the run takes about `target_s`), i.e. over steady-state work, not process start-up.
    Throughput does not grow monotonically with the process count on a two-socket host, so a quarter, half and all
    of the physical cores are tried and the best is reported.
    scenes/s = scenes' worth of fits finished per second, with the partition + schedule core-seconds spread over
    the same cores added."""
    import concurrent.futures as cf
    import multiprocessing as mp

    from oracle import gen_ps_oracle as O

    phys = max(1, (os.cpu_count() or 2) // 2)  # SMT siblings do not add float64 throughput
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        os.environ[var] = "1"  # inherited by the spawned workers: LAPACK / OpenMP inside a worker stay on its core
    top = max(1, min(int(os.environ.get("GAPRO_CPU_WORKERS", max_workers)), phys))
    t_parts, jobs = [], []
    D = 6
    for kw in scene_kws:
        t0 = time.perf_counter()
        boxes, cls, vol = O.assemble_boxes(kw["coords_float"], kw["instance_cls"], kw["instance_box"],
                                           kw["instance_box_volume"], kw["wall_box"], kw["wall_box_volume"])
        part = O.partition(kw["coords_float"], kw["mask_feats"], kw["spp"], boxes, cls, vol, kw["thresh_spp_occu"])
        events = O.enumerate_schedule(boxes, part.occ_spp, part.n_bbs_per_spp)
        t_parts.append(time.perf_counter() - t0)
        D = part.feats_spp.shape[1]
        jobs += [(part.feats_spp, e.b1_inds, e.b2_inds, e.intersect_inds) for e in events if e.kind == "fit"]
    n = len(scene_kws)
    t_part = float(np.mean(t_parts))
    if not jobs:
        return {"value": 1.0 / t_part, "unit": "scenes/s", "cores": 1, "kind": "port", "sample": "no GP fits"}

    def flops(j):
        m, t = float(len(j[1]) + len(j[2])), float(len(j[3]))
        return 50 * (8.33 * m**3 + 12 * D * m * m) + m**3 / 3 + 2 * m * m * t + 2 * D * (m * m + m * t)

    order = sorted(range(len(jobs)), key=lambda i: flops(jobs[i]), reverse=True)
    # chunks of 4 fits of mixed sizes (one from each quarter of the size-sorted list)
    q = (len(order) + 3) // 4
    chunks = [[order[k + c * q] for c in range(4) if k + c * q < len(order)] for k in range(q)]
    best, tried = None, []
    for workers in sorted({max(1, top // 4), max(1, top // 2), top}):
        ex = cf.ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn"),
                                    initializer=_cpu_worker_init, initargs=(jobs,))
        try:
            list(ex.map(_cpu_worker_warm, range(2 * workers)))  # workers up, libraries initialised
            t1 = time.perf_counter()
            list(ex.map(_cpu_worker_run, chunks))
            first = time.perf_counter() - t1
            reps = int(min(60, max(1, round(target_s / max(first, 1e-3)))))
            t2 = time.perf_counter()
            done = sum(ex.map(_cpu_worker_run, chunks * reps))
            wall = time.perf_counter() - t2
        finally:
            ex.shutdown(wait=False, cancel_futures=True)
        scenes_done = n * done / float(len(jobs))
        value = scenes_done / (wall + scenes_done * t_part / workers)
        tried.append("%d workers %.2f scenes/s" % (workers, value))
        if best is None or value > best[0]:
            best = (value, workers, reps, wall, first)
    value, workers, reps, wall, first = best
    return {"value": value, "unit": "scenes/s", "cores": workers, "kind": "port",
            "sample": "the %d GP fits of %d scenes of the workload (torch float64 autograd oracle), %d passes in small "
                      "mixed chunks over %d single-threaded worker processes: %.1f s (one untimed pass before: %.1f s); "
                      "partition+schedule %.2f core-s per scene on the same cores; tried: %s"
                      % (len(jobs), n, reps, workers, wall, first, t_part, ", ".join(tried))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scenes-per-step", type=int, default=256)
    ap.add_argument("--points", type=int, default=150000)
    ap.add_argument("--feat-dim", type=int, default=6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--force-staged", action="store_true", help="never use the strip-streaming fit kernel (A/B)")
    ap.add_argument("--stage-times", action="store_true", help="print per-stage wall clock to stderr (adds syncs)")
    ap.add_argument("--trace", action="store_true", help="print the host-side stage timeline of the timed steps to stderr")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    if args.cpu_baseline_child:
        # helper process of the CPU baseline: started by the parent before it touched the GPU, runs when told
        if sys.stdin.readline().strip() == "go":
            print(json.dumps(cpu_baseline([build_scene_inputs(s, args.points, args.feat_dim) for s in range(4)])),
                  flush=True)
        return

    # CPU baseline on rank 0 at N=1 only.  Its worker processes must not be forked/exec'd from a process that
    # holds a HIP context, so a helper process is started NOW, before this one touches the GPU; it idles until
    # the GPU timing is over (a 30 s burst of 32 busy host processes right before the timed region left the
    # first GPU steps measurably slow) and then times the oracle.
    cpu, cpu_child = None, None
    if world == 1 and not args.no_cpu_baseline:
        import subprocess

        cpu_child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--points",
                                      str(args.points), "--feat-dim", str(args.feat_dim)], stdin=subprocess.PIPE,
                                     stdout=subprocess.PIPE, text=True)

    import torch
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from gapro_amd.pipeline import Pipeline, make_job

    B = args.scenes_per_step
    # a few distinct scenes, reused cyclically; every rank gets its own seeds
    n_distinct = min(B, 4)
    scene_kws = [build_scene_inputs(1000 * rank + s, args.points, args.feat_dim) for s in range(n_distinct)]
    resident = []
    for kw in scene_kws:  # inputs live in HBM before the timed region
        resident.append(dict(kw, coords_float=torch.from_numpy(kw["coords_float"]).to(dev),
                             mask_feats=torch.from_numpy(kw["mask_feats"]).to(dev),
                             spp=torch.from_numpy(kw["spp"]).to(dev)))
    pipe = Pipeline(device=local_rank, training_iter=50, force_staged=args.force_staged)

    def make_jobs():
        return [make_job(r["coords_float"], r["mask_feats"], r["spp"], r["instance_cls"], r["instance_box"],
                         r["instance_box_volume"], r["wall_box"], r["wall_box_volume"], 18, 0.1, 0.999, device=dev)
                for r in (resident[i % n_distinct] for i in range(B))]

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    if args.warmup:
        pipe.run_pipelined(make_jobs() for _ in range(args.warmup))
    pipe.profile_fit = True
    pipe.fit_events = []
    pipe.profile_stages = args.stage_times
    if args.trace:
        pipe.trace = []
    barrier()
    t0 = time.perf_counter()
    if args.stage_times:
        for _ in range(args.steps):
            pipe.run(make_jobs())
    else:  # K steps back to back; the host work of step i+1 overlaps the fit launch of step i
        pipe.run_pipelined(make_jobs() for _ in range(args.steps))  # jobs of step i+2 are built during fit(i)
    barrier()
    elapsed = time.perf_counter() - t0
    from gapro_amd.dist_utils import barrier_and_max

    elapsed = barrier_and_max(elapsed, dev)  # MAX over ranks

    # device-side duration of every fit launch, from HIP events the library records on the streams its kernels
    # run on: (staged kernel ms, strip kernel ms, first start -> last end ms)
    fit_ms3 = [ev.read() for ev in pipe.fit_events]
    fit_ms = [t[2] for t in fit_ms3]
    if rank == 0:
        print("fit launch device ms per step (staged, strip, span, small strip): " + "  ".join("%.1f/%.1f/%.1f/%.1f" % t for t in fit_ms3),
              file=sys.stderr)
    fit_fl = [ev.flops for ev in pipe.fit_events]
    stats = pipe.last_stats
    if rank == 0 and getattr(pipe, "last_fit_m", None) is not None:
        m = pipe.last_fit_m
        edges = [0, 32, 64, 96, 128, 192, 256, 512, 1 << 30]
        hist = np.histogram(m, bins=edges)[0]
        w = np.histogram(m, bins=edges, weights=m.astype(np.float64) ** 3)[0]
        print("inducing-set size M per fit (last launch): " + ", ".join(
            "(%d,%s] n=%d M^3-share=%.0f%%" % (edges[i], edges[i + 1] if i < 7 else "inf", hist[i], 100 * w[i] / w.sum())
            for i in range(8) if hist[i]) + "; max M %d" % m.max(), file=sys.stderr)
    if rank == 0 and args.trace:
        t_first = pipe.trace[0][0]
        ids = {}
        for t, bid, name in pipe.trace:
            print("  %9.2f ms  batch %d  %s" % (1e3 * (t - t_first), ids.setdefault(bid, len(ids)), name), file=sys.stderr)
    if rank == 0 and args.stage_times:
        print("stage times per step (ms): " + ", ".join("%s %.2f" % (k, 1e3 * v / args.steps)
                                                         for k, v in pipe.stage_times.items()), file=sys.stderr)
    if rank == 0:
        avg_ms = float(np.mean(fit_ms)) if fit_ms else 0.0
        launch_tflops = (float(np.mean(fit_fl)) / (avg_ms * 1e-3) / 1e12) if avg_ms > 0 else 0.0
        # dominant kernel: the strip-streaming fit kernel (fits with round_up(M, 32) <= 128)
        strip_ms = float(np.mean([t[1] for t in fit_ms3])) if fit_ms3 else 0.0
        strip_fl = float(np.mean([ev.flops_strip for ev in pipe.fit_events])) if fit_ms3 else 0.0
        staged_ms = float(np.mean([t[0] for t in fit_ms3])) if fit_ms3 else 0.0
        staged_fl = float(np.mean([ev.flops_staged for ev in pipe.fit_events])) if fit_ms3 else 0.0
        achieved = (strip_fl / (strip_ms * 1e-3) / 1e12) if strip_ms > 0 else 0.0
        n_strip = int(sum(1 for ev in pipe.fit_events[-1:] for v in ev.m if lib_route(int(v), args.feat_dim) == 0))
        traffic = pmc_traffic(args, B)
        descs = stats.get("fit") or {}
        out = {
            "metric": "scenes/sec pseudo-label gen (ScanNetV2-train-shaped synthetic scenes)",
            "value": B * args.steps * world / elapsed,
            "unit": "scenes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "configs[1]-shaped scenes (single ScanNet-like scene: %d points, 25 objects, "
                                   "D=%d, ~50-pt object / ~400-pt planar superpoints), %d scenes per step per GPU, "
                                   "50 Adam steps per GP fit" % (args.points, args.feat_dim, B),
                       "scenes_per_step_per_gpu": B, "points_per_scene": args.points, "feat_dim": args.feat_dim,
                       "gp_fits_per_step_per_gpu": int(stats.get("n_fits", 0)),
                       "parallelism": "scene-sharded x%d, no collective" % world},
            "roofline": {"bound": "mfma",
                         "kernel": "k_svgp_fit_strip<%d,%d> (batched SVGP fit, fits with 64 < M_p <= 128; f64 MFMA "
                                   "16x16x4)" % ((args.feat_dim, args.feat_dim) if args.feat_dim in (6, 32) else (32, 0)),
                         "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "avg_launch_ms": strip_ms, "flops_per_launch": strip_fl, "fits_per_launch": n_strip,
                         "timing": "HIP events recorded by the library on the stream the kernel is launched on"},
            # the whole fit launch: strip kernel, small-fit strip kernel (M_p <= 64) and staged kernel (M_p > 128) side by side
            "fit_launch": {"avg_ms_first_start_to_last_end": avg_ms, "flops": float(np.mean(fit_fl)) if fit_fl else 0.0,
                           "tflops": launch_tflops, "frac_of_fp64_mfma_peak": launch_tflops / FP64_MFMA_PEAK_TFLOPS,
                           "fits_per_s": (stats.get("n_fits", 0) / (avg_ms * 1e-3)) if avg_ms > 0 else 0.0,
                           "staged_kernel_avg_ms": staged_ms, "staged_kernel_flops": staged_fl,
                           "small_strip_kernel_avg_ms": float(np.mean([t[3] for t in fit_ms3])) if fit_ms3 else 0.0,
                           "small_strip_kernel_flops": float(np.mean([ev.flops_small for ev in pipe.fit_events])) if fit_ms3 else 0.0,
                           "share_of_step": (sum(fit_ms) / (1e3 * elapsed)) if elapsed > 0 else None},
        }
        if cpu_child is not None:
            try:
                reply, _ = cpu_child.communicate("go\n", timeout=600)
                cpu = json.loads(reply.strip().splitlines()[-1])
            except Exception as e:  # noqa: BLE001 - the GPU line is still worth printing
                print("cpu baseline failed: %r" % (e,), file=sys.stderr)
                cpu_child.kill()
        out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
