#!/usr/bin/env python3
"""Benchmark of the pseudo-label hot path (BASELINE.json metric: scenes/sec pseudo-label generation).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scenes-per-step B] [--workload stream|fixed]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (a child
`python -m torch.distributed.run ...` created before this process imports torch or touches the GPU; its
rank 0 prints the JSON line).  Under a launcher, WORLD_SIZE must equal --gpus.

A "step" is one pass of the hot path -- gen_pseudo_label_gaussian_process of reference
gapro/gen_ps_utils.py:293-482, i.e. partition + static schedule + every GP fit + merge + broadcast --
over one batch of B synthetic ScanNet-shaped scenes whose inputs (coords f64, features f32,
superpoint ids i64) are already resident in HBM.  Scenes are independent, so with N GPUs every rank
runs its own batch (weak scaling, no data-path collective; the only collective is the MAX of the
wall time).  Rank 0 prints ONE JSON line.

Workloads (`config.workload`):
  stream (default)  SURVEY.md 8d config 3's scene stream on one GPU: `--distinct` (64) different seeds per rank,
                    N ~ logN(median 150k, sigma 0.5) clipped to [40k, 450k], 10..40 objects, wall quads for ~70 %
                    of the scenes, repeated cyclically to B scenes per step.
  fixed             the round-1 line: 4 scenes of exactly 150k points / 25 objects, no walls, repeated 64x
                    (kept as the extra key `fixed_size_line` of the default run, for continuity).

`roofline` is for the dominant kernel, the batched SVGP fit: achieved = algorithmic FLOPs of one
launch (F_fit = I(8.33 M^3 + 12 D M^2) + M^3/3 + 2 M^2 T + 2 D (M^2 + M T), summed over
the fits the kernel processes) / the kernel's duration measured with HIP events that the library
records on the stream the kernel is launched on.  `roofline.traffic` is NOT measured in this run: it is
read from the committed PMC passes over this same command (`traffic_source` names the file) or null.
`partition` prices the HBM-bound integer half (prepare + pool + broadcast kernels) against SURVEY 8d's
68 B/point (D = 6) with torch events on the stream the kernels run on.
`cpu_baseline` times the CPU oracle (torch float64 autograd restatement of the reference algorithm,
kind "port") on this box's host cores on the first scenes of the same workload.
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
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s (~6.3 TB/s achievable)


def scene_params(workload, seed, points):
    """(n_points, n_objects, with_walls_json) of scene `seed` of the workload."""
    if workload == "fixed":
        return points, 25, False
    rng = np.random.default_rng(0x5CA77E + seed)  # size stream independent of the scene's own random stream
    n = int(np.clip(150000.0 * np.exp(0.5 * rng.standard_normal()), 40000, 450000))
    return n, None, None  # make_scene draws K ~ U{10..40} and walls with p = 0.7 from the scene seed


def build_scene_inputs(seed, n_points, feat_dim, workload="fixed"):
    from gapro_amd.gen_ps_utils import getInstanceInfo
    from gapro_amd.synth import make_scene

    n, k, walls = scene_params(workload, seed, n_points)
    # vertex order: the stream workload's scenes come in mesh order (Z-order within every face: consecutive vertices
    # are neighbours and mostly share a superpoint, as in a reconstructed ScanNet mesh; superpoint runs of ~20
    # vertices); round 1's fixed-size scenes keep the order their points were sampled in (random within a face, runs of
    # ~2), which only the partition kernels can tell apart -- every output is invariant under the vertex order
    sc = make_scene(seed=seed, n_points=n, n_objects=k, with_walls_json=walls,
                    mesh_order=(workload == "stream" and not os.environ.get("GAPRO_BENCH_SAMPLED_ORDER")))
    xyz = sc.aligned_xyz()
    _, cls, box, vol, _ = getInstanceInfo(xyz, sc.inst, sc.sem)
    wall_box, wall_vol = [], []
    if sc.quads is not None:  # reference gen_ps.py:77: wall boxes from the ScanNet-Planes quads of the scan
        import tempfile

        from gapro_amd.scannet_planes import get_wall_boxes

        with tempfile.TemporaryDirectory() as td:
            os.makedirs(os.path.join(td, "scans_transform", sc.scan_name))
            os.makedirs(os.path.join(td, "scannet_planes"))
            with open(os.path.join(td, "scans_transform", sc.scan_name, sc.scan_name + ".txt"), "w") as f:
                f.write("axisAlignment = " + " ".join(repr(float(v)) for v in sc.axis_align.reshape(-1)) + "\n")
            with open(os.path.join(td, "scannet_planes", sc.scan_name + ".json"), "w") as f:
                json.dump(sc.quads, f)
            _, wb, wv = get_wall_boxes(sc.scan_name, data_root=td)
        if len(wb):
            wall_box, wall_vol = np.asarray(wb, np.float32), np.asarray(wv, np.float32)
    if feat_dim == 6:
        feats = sc.default_feats().astype(np.float32)
    else:
        rng = np.random.default_rng(seed + 7)
        proj = rng.standard_normal((6, feat_dim)) / np.sqrt(6.0)
        feats = (sc.default_feats() @ proj).astype(np.float32)
    return dict(coords_float=xyz, mask_feats=feats, spp=sc.spp, instance_cls=cls.astype(np.int64),
                instance_box=box.astype(np.float32), instance_box_volume=vol.astype(np.float32), wall_box=wall_box,
                wall_box_volume=wall_vol, instance_classes=18, ground_h=0.1, thresh_spp_occu=0.999)


def lib_route(m, feat_dim):
    from gapro_amd import _lib

    return int(_lib.load().gapro_fit_route(m, feat_dim))


def pmc_traffic(workload, scenes_per_step, points, feat_dim, distinct):
    """(HBM bytes per launch of the dominant kernel, file) from the committed PMC passes (profiles/*_pmc_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over this same command, corrected as calibrated there); (None, None)
    when no committed pass profiled the workload being run."""
    import glob

    best = None
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        try:
            with open(fn) as fh:
                rec = json.load(fh)
        except (OSError, ValueError):
            continue
        w = rec.get("workload", {})
        key = (w.get("name", "fixed"), w.get("scenes_per_step"), w.get("feat_dim"))
        if key != (workload, scenes_per_step, feat_dim):
            continue
        if workload == "fixed" and w.get("points") != points:
            continue
        if workload == "stream" and w.get("distinct") != distinct:
            continue
        best = (rec, fn)
    if best is None:
        return None, None
    key = "fit_launch" if "fit_launch" in best[0] else "strip_kernel"
    return float(best[0][key]["hbm_bytes_per_launch"]), os.path.relpath(best[1], ROOT)


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


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU, RCCL rendezvous on 127.0.0.1) as
    a child `python -m torch.distributed.run`, created before this process has imported torch or touched the GPU;
    the child's rank 0 prints the JSON line on the inherited stdout.  Returns the child's exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:  # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


class _DryPipeline:
    """Stand-in for the device pipeline in --dry-run (control-path test of the N-rank launch on a host without a
    GPU): every step sleeps a rank-dependent few milliseconds.  Never used for a measurement."""

    def __init__(self, rank):
        self.rank = rank

    def step(self):
        time.sleep(0.002 * (self.rank + 1))


def dry_run(args, rank, world, local_rank):
    import torch.distributed as dist

    from gapro_amd.dist_utils import barrier_and_max

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    pipe = _DryPipeline(rank)
    for _ in range(args.warmup):
        pipe.step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pipe.step()
    if world > 1:
        dist.barrier()
    elapsed = barrier_and_max(time.perf_counter() - t0)
    if rank == 0:
        print(json.dumps({"metric": "scenes/sec pseudo-label gen (ScanNetV2-train-shaped synthetic scenes)",
                          "value": None, "unit": "scenes/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "dry_run": True,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
                          "data": "none (control-path dry run, no GPU work)",
                          "config": {"workload": "dry-run", "parallelism": "scene-sharded x%d, no collective" % world}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_workload(args, pipe, dev, rank, world, workload, B, distinct, steps, warmup, barrier, reduce_dev=None):
    """Timed region of one workload on this rank.  Returns a dict of measurements (rank-local, elapsed = MAX over
    ranks)."""
    import torch

    from gapro_amd.dist_utils import barrier_and_max
    from gapro_amd.pipeline import make_job

    n_distinct = min(B, distinct)
    # every rank gets its own seeds
    scene_kws = [build_scene_inputs(1000 * rank + s, args.points, args.feat_dim, workload) for s in range(n_distinct)]
    resident = []
    for kw in scene_kws:  # inputs live in HBM before the timed region
        resident.append(dict(kw, coords_float=torch.from_numpy(kw["coords_float"]).to(dev),
                             mask_feats=torch.from_numpy(kw["mask_feats"]).to(dev),
                             spp=torch.from_numpy(kw["spp"]).to(dev)))

    def make_jobs():
        return [make_job(r["coords_float"], r["mask_feats"], r["spp"], r["instance_cls"], r["instance_box"],
                         r["instance_box_volume"], r["wall_box"], r["wall_box_volume"], 18, 0.1, 0.999, device=dev)
                for r in (resident[i % n_distinct] for i in range(B))]

    pipe.profile_fit = False
    if warmup:
        pipe.run_pipelined(make_jobs() for _ in range(warmup))
    pipe.profile_fit = True
    pipe.fit_events = []
    pipe.part_events = []
    pipe.profile_stages = args.stage_times
    if args.trace:
        pipe.trace = []
    barrier()
    t0 = time.perf_counter()
    if args.stage_times:
        for _ in range(steps):
            pipe.run(make_jobs())
    else:  # K steps back to back; the host work of step i+1 overlaps the fit launch of step i
        pipe.run_pipelined(make_jobs() for _ in range(steps))  # jobs of step i+2 are built during fit(i)
    barrier()
    elapsed = barrier_and_max(time.perf_counter() - t0, reduce_dev)  # MAX over ranks
    # device-side duration of every fit launch, from HIP events the library records on the streams its kernels
    # run on: (staged kernel ms, strip kernel ms, first start -> last end ms, small-fit strip kernel ms, cluster ms)
    fit_ms3 = [ev.read() for ev in pipe.fit_events]
    # where every launch lies on the time axis of the first one: consecutive launches overlap (fit(i+1) is enqueued in
    # the tail of fit(i), and a start event fires when its stream reaches it, not when the kernel gets CUs), so the mean
    # of the spans counts the overlapped time twice; the union of the intervals counts it once
    fit_iv = [ev.offsets(pipe.fit_events[0]) for ev in pipe.fit_events] if pipe.fit_events else []
    part = {}
    for ev in pipe.part_events:
        d = part.setdefault(ev["name"], dict(ms=0.0, points=0, n=0))
        d["ms"] += ev["start"].elapsed_time(ev["end"])
        d["points"] += ev["points"]
        d["n"] += 1
    # the integer half alone on the GPU: one un-pipelined step (prepare, pool | fits | broadcast, each synchronised)
    fit_events = list(pipe.fit_events)
    pipe.part_events = []
    torch.cuda.synchronize(dev)
    pipe.run(make_jobs())
    torch.cuda.synchronize(dev)
    probe = {}
    for ev in pipe.part_events:
        d = probe.setdefault(ev["name"], dict(ms=0.0, points=0, n=0))
        d["ms"] += ev["start"].elapsed_time(ev["end"])
        d["points"] += ev["points"]
        d["n"] += 1
    pipe.fit_events = fit_events
    return dict(elapsed=elapsed, fit_ms3=fit_ms3, fit_iv=fit_iv, fit_events=fit_events, part=part, part_probe=probe,
                stats=dict(pipe.last_stats), last_fit_m=getattr(pipe, "last_fit_m", None), scene_kws=scene_kws,
                points_per_step=sum(int(resident[i % n_distinct]["coords_float"].shape[0]) for i in range(B)),
                n_distinct=n_distinct, trace=pipe.trace, stage_times=dict(pipe.stage_times))


def union_length(intervals):
    """Total length of the union of [lo, hi] intervals (time in which at least one launch was in flight)."""
    total, cur_lo, cur_hi = 0.0, None, None
    for lo, hi in sorted(intervals):
        if cur_hi is None or lo > cur_hi:
            total += (cur_hi - cur_lo) if cur_hi is not None else 0.0
            cur_lo, cur_hi = lo, hi
        else:
            cur_hi = max(cur_hi, hi)
    return total + ((cur_hi - cur_lo) if cur_hi is not None else 0.0)


def summarize(args, res, B, world, steps, workload, peak):
    """Bench keys of one measured workload."""
    fit_ms3, evs, elapsed = res["fit_ms3"], res["fit_events"], res["elapsed"]
    fit_ms = [t[2] for t in fit_ms3]
    fit_fl = [ev.flops for ev in evs]
    span_ms = float(np.mean(fit_ms)) if fit_ms else 0.0  # mean first start -> last end, overlaps counted twice
    union = union_length(res.get("fit_iv") or [])
    avg_ms = union / len(fit_ms) if fit_ms and union > 0 else span_ms
    launch_tflops = (float(np.mean(fit_fl)) / (avg_ms * 1e-3) / 1e12) if avg_ms > 0 else 0.0
    # the kernel that holds the largest share of the launch's algorithmic FLOPs is the one priced in `roofline`
    kernels = {
        "strip": ("k_svgp_fit_strip<%s> 512 threads (64 < M_p <= 128)", 1, "flops_strip"),
        "staged": ("k_svgp_fit<WPS> LDS-staged (128 < M_p <= 512) + generic beyond", 0, "flops_staged"),
        "small": ("k_svgp_fit_strip<%s> 256 threads, two fits per CU (M_p <= 64)", 3, "flops_small"),
        "cluster": ("k_svgp_fit_cluster: one fit over 2..32 workgroups (M_p > 384)", 4, "flops_cluster"),
    }
    per = {}
    for key, (name, slot, attr) in kernels.items():
        ms = float(np.mean([t[slot] for t in fit_ms3])) if fit_ms3 else 0.0
        fl = float(np.mean([getattr(ev, attr) for ev in evs])) if evs else 0.0
        per[key] = dict(kernel=name.replace("%s", "%d,%d" % ((args.feat_dim, args.feat_dim) if args.feat_dim in (6, 32) else (32, 0))),
                        avg_ms=ms, flops=fl, tflops=(fl / (ms * 1e-3) / 1e12) if ms > 0 else 0.0)
    stats = res["stats"]
    out = {"value": B * steps * world / elapsed, "ms_per_step": 1e3 * elapsed / steps,
           "fits_per_step": int(stats.get("n_fits", 0)), "points_per_step": int(res["points_per_step"]),
           "distinct_scenes": int(res["n_distinct"])}
    traffic, src = pmc_traffic(workload, B, args.points, args.feat_dim, res["n_distinct"])
    # The fit kernels of one gapro_svgp_fit_batch launch run SIDE BY SIDE on the library's streams and share the CUs, so
    # a single kernel's own duration is inflated by its neighbours; the figure that can be priced against the MFMA
    # peak is the launch: algorithmic FLOPs of all its fits / (first kernel start -> last kernel end).
    out["roofline"] = {"bound": "mfma",
                       "kernel": "gapro_svgp_fit_batch launch = " + " + ".join(
                           k for k in ("cluster", "staged", "strip", "small") if per[k]["flops"] > 0)
                                 + " fit kernels side by side; f64 MFMA 16x16x4",
                       "achieved": launch_tflops, "peak": peak, "unit": "TFLOP/s", "frac": launch_tflops / peak,
                       "traffic": traffic, "traffic_source": src, "avg_launch_ms": avg_ms,
                       "flops_per_launch": float(np.mean(fit_fl)) if fit_fl else 0.0,
                       "mean_span_ms": span_ms,
                       "timing": "HIP events recorded by the library on the streams the kernels are launched on; "
                                 "avg_launch_ms = union of the launches' [first kernel start, last kernel end] "
                                 "intervals over the timed region / number of launches: consecutive launches overlap "
                                 "(the next one is enqueued in the tail of the previous one and its start events "
                                 "fire at once), mean_span_ms counts that time twice"}
    out["fit_launch"] = {"avg_ms_first_start_to_last_end": avg_ms, "mean_span_ms": span_ms, "flops": float(np.mean(fit_fl)) if fit_fl else 0.0,
                         "tflops": launch_tflops, "frac_of_fp64_mfma_peak": launch_tflops / peak,
                         "fits_per_s": (stats.get("n_fits", 0) / (avg_ms * 1e-3)) if avg_ms > 0 else 0.0,
                         "kernels": per, "share_of_step": (sum(fit_ms) / (1e3 * elapsed)) if elapsed > 0 else None,
                         # consecutive launches overlap (the next one starts in the tail of the previous one), which
                         # stretches every launch's first-start -> last-end span; the chip's rate over the timed region:
                         "tflops_over_timed_region": (sum(fit_fl) / elapsed / 1e12) if elapsed > 0 else None}
    # HBM-bound half: SURVEY 8d algorithmic bytes per point = 24 (xyz f64) + 4 D (feats f32) + 8 (spp i64) + 12 (out)
    bpp = 24 + 4 * args.feat_dim + 8 + 12
    part = res.get("part_probe") or res["part"]
    tot_ms = sum(v["ms"] for v in part.values())
    pts = part.get("pool", {}).get("points", 0)
    out["partition"] = {
        "kernels": "gapro_partition_prepare_batch (k_stats, k_flags, k_scan_*, k_rank_lookup) + gapro_partition_pool_batch "
                   "(k_pool, k_pool_finalize) + gapro_broadcast_labels_batch (k_broadcast)",
        "bound": "hbm", "bytes_per_point": bpp, "points": int(pts),
        "ms": {k: v["ms"] for k, v in part.items()}, "total_ms": tot_ms,
        "GB/s": (bpp * pts / (tot_ms * 1e-3) / 1e9) if tot_ms > 0 else 0.0,
        "frac_of_hbm": (bpp * pts / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if tot_ms > 0 else 0.0,
        "pool_only_GB/s": (bpp * pts / (part["pool"]["ms"] * 1e-3) / 1e9) if part.get("pool", {}).get("ms", 0) > 0 else 0.0,
        "timing": "torch events on the stream the kernels are launched on, one extra un-pipelined step after the timed "
                  "region (inside the pipelined steps these short kernels queue behind the fit kernels, which an event "
                  "pair would count as kernel time)"}
    m = res["last_fit_m"]
    if m is not None and len(m):
        if os.environ.get("GAPRO_DUMP_FIT_M"):  # the M of every fit of the last launch, for tools/bench_fit.py --mix
            np.save(os.environ["GAPRO_DUMP_FIT_M"], np.sort(m))
        edges = [0, 32, 64, 96, 128, 192, 256, 384, 512, 1 << 30]
        hist = np.histogram(m, bins=edges)[0]
        w = np.histogram(m, bins=edges, weights=m.astype(np.float64) ** 3)[0]
        out["fit_m_hist"] = {"edges": edges[:-1] + ["inf"], "fits": [int(v) for v in hist],
                             "m3_share": [round(float(v / w.sum()), 4) for v in w], "max_m": int(m.max())}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scenes-per-step", type=int, default=256)
    ap.add_argument("--workload", choices=["stream", "fixed"], default="stream")
    ap.add_argument("--distinct", type=int, default=64, help="distinct scene seeds per rank (stream workload)")
    ap.add_argument("--points", type=int, default=150000, help="points per scene of the fixed workload")
    ap.add_argument("--feat-dim", type=int, default=6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fixed-line", action="store_true", help="skip the round-1 fixed-size workload (extra key)")
    ap.add_argument("--no-driver-line", action="store_true", help="skip the disk-inclusive gen_ps run (extra key)")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dry-run", action="store_true",
                    help="control-path test: N ranks over gloo, no GPU work, prints a line with dry_run=true")
    ap.add_argument("--share-gpu", action="store_true",
                    help="ranks beyond the device count share GPUs (gloo control plane; test of the N>1 path on one GPU)")
    ap.add_argument("--force-staged", action="store_true", help="never use the strip-streaming fit kernel (A/B)")
    ap.add_argument("--serial-kernels", action="store_true",
                    help="diagnostic: the fit kernels of a launch one after the other on one stream (each kernel's own rate)")
    ap.add_argument("--stage-times", action="store_true", help="print per-stage wall clock to stderr (adds syncs)")
    ap.add_argument("--trace", action="store_true", help="print the host-side stage timeline of the timed steps to stderr")
    args = ap.parse_args()

    if args.cpu_baseline_child:
        # helper process of the CPU baseline: started by the parent before it touched the GPU, runs when told
        if sys.stdin.readline().strip() == "go":
            print(json.dumps(cpu_baseline([build_scene_inputs(s, args.points, args.feat_dim, args.workload)
                                           for s in range(4)])), flush=True)
        return 0

    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, sys.argv[1:])  # before torch is imported / the GPU is touched
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world), file=sys.stderr)
        return 2
    if args.dry_run:
        dry_run(args, rank, world, local_rank)
        return 0

    # CPU baseline on rank 0 at N=1 only.  Its worker processes must not be forked/exec'd from a process that
    # holds a HIP context, so a helper process is started NOW, before this one touches the GPU; it idles until
    # the GPU timing is over (a 30 s burst of 32 busy host processes right before the timed region left the
    # first GPU steps measurably slow) and then times the oracle.
    cpu, cpu_child = None, None
    if world == 1 and not args.no_cpu_baseline:
        import subprocess

        cpu_child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--points",
                                      str(args.points), "--feat-dim", str(args.feat_dim), "--workload", args.workload],
                                     stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)

    import torch
    import torch.distributed as dist

    n_dev = torch.cuda.device_count()
    shared = args.share_gpu and n_dev < world
    if world > n_dev and not args.share_gpu:
        print("bench.py: %d ranks but %d visible GPUs (pass --share-gpu to test the path on fewer)" % (world, n_dev),
              file=sys.stderr)
        return 2
    dev_index = local_rank % max(n_dev, 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:  # RCCL refuses two ranks on one device; the only collectives are control-plane
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    from gapro_amd.pipeline import Pipeline

    B = args.scenes_per_step
    pipe = Pipeline(device=dev_index, training_iter=50, force_staged=args.force_staged)
    if args.serial_kernels:
        pipe.opt.reserved |= 2

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    # measured matrix-core peak next to the datasheet figure the roofline uses
    peak_measured = None
    if rank == 0:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import mfma_peak

            peak_measured = mfma_peak.measure(dev_index, iters=10000)
        except Exception as e:  # noqa: BLE001 - diagnostic only
            print("mfma peak micro-benchmark failed: %r" % (e,), file=sys.stderr)
    barrier()

    res = run_workload(args, pipe, dev, rank, world, args.workload, B, args.distinct, args.steps, args.warmup, barrier,
                       reduce_dev=None if shared else dev)
    main_keys = summarize(args, res, B, world, args.steps, args.workload, FP64_MFMA_PEAK_TFLOPS) if rank == 0 else None
    if rank == 0:
        print("fit launch device ms per step (staged, strip, span, small strip, cluster): "
              + "  ".join("%.1f/%.1f/%.1f/%.1f/%.1f" % t for t in res["fit_ms3"]), file=sys.stderr)
        if args.trace and res["trace"]:
            t_first = res["trace"][0][0]
            ids = {}
            for t, bid, name in res["trace"]:
                print("  %9.2f ms  batch %d  %s" % (1e3 * (t - t_first), ids.setdefault(bid, len(ids)), name), file=sys.stderr)
        if args.stage_times:
            print("stage times per step (ms): " + ", ".join("%s %.2f" % (k, 1e3 * v / args.steps)
                                                             for k, v in res["stage_times"].items()), file=sys.stderr)

    fixed_keys = None
    if world == 1 and args.workload == "stream" and not args.no_fixed_line and not args.stage_times:
        fres = run_workload(args, pipe, dev, rank, world, "fixed", B, 4, args.steps, 1, barrier, reduce_dev=dev)
        fk = summarize(args, fres, B, world, args.steps, "fixed", FP64_MFMA_PEAK_TFLOPS)
        fixed_keys = {"workload": "round-1 line: 4 scenes of %d points, 25 objects, no walls, repeated to %d per step"
                                  % (args.points, B), "value": fk["value"], "ms_per_step": fk["ms_per_step"],
                      "fits_per_step": fk["fits_per_step"], "roofline": fk["roofline"],
                      "fit_launch_tflops": fk["fit_launch"]["tflops"],
                      "fit_launch_frac_of_fp64_mfma_peak": fk["fit_launch"]["frac_of_fp64_mfma_peak"]}

    if rank == 0:
        wl = ("SURVEY 8d config-3 scene stream: %d distinct seeds per GPU, N ~ logN(150k, 0.5) clipped [40k, 450k], "
              "10..40 objects, wall quads for ~70%% of the scenes, D=%d, %d scenes per step per GPU, 50 Adam steps per "
              "GP fit" % (res["n_distinct"], args.feat_dim, B)) if args.workload == "stream" else \
             ("configs[1]-shaped scenes (single ScanNet-like scene: %d points, 25 objects, D=%d, ~50-pt object / "
              "~400-pt planar superpoints), %d scenes per step per GPU, 50 Adam steps per GP fit"
              % (args.points, args.feat_dim, B))
        out = {
            "metric": "scenes/sec pseudo-label gen (ScanNetV2-train-shaped synthetic scenes)",
            "value": main_keys["value"],
            "unit": "scenes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": main_keys["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": wl, "scenes_per_step_per_gpu": B, "distinct_scenes_per_gpu": res["n_distinct"],
                       "points_per_step_per_gpu": main_keys["points_per_step"], "feat_dim": args.feat_dim,
                       "gp_fits_per_step_per_gpu": main_keys["fits_per_step"],
                       "parallelism": "scene-sharded x%d, no collective%s" % (world, " (ranks share GPUs: test mode)" if shared else "")},
            "roofline": main_keys["roofline"],
            "fit_launch": main_keys["fit_launch"],
            "partition": main_keys["partition"],
            "fit_m_hist": main_keys.get("fit_m_hist"),
            "peak_measured": peak_measured,
        }
        if fixed_keys is not None:
            out["fixed_size_line"] = fixed_keys
        if world == 1 and not args.no_driver_line and not args.stage_times:
            out["gen_ps_disk_inclusive"] = driver_line()
        if cpu_child is not None:
            try:
                reply, _ = cpu_child.communicate("go\n", timeout=900)
                cpu = json.loads(reply.strip().splitlines()[-1])
            except Exception as e:  # noqa: BLE001 - the GPU line is still worth printing
                print("cpu baseline failed: %r" % (e,), file=sys.stderr)
                cpu_child.kill()
        out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def driver_line(scenes=1536, unique=24, procs=16):
    """Disk- and PCIe-inclusive rate of the gen_ps driver (torch.load of ScanNet-layout .pth files, host
    preprocessing, upload, generation, torch.save of the 5-tuples) on a small on-disk dataset, as a child process
    (tools/bench_driver.py).  Never `value`: an extra key."""
    import re
    import subprocess

    cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_driver.py"), "--scenes", str(scenes), "--unique",
           str(unique), "--procs", str(procs), "--batch", "32", "--raw-cache"]
    try:
        txt = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT).stdout
        ms = re.findall(r"(\d+) scenes written.*?\(([\d.]+) scenes/s\)", txt)
        if not ms:
            return {"error": txt[-300:]}
        out = {"value": float(ms[0][1]), "unit": "scenes/s", "scenes": int(ms[0][0]),
               "what": "gen_ps driver end to end on disk data (fixed 150k-point scenes, %d loader processes): torch.load "
                       "of the .pth files, host preprocessing, upload, generation, torch.save; tools/bench_driver.py" % procs}
        if len(ms) > 1:
            out["value_raw_cache"] = float(ms[1][1])
            out["raw_cache"] = "the same run over the opt-in raw scene cache (gen_ps --raw_cache: memory-mapped flat files)"
        return out
    except Exception as e:  # noqa: BLE001 - extra key only
        return {"error": repr(e)}


if __name__ == "__main__":
    sys.exit(main())
