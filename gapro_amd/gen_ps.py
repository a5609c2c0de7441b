#!/usr/bin/env python3
"""Offline pseudo-label driver: same-flag mirror of reference gapro/gen_ps.py.

    python -m gapro_amd.gen_ps [--save_folder DIR] [--use_deepfeat] [--deepfeat_folder DIR] [--eval_pslabel]

The four reference flags keep their names, defaults and behaviour (gen_ps.py:15-21); scenes come from
``<data_root>/train/*_inst_nostuff.pth`` in sorted order, finished scenes are skipped (the reference's
resume mechanism, gen_ps.py:39-41) and every scene is written as the same 5-tuple of numpy arrays
(gen_ps.py:126-132) that ISBNet/SPFormer ``torch.load``.  Additive options:

    --data_root DIR        dataset root (default dataset/scannetv2, the reference's relative path)
    --split train|val      which scene list (reference: train)
    --devices 0,1,..       one worker process per listed GPU (independent scenes, no collective); default: this
                           process, cuda:0
    --farm queue|lpt|roundrobin
                           how the workers share the scene list.  queue (default with several devices): a shared
                           work queue -- every worker claims its next batch from the common cost-sorted list
                           (largest file first) through O_EXCL claim files under <save_folder>/.claims.*, so a fast
                           GPU takes more; lpt: static longest-processing-time-first partition by file size;
                           roundrobin: rank r takes scenes r, r+world, .. of the sorted list
    --batch_scenes B       scenes whose GP fits share one launch (default 256: a launch lasts at least as long as its
                           largest fit -- a floor / wall pair of M ~ 1000 takes ~0.3 s on its 32 CUs -- so small batches
                           of real scene mixes leave most of the chip idle: 32 scenes per launch ran at 90 scenes/s,
                           256 at 330; with several workers the claim queue hands out shrinking shares of what is left,
                           at most B at a time)
    --init_mean_std S      std of the random initial variational mean (gpytorch: 1e-3 unseeded;
                           default 0 = deterministic), --seed seeds it
    --broadcast_mu_var     write mu/var at point length (what the released data loaders index)
    --loader_threads T     threads of the library's batch feeder (csrc/feeder.hip) that read scenes from disk two
                           batches ahead, preprocess and upload them, and write the label files (default -1 =
                           min(16, usable CPUs / W - 1) per worker for W workers, at least 2; "usable" honours the
                           cgroup CPU quota, not just the core count the container sees).  Round 5: the whole
                           per-scene host chain -- native file decoding (gapro_pth_*), default features, axis alignment,
                           GT boxes, pinned staging, asynchronous upload; device -> host and the label file on the way
                           out -- runs on these C++ threads; Python moves batches.  (Rounds 1-4: loader processes with a
                           shared-memory hand-over, a memory-mapped raw scene cache, then Python closures on threads:
                           --loader_procs / --raw_cache are still accepted and ignored.)

Exit status: 0 = every scene of the list is written (or was already there); 3 = the run finished but at least one scene
was skipped or failed (no instances, unreadable file, non-finite input, a GP fit that stayed non-positive-definite or
timed out twice): their scan names are listed on stderr, every other scene is written, and a re-run retries only
those; 1 = a worker process died (its claimed-but-unwritten scenes are picked up by a re-run: finished scenes are
skipped by their output files).  With --eval_pslabel the run ends with the reference's `Mean instance iou of pseudo
labels` line (gen_ps.py:133-135) over ALL scenes processed by this run, aggregated across --devices workers.

Differences from the reference, on purpose: output files are written atomically (tmp + rename); a scene
that fails to load is reported and skipped instead of killing the run; a scene without instances (the
reference crashes unpacking None, gen_ps_utils.py:229-230 / gen_ps.py:72) is skipped; a scene with a
non-finite coordinate / feature or a GP fit that stays non-positive-definite after the jitter retries is
reported and skipped while the other scenes of its batch are written; batches are
software-pipelined (the partition of batch i+1 and the merge of batch i-1 run around the GP fits of batch i).
"""
from __future__ import annotations

import argparse
import os
import zlib
import os.path as osp
import sys
import time
from glob import glob

import numpy as np

from . import _lib, pth_io
from .dist_utils import ClaimQueue, effective_cpus, pending_scenes, shard_scenes, shard_scenes_lpt
from .scannet_planes import get_wall_boxes, read_axis_align_matrix


class _LazyModule:
    """`import torch` on first use.  The parent of a `--devices 0,..,7` run only parses arguments, lists the scenes and
    starts the workers; importing torch costs it 1.2 s -- a third of what an eight-GPU job over the ScanNet train split
    then takes.  (The workers import it while their feeder threads are already reading.)"""

    def __init__(self, name):
        self.__dict__["_name"] = name

    def __getattr__(self, attr):
        import importlib

        mod = importlib.import_module(self._name)
        globals()[self._name] = mod  # later uses go straight to the module
        return getattr(mod, attr)


torch = _LazyModule("torch")


class SceneScratch:
    """Grow-only host buffers of one loader thread.  A scene is ~30 MB of short-lived arrays; allocated afresh per
    scene they are mmap'd, page-faulted and unmapped by glibc every time, which 8 .. 16 threads of one process
    serialise on.  read_scene(..., scratch=...) decodes and preprocesses into these instead; the returned arrays are
    views that stay valid until the same thread reads its next scene (the caller uploads / copies them first)."""

    def __init__(self):
        self.bufs = {}

    def get(self, name, shape, dtype):
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        b = self.bufs.get(name)
        if b is None or b.nbytes < n:
            b = self.bufs[name] = np.empty(max(n * 5 // 4, 1 << 16), dtype=np.uint8)
        return b[:n].view(dtype).reshape(shape)


def _load_numpy_file(path, scratch, prefix):
    """torch.load of a NumPy payload through the native reader, into the thread's scratch buffers when given."""
    if pth_io.native_enabled():
        got = pth_io.load_arrays(path, out=(lambda i, shape, dt: scratch.get("%s%d" % (prefix, i), shape, dt))
                                 if scratch is not None else None)
        if got is not None:
            arrays, is_seq = got
            return tuple(arrays) if is_seq else arrays[0]
    return torch.load(path, weights_only=False)


def read_scene(filename, data_root, use_deepfeat=False, deepfeat_folder=None, scratch=None):
    """gen_ps.py:37-69, the disk / host half (thread-safe, no device work): load the scene, build the features
    from UN-aligned xyz, axis-align, read the wall quads.  The scene and superpoint files go through the native reader
    (gapro_pth_*: no unpickling, GIL released) where they are NumPy payloads as prepare_data_inst.py:104 /
    prepare_superpoint.py:27 write them, through torch.load otherwise.  With `scratch` (a SceneScratch owned by the
    calling thread) every large array lives in reused buffers."""
    scan_name = filename.split("/")[-1][:12]
    xyz, rgb, semantic_label, instance_label = _load_numpy_file(filename, scratch, "scene")
    spp = _load_numpy_file(osp.join(data_root, "superpoints", scan_name + ".pth"), scratch, "spp")
    spp = spp.numpy() if isinstance(spp, torch.Tensor) else np.asarray(spp)
    n = xyz.shape[0]
    if use_deepfeat:
        mask_feats = torch.load(osp.join(deepfeat_folder, scan_name + ".pth"), weights_only=False)
        mask_feats = mask_feats.numpy() if isinstance(mask_feats, torch.Tensor) else np.asarray(mask_feats)
        mask_feats = np.asarray(mask_feats, dtype=np.float32)
    elif scratch is not None and xyz.ndim == 2 and xyz.shape[1] == 3 and rgb.shape == xyz.shape:
        # np.concatenate([xyz, rgb], -1).astype(float32) (:55, before the alignment) without the float64 intermediate:
        # the same per-element rounding, written straight into the reused float32 buffer
        mask_feats = scratch.get("feats", (n, 6), np.float32)
        if (xyz.dtype == np.float64 and rgb.dtype == np.float64 and xyz.flags.c_contiguous and rgb.flags.c_contiguous):
            from . import _lib

            _lib.load().gapro_scene_default_feats(xyz.ctypes.data, rgb.ctypes.data, n, mask_feats.ctypes.data)
        else:
            mask_feats[:, 0:3] = xyz
            mask_feats[:, 3:6] = rgb
    else:
        mask_feats = np.asarray(np.concatenate([xyz, rgb], axis=-1), dtype=np.float32)  # :55
    A = read_axis_align_matrix(osp.join(data_root, "scans_transform", scan_name, scan_name + ".txt"))
    if scratch is not None:
        pts = scratch.get("pts", (n, 4), np.float64)
        pts[:, 3] = 1.0
        pts[:, 0:3] = xyz[:, 0:3]
        xyz_al = np.dot(pts, A.transpose(), out=scratch.get("aligned", (n, 4), np.float64))[:, :3]  # :65-69
    else:
        pts = np.ones((n, 4))
        pts[:, 0:3] = xyz[:, 0:3]
        xyz_al = np.dot(pts, A.transpose())[:, :3]  # :65-69
    wall_cls, wall_box, wall_volume = get_wall_boxes(scan_name, data_root=data_root)
    return dict(scan_name=scan_name, coords_float=xyz_al, mask_feats=mask_feats,
                spp=spp.astype(np.int64, copy=False),
                wall_box=np.asarray(wall_box, dtype=np.float32) if len(wall_box) else [],
                wall_box_volume=np.asarray(wall_volume, dtype=np.float32) if len(wall_box) else [],
                semantic_label=semantic_label, instance_label=instance_label)


def add_instance_info(sc, device=None):
    """gen_ps.py:71-77: the GT boxes of a scene read by read_scene; None when the scene has no instance."""
    from .gen_ps_utils import getInstanceInfo, getInstanceInfo_device, getInstanceInfo_native

    if device == "host":  # the driver's loader threads: one native pass on the host arrays, before the upload
        info = getInstanceInfo_native(sc["coords_float"], sc["instance_label"], sc["semantic_label"])
    elif device is not None:  # one pass on the device (gapro_instance_info)
        info = getInstanceInfo_device(sc["coords_float"], sc["instance_label"], sc["semantic_label"], device=device)
    else:  # no device given: the reference's host function, mirrored
        info = getInstanceInfo(sc["coords_float"], instance_label=sc["instance_label"],
                               semantic_label=sc["semantic_label"])
    if info is None:
        return None
    _, instance_cls, instance_box, instance_box_volume, _ = info
    sc.update(instance_cls=np.asarray(instance_cls).astype(np.int64), instance_box=instance_box.astype(np.float32),
              instance_box_volume=instance_box_volume.astype(np.float32))
    return sc


def load_scene(filename, data_root, use_deepfeat=False, deepfeat_folder=None, device=None):
    """gen_ps.py:37-77: load, build features from UN-aligned xyz, axis-align, boxes, wall boxes."""
    return add_instance_info(read_scene(filename, data_root, use_deepfeat, deepfeat_folder), device)


def save_scene(save_path, outs, spp_inv=None, broadcast_mu_var=False):
    """gen_ps.py:126-132, written atomically."""
    sem, ins, prob, mu, var = outs
    if broadcast_mu_var:
        mu, var = mu[spp_inv.long()], var[spp_inv.long()]
    tup = (sem.int().cpu().numpy(), ins.int().cpu().numpy(), prob.cpu().numpy(), mu.cpu().numpy(), var.cpu().numpy())
    write_label_file(save_path, tup)


def write_label_file(save_path, tup):
    """The 5-tuple of NumPy arrays as a torch.load()-able file (gen_ps.py:132), atomically: the native writer
    (gapro_pth_write: no pickling under the GIL; names numpy.core.multiarray, which NumPy 1.x -- the reference's own
    environment -- and 2.x both import), torch.save where it declines (an empty array)."""
    if pth_io.native_enabled() and pth_io.save_arrays(save_path, tup, as_tuple=True):
        return
    tmp = save_path + ".tmp.%d" % os.getpid()
    torch.save(tuple(tup), tmp)
    os.replace(tmp, save_path)


_BLAS_LIMIT = []


def _loader_init():
    """Worker start-up: single-threaded BLAS (torch's own pool is limited by the Worker once torch is imported, beside
    the feeder threads that are already reading).  The wall-box geometry of scannet_planes.py is a few 3 x 3
    inverses per scene on helper threads; a 128-thread OpenBLAS forks and joins over each of them."""
    try:
        from threadpoolctl import threadpool_limits

        _BLAS_LIMIT.append(threadpool_limits(1))  # kept alive: the limit is process-wide until this object is dropped
    except Exception:  # noqa: BLE001 - optional
        pass


def scene_to_device(sc, device):
    """Upload the per-point arrays of a read_scene dict once; add_instance_info and make_job then share them."""
    sc = dict(sc)
    for k, dt in _device_dtypes().items():
        if not isinstance(sc[k], torch.Tensor):
            sc[k] = torch.from_numpy(np.ascontiguousarray(np.asarray(sc[k]))).to(device=device, dtype=dt)
    return sc


def _device_dtypes():
    return {"coords_float": torch.float64, "mask_feats": torch.float32, "spp": torch.int64,
            "semantic_label": torch.float64, "instance_label": torch.float64}


_T_IMPORT = time.time()
_T0_PERF = [0.0]
_EXIT_AFTER_MAIN = [False]  # set by `python -m gapro_amd.gen_ps`: teardown that only returns memory is left to the exit


def _chunks(filenames, args, queue):
    """Batches of scenes still to do (:39-41): from the shared queue, or this worker's static shard."""
    if queue is not None:
        n_workers = max(1, int(getattr(args, "n_workers", 1)))
        while True:
            got = queue.claim(queue.guided(args.batch_scenes, n_workers))
            if not got:
                return
            got = pending_scenes(got, args.save_folder)
            if got:
                yield got
    else:
        pending = pending_scenes(filenames, args.save_folder)
        for i in range(0, len(pending), args.batch_scenes):
            yield pending[i:i + args.batch_scenes]


def _scan_name(fn):
    return fn.split("/")[-1][:12]


class Worker:
    """One GPU of a run (or, with `dry`, its host side alone).

    Round 5: the per-scene host work lives in the library's batch feeder (csrc/feeder.hip: read, default features, axis
    alignment, GT boxes, pinned staging, asynchronous upload; on the way out device -> host and the label file), on
    threads that never take the GIL.  This class only moves BATCHES: it keeps the feeder two batches of file names
    ahead, takes the scenes that are loaded, wraps their slab into SceneJobs, runs them through the software-pipelined
    generator (Pipeline.run_stream) and queues the outputs for writing.  What is left in Python per scene is the wall
    boxes (scannet_planes.py, a handful of quads: helper threads) and ~0.1 ms of tensor views.

    The FIRST batch is whatever has been loaded `first_window` seconds after the start (at least `first_min` scenes):
    the generator starts after ~0.4 s instead of after a full batch, and the reads are a batch ahead from then on."""

    first_window = 0.20
    first_min = 16

    def __init__(self, filenames, args, device_index, dry=False):
        import concurrent.futures as cf

        from .feeder import NativeFeeder

        self.args, self.dry, self.device_index = args, dry, device_index
        self.make_job = None
        self.n_workers = max(1, int(getattr(args, "n_workers", 1)))  # GPU workers sharing this host (--devices)
        n_threads = int(getattr(args, "loader_threads", -1))
        if n_threads <= 0:
            # the CPUs this container may really use (affinity AND cgroup quota: 16 on the boxes of this pool, which show
            # 256 cores), shared by the W workers of the host; one is left to the worker's own Python thread
            n_threads = min(16, max(2, effective_cpus() // self.n_workers - 1))
        self.n_threads = n_threads
        self.filenames = filenames
        self.queue = ClaimQueue(filenames, args.claim_dir) if getattr(args, "claim_dir", None) else None
        self.chunks = _chunks(filenames, args, self.queue)
        self.chunks_done = False
        # scenes this worker will process, when that is known up front (no claim queue): see _take
        self.n_static = len(pending_scenes(filenames, args.save_folder)) if self.queue is None else None
        # Round 6: a worker runs on the library's own arena, streams and events (devmem.NativeBackend) and never imports
        # torch -- 0.75 s of `import torch` + its HIP start-up stood in front of every worker's first read, for a job
        # whose share of an eight-GPU node is 0.64 s of GPU work.  torch is still what the options
        # that do tensor arithmetic in Python need (--eval_pslabel, --broadcast_mu_var, the host-only measurement mode)
        # and what GAPRO_BACKEND=torch selects; then it comes FIRST: libgapro_hip.so binds to the HIP runtime already in
        # the process, and loaded the other way round torch.cuda.is_available() comes up False.
        self.pipe = None
        self.backend = None
        if not dry:
            want = os.environ.get("GAPRO_BACKEND", "").strip().lower()
            needs_torch = bool(args.eval_pslabel or args.broadcast_mu_var or os.environ.get("GAPRO_DRIVER_HOST_ONLY"))
            self.backend = "torch" if (needs_torch or want == "torch") else "native"
            if self.backend == "torch":
                self._torch_threads = torch.get_num_threads()
                torch.set_num_threads(1)  # restored by run(): tests and notebooks call main() in-process
            from .pipeline import Pipeline, make_job

            self.make_job = make_job
            self.pipe = Pipeline(device=device_index, training_iter=50, init_mean_std=args.init_mean_std, seed=args.seed,
                                 backend=self.backend)
            self.pipe.strict = False  # a scene that cannot be processed is reported and skipped, the rest is written
            if os.environ.get("GAPRO_DRIVER_TIMES"):
                self.pipe.trace = []  # host-side stage timeline of the pipeline (printed at the end)
            self.dev = self.pipe.device
        # pinned staging: scenes loaded but not yet uploaded (two batches of ~11 MB scenes) plus label files in flight
        budget = int(os.environ.get("GAPRO_FEED_BUDGET_MB", "8192")) << 20
        self.feeder = NativeFeeder(-1 if dry else device_index, n_threads, budget)
        self.walls = {}  # file name -> future of get_wall_boxes
        self.wall_pool = cf.ThreadPoolExecutor(max_workers=2)
        self.submitted = 0
        self.meta = []  # per yielded batch: (scene dicts, jobs, slab)
        self.keep = []  # (exported count after this batch, objects whose device memory the writers still read)
        self.done = self.failed = 0
        self.failed_names = []
        self.miou = {}
        self.out_folder = args.save_folder
        self.t_first = None
        self.n_first = 0
        self._submit_more()  # the feeder threads start reading now, before run() is called

    # ---- input side -------------------------------------------------------------------------------------------
    def _prealloc(self):
        """The fit workspace is allocated NOW, by a helper thread, while the first scenes are read: its size is only
        known after the first batch has been scheduled, and the hipMalloc + clear of ~20 GB then sits in front of the
        first launch.  Estimate: ~330 bytes of workspace per point of a batch of ScanNet-like scenes with the usual
        30 % headroom, ~91 bytes of scene file per point, clamped to a third of the free device memory; a batch that
        needs more grows it as before."""
        import threading

        try:
            sizes = sorted((os.path.getsize(fn) for fn in self.filenames), reverse=True)
        except OSError:
            return
        if not sizes or os.environ.get("GAPRO_NO_PREALLOC"):
            return
        nb = max(1, min(int(self.args.batch_scenes), len(sizes)))
        batch_bytes = sum(sizes[:nb]) if self.queue is not None else 1.2 * nb * sum(sizes) / len(sizes)
        est = int(batch_bytes / 91.0 * 330.0 * 1.3)
        try:
            free = self.pipe.be.mem_get_info()[0]
        except Exception:  # noqa: BLE001
            free = 64 << 30
        est = min(est, 64 << 30, free // 3)
        if est <= (1 << 28):
            return

        def work():
            try:
                self.pipe.prealloc_workspace(est)
            except Exception as e:  # noqa: BLE001 - an estimate: the first launch allocates what it needs
                print("[gen_ps] workspace pre-allocation skipped: %r" % (e,), file=sys.stderr)

        self._prealloc_thread = threading.Thread(target=work, daemon=True)
        self._prealloc_thread.start()

    def _submit_more(self):
        """Keep the feeder two batches of file names ahead of what has been taken."""
        a = self.args
        while not self.chunks_done and self.submitted - self.feeder.taken < 2 * a.batch_scenes:
            c = next(self.chunks, None)
            if not c:
                self.chunks_done = True
                self.feeder.close()
                break
            self.feeder.submit(c, a.data_root, a.use_deepfeat, a.deepfeat_folder)
            for fn in c:
                self.walls[fn] = self.wall_pool.submit(get_wall_boxes, _scan_name(fn), a.data_root)
            self.submitted += len(c)

    def _take(self, first):
        """(records, slab, batch id) of the next batch, or None when the list is exhausted."""
        a = self.args
        self._submit_more()
        left = self.submitted - self.feeder.taken
        if left <= 0:
            return None
        want = min(a.batch_scenes, left)
        cap = a.batch_scenes
        if self.n_static is not None and not first:
            # a static list (one worker, or a static shard): what is left is known, so it goes out in EQUAL batches -- the
            # first batch is whatever was loaded after 0.2 s, and fixed 256s behind it end in a launch of a dozen scenes
            # that lasts as long as its largest fit (0.3 s of a 4 s job)
            rem = max(self.n_static - self.feeder.taken, 1)
            nb = -(-rem // a.batch_scenes)
            want = cap = max(1, min(-(-rem // nb), left))
        if first:
            # (a static list that fits ONE batch is waited for as a whole, within reason: two launches of 120 and 30
            # scenes last twice as long as one of 150 -- a launch is as long as its largest fit)
            window = max(self.first_window, 0.6) if (self.n_static is not None and self.n_static <= a.batch_scenes) \
                else self.first_window
            wait_ms = int(1000 * max(0.0, window - (time.time() - self.t0)))
            n, nbytes = self.feeder.poll(want, a.batch_scenes, wait_ms)
            if n < min(self.first_min, want):
                n, nbytes = self.feeder.poll(min(self.first_min, want), a.batch_scenes, -1)
        else:
            n, nbytes = self.feeder.poll(want, cap, -1)
        if n <= 0:
            return None
        if self.dry:
            bid, recs = self.feeder.upload(n, 0, 0)
            return recs, None, bid
        be = self.pipe.be
        slab = be.empty(max(nbytes, 256))
        # the slab comes from a caching allocator (torch's, or the library's arena) on this stream: the feed's copies (its
        # own stream) wait for what is queued here, e.g. the kernels of the block's previous owner
        cur = be.current_stream().cuda_stream
        bid, recs = self.feeder.upload(n, slab.data_ptr(), nbytes, cur)
        self.feeder.batch_wait(bid, cur)
        return recs, slab, bid

    def _fail(self, fn, why):
        print("[gen_ps] %s: %s" % (fn, why), file=sys.stderr)
        self.failed += 1
        self.failed_names.append(_scan_name(fn))

    def _walls(self, fn):
        fut = self.walls.pop(fn, None)
        _, wall_box, wall_vol = fut.result() if fut is not None else get_wall_boxes(_scan_name(fn), self.args.data_root)
        if len(wall_box):
            return np.asarray(wall_box, dtype=np.float32), np.asarray(wall_vol, dtype=np.float32)
        return [], []

    def _job_from_slab(self, r, slab):
        n, d = r.n_points, r.feat_dim
        oc, of, os_, osem, oinst = r.off

        def view(off, nbytes, dtype, shape):
            return slab[off:off + nbytes].view(dtype).reshape(shape)

        be = self.pipe.be
        sc = dict(scan_name=_scan_name(r.filename), coords_float=view(oc, 24 * n, be.f64, (n, 3)),
                  mask_feats=view(of, 4 * d * n, be.f32, (n, d)), spp=view(os_, 8 * n, be.i64, (n,)),
                  semantic_label=view(osem, 8 * n, be.f64, (n,)),
                  instance_label=view(oinst, 8 * n, be.f64, (n,)))
        wall_box, wall_vol = self._walls(r.filename)
        sc["job"] = self.make_job(sc["coords_float"], sc["mask_feats"], sc["spp"], r.instance_cls, r.instance_box,
                             r.instance_box_volume, wall_box, wall_vol, instance_classes=18, ground_h=0.1,
                             thresh_spp_occu=0.999, device=self.dev,  # :106-110
                             scene_key=zlib.crc32(sc["scan_name"].encode()), backend=be)
        return sc

    def _job_fallback(self, fn):
        """A scene the native reader does not handle (tensor payloads, unusual dtypes): the reference's own loaders."""
        a = self.args
        self.walls.pop(fn, None)
        sc = load_scene(fn, a.data_root, a.use_deepfeat, a.deepfeat_folder)
        if sc is None:
            return None
        if self.backend == "torch":
            sc = scene_to_device(sc, self.dev)
        # (native backend: make_job uploads the host arrays itself; the file was read with CPU torch, which is all that
        # the reference's own loaders need)
        sc["job"] = self.make_job(sc["coords_float"], sc["mask_feats"], sc["spp"], sc["instance_cls"], sc["instance_box"],
                             sc["instance_box_volume"], sc["wall_box"], sc["wall_box_volume"], instance_classes=18,
                             ground_h=0.1, thresh_spp_occu=0.999, device=self.dev,
                             scene_key=zlib.crc32(sc["scan_name"].encode()), backend=self.pipe.be)
        return sc

    def batches(self):
        first = True
        while True:
            t = time.time()
            got = self._take(first)
            self.spent["wait"] += time.time() - t
            if got is None:
                return
            first = False
            recs, slab, bid = got
            t = time.time()
            scenes = []
            for r in recs:
                try:
                    if r.status == _lib.GAPRO_ERR_UNSUPPORTED:
                        sc = self._job_fallback(r.filename)
                    elif r.status != 0:
                        self._fail(r.filename, "load failed: %s" % _lib.STATUS_NAMES.get(r.status, r.status))
                        continue
                    else:
                        sc = self._job_from_slab(r, slab) if r.n_instances > 0 else None
                    if sc is None:
                        self._fail(r.filename, "no instances, skipped")
                        continue
                    scenes.append(sc)
                except Exception as e:  # noqa: BLE001 - one bad scene must not kill the run
                    self._fail(r.filename, "load failed: %r" % (e,))
            self.spent["jobs"] += time.time() - t
            self.feeder.release_batch(bid)
            if not scenes:
                continue
            if self.pipe.trace is not None:
                self.pipe.trace.append((time.perf_counter(), -1, "%d scenes of a batch taken" % len(scenes)))
            self.meta.append((scenes, [s["job"] for s in scenes], slab))
            yield self.meta[-1][1]

    # ---- output side ------------------------------------------------------------------------------------------
    def _host_only_stream(self, batch_iter):
        """GAPRO_DRIVER_HOST_ONLY=1 (tools/bench_driver.py --host-only): everything the host does for a scene with the
        generation replaced by zero outputs of the right shapes; never a product mode."""
        dev = self.dev
        for jobs in batch_iter:
            outs = []
            for j in jobs:
                n = int(j.coords.shape[0])
                outs.append((torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev),
                             torch.ones(n, dtype=torch.float32, device=dev),
                             torch.full((max(1, n // 50),), -100.0, dtype=torch.float32, device=dev),
                             torch.full((max(1, n // 50),), -100.0, dtype=torch.float32, device=dev)))
            yield outs

    def _export(self, scenes, jobs, outs, slab):
        a = self.args
        be = self.pipe.be
        ready = be.current_stream().record_event()  # run_stream ordered the outputs on this stream
        items, alive = [], [outs, slab, ready]
        for s, job, o in zip(scenes, jobs, outs):
            if o is None:  # Pipeline.strict = False: this scene could not be processed, the others could
                print("[gen_ps] warning: %s skipped: %s" % (s["scan_name"], job.error), file=sys.stderr)
                self.failed += 1
                self.failed_names.append(s["scan_name"])
                continue
            if a.eval_pslabel:
                from .eval_ps_labels import get_miou_scene

                sem_gt = s["semantic_label"].int()
                ins_gt = s["instance_label"].int()
                sem_gt[sem_gt != -100] -= 2  # :119-120
                sem_gt[(sem_gt == -1) | (sem_gt == -2)] = 18
                ious = get_miou_scene(sem_gt.long(), ins_gt.long(), o[0].long(), o[1].long())
                print("miou", ious)
                self.miou[s["scan_name"]] = ious.float().cpu().numpy()  # :125 ious_arr.append(ious)
            sem, ins, prob, mu, var = o
            if sem.dtype != be.i32 or ins.dtype != be.i32:
                sem, ins = sem.int(), ins.int()
                alive += [sem, ins]
            if a.broadcast_mu_var:
                inv = job.spp_inv.long()
                mu, var = mu[inv], var[inv]
                alive += [mu, var]
            items.append((osp.join(self.out_folder, s["scan_name"] + ".pth"), sem.data_ptr(), ins.data_ptr(),
                          prob.data_ptr(), mu.data_ptr(), var.data_ptr(), sem.numel(), mu.numel()))
            self.done += 1
        if a.eval_pslabel or a.broadcast_mu_var:
            ready = be.current_stream().record_event()
            alive.append(ready)
        self.feeder.export(items, ready.cuda_event)
        self.keep.append((self.feeder.exported, alive))
        done_now, _ = self.feeder.export_wait(0, 0)  # (no wait: just the count) -- drop what the writers are done with
        while self.keep and self.keep[0][0] <= done_now:
            self.keep.pop(0)

    def _run_dry(self):
        """`--dry_run`: the host side alone -- same list / claim queue, same native loaders and writer, all-zero
        stand-in outputs of the right shapes into <save_folder>.DRY_RUN.  Nothing here is a pseudo-label."""
        first = True
        while True:
            got = self._take(first)
            if got is None:
                break
            first = False
            recs, _, bid = got
            items, alive = [], []
            for r in recs:
                self.walls.pop(r.filename, None)
                if r.status != 0 or r.n_instances <= 0:
                    self._fail(r.filename, "load failed or no instances (status %d)" % r.status)
                    continue
                n = r.n_points
                s = max(1, n // 50)
                arrs = (np.zeros(n, np.int32), np.zeros(n, np.int32), np.ones(n, np.float32),
                        np.full(s, -100.0, np.float32), np.full(s, -100.0, np.float32))
                alive.append(arrs)
                items.append((osp.join(self.out_folder, _scan_name(r.filename) + ".pth"),) +
                             tuple(x.ctypes.data for x in arrs) + (n, s))
                self.done += 1
            self.feeder.release_batch(bid)
            self.feeder.export(items)
            self.keep.append((self.feeder.exported, alive))
            if self.t_first is None:
                self.t_first, self.n_first = time.time(), len(items)
            done_now, _ = self.feeder.export_wait(0, 0)
            while self.keep and self.keep[0][0] <= done_now:
                self.keep.pop(0)

    def run(self):
        a = self.args
        native = True
        host_only = bool(os.environ.get("GAPRO_DRIVER_HOST_ONLY")) and not self.dry
        if self.dry:
            self.out_folder = osp.normpath(a.save_folder) + ".DRY_RUN"
            print("[gen_ps] WARNING: --dry_run: NO pseudo-labels are generated; all-zero stand-in outputs go to %s"
                  % self.out_folder, file=sys.stderr)
        elif host_only:  # a measurement mode must never leave files that look like pseudo-labels
            self.out_folder = osp.normpath(a.save_folder) + ".HOST_ONLY_MEASUREMENT"
            print("[gen_ps] WARNING: GAPRO_DRIVER_HOST_ONLY is set: NO pseudo-labels are generated; the all-zero "
                  "stand-in outputs go to %s, never to --save_folder" % self.out_folder, file=sys.stderr)
        os.makedirs(self.out_folder, exist_ok=True)
        self.spent = dict(wait=0.0, jobs=0.0, export=0.0)  # main-thread seconds, GAPRO_DRIVER_TIMES=1
        _T0_PERF[0] = time.perf_counter()
        self.t0 = t0 = time.time()
        try:
            if self.dry:
                self._run_dry()
            else:
                self._prealloc()
                # (Pipeline.warmup() here -- a small launch through every kernel while the first batch loads -- was tried:
                # it queues behind the workspace allocation, whose hipMalloc in turn holds up the loaders' pinned
                # allocations: first batch of 16 scenes at 0.72 s instead of 185 at 0.28 s.  GAPRO_WARMUP=1 for A/B.)
                if os.environ.get("GAPRO_WARMUP"):
                    self.pipe.warmup()
                stream = self._host_only_stream(self.batches()) if host_only else self.pipe.run_stream(self.batches())
                for outs in stream:
                    if self.t_first is None:
                        self.t_first, self.n_first = time.time(), len(outs)
                    scenes, jobs, slab = self.meta.pop(0)
                    t = time.time()
                    self._export(scenes, jobs, outs, slab)
                    self.spent["export"] += time.time() - t
            n_done, n_failed = self.feeder.export_wait(-1, -1)
            t_written = time.time()  # the last label file is on disk: the job is done, what follows is teardown
            for msg in self.feeder.export_errors(n_failed):
                print("[gen_ps] a label file could not be written: %s" % msg, file=sys.stderr)
                self.failed += 1
                self.done -= 1
                self.failed_names.append("<write failed: %s>" % msg)
        finally:
            # The feeder's threads stop FIRST (a writer finishes the file it is on; on the error path the exports still
            # queued are dropped -- a restart writes them, skip-if-exists), and only then may the device arrays and the
            # event they read go back to the allocator (the other order freed memory under a running copy).
            self.wall_pool.shutdown(wait=False)
            th = getattr(self, "_prealloc_thread", None)
            if th is not None:
                th.join()
            t_fin = time.time()
            if os.environ.get("GAPRO_DRIVER_TIMES"):
                try:
                    print("[gen_ps] feeder threads: %r" % (self.feeder.stats(),))
                except Exception:  # noqa: BLE001
                    pass
            self.feeder.destroy(process_is_exiting=_EXIT_AFTER_MAIN[0])
            self.keep = []
            if getattr(self, "_torch_threads", None):
                torch.set_num_threads(self._torch_threads)
        dt = t_written - t0
        if os.environ.get("GAPRO_DRIVER_TIMES"):
            print("[gen_ps] teardown after the last file: %.2f s (%.2f s of it the feeder's pinned memory)"
                  % (time.time() - t_written, time.time() - t_fin))
        dev_i, done, failed = self.device_index, self.done, self.failed
        print("[gen_ps] device %d: %d scenes written, %d skipped/failed, %.3f s (%.2f scenes/s)"
              % (dev_i, done, failed, dt, done / dt if dt > 0 else 0.0))
        # start-up is reported apart from the rate: at ~300 scenes/s a 1201-scene split is a few seconds of work, and
        # the interpreter / library loads of a worker take a noticeable part of that
        t_first = self.t_first
        print("[gen_ps] device %d: start-up %.1f s (process start -> generator ready), first batch out after %.1f s more, "
              "%d loader threads, %d loader processes, %s file I/O, %s%s"
              % (dev_i, t0 - _T_IMPORT, (t_first - t0) if t_first else 0.0, self.n_threads, 0,
                 "native feeder (gapro_feed_*)" if native else "torch.load / torch.save",
                 ("device plumbing: %s" % ("the library's own arena, streams and events" if self.backend == "native"
                                           else "torch")) if self.backend else "no device",
                 ", dry run (no GPU)" if self.dry else (", host-only measurement mode" if host_only else "")))
        t_end = t0 + dt
        if t_first is not None and done > self.n_first and t_end > t_first:
            print("[gen_ps] device %d: steady state %.2f scenes/s (%d scenes in the %.2f s after the first batch of %d)"
                  % (dev_i, (done - self.n_first) / (t_end - t_first), done - self.n_first, t_end - t_first, self.n_first))
        if os.environ.get("GAPRO_DRIVER_TIMES") and self.pipe is not None:
            print("[gen_ps] main-thread seconds: " + ", ".join("%s %.2f" % kv for kv in self.spent.items()))
            ids = {}
            for t, bid, name in (self.pipe.trace or [])[:60]:
                print("[gen_ps]   %8.3f s  batch %d  %s" % (t - _T0_PERF[0], ids.setdefault(bid, len(ids)), name))
        result = dict(done=done, failed=sorted(self.failed_names),
                      miou={k: [float(x) for x in v] for k, v in self.miou.items()}, seconds=dt,
                      startup_seconds=t0 - _T_IMPORT, first_batch_seconds=(t_first - t0) if t_first else 0.0,
                      timeout_retries=int(getattr(self.pipe, "timeout_retries", 0)) if self.pipe is not None else 0)
        if self.dry:
            result["dry_run"] = True
        job_dir = getattr(a, "job_dir", None)
        if job_dir:  # several workers: the parent aggregates (mean IoU over all scenes, failed scans, exit status)
            import json

            tmp = osp.join(job_dir, "result.%d.json.tmp" % device_index_rank(a))
            with open(tmp, "w") as fh:
                json.dump(result, fh)
            os.replace(tmp, tmp[:-4])
        return result


def run_worker(filenames, args, device_index):
    return Worker(filenames, args, device_index).run()


def run_worker_dry(filenames, args, rank):
    return Worker(filenames, args, rank, dry=True).run()


def device_index_rank(args):
    return max(int(getattr(args, "worker_rank", -1)), 0)


def mean_instance_iou(miou_by_scan):
    """gen_ps.py:133-135: torch.mean over the per-instance IoUs of every scene, concatenated in the (sorted) scene
    order the reference iterates in; float32 as there.  None when no scene was evaluated."""
    arrs = [torch.as_tensor(np.asarray(miou_by_scan[k], dtype=np.float32)) for k in sorted(miou_by_scan)]
    arrs = [a for a in arrs if a.numel()]
    if not arrs:
        return None
    return torch.mean(torch.cat(arrs, dim=0)).item()


def finish_run(args, results, crashed=()):
    """End of a run (one worker or the parent of several): the reference's summary line, the list of scenes that were
    not written, the exit status (0 / 3 / 1: see the module docstring)."""
    miou, failed = {}, []
    for r in results:
        miou.update(r.get("miou", {}))
        failed += list(r.get("failed", []))
    if args.eval_pslabel:
        m = mean_instance_iou(miou)
        if m is not None:
            print("Mean instance iou of pseudo labels", m)  # :135
    if crashed:
        print("[gen_ps] worker(s) %s died: the scenes they had claimed but not written are NOT done; run the same "
              "command again -- finished scenes are skipped by their output files (a fresh claim directory is used per "
              "run)" % ", ".join(str(c) for c in crashed), file=sys.stderr)
    if failed:
        print("[gen_ps] %d scene(s) skipped or failed, not written: %s" % (len(failed), " ".join(sorted(failed))),
              file=sys.stderr)
    print("Finish")
    return 1 if crashed else (3 if failed else 0)


def main(argv=None):
    parser = argparse.ArgumentParser("GaPro_GenPS")
    parser.add_argument("--save_folder", type=str, default="dataset/scannetv2/gaussian_process_kl_pseudo_labels")
    parser.add_argument("--use_deepfeat", action="store_true")
    parser.add_argument("--deepfeat_folder", type=str, default="dataset/scannetv2/pretrain_maskfeats2")
    parser.add_argument("--eval_pslabel", action="store_true")
    # additive options
    parser.add_argument("--data_root", type=str, default="dataset/scannetv2")
    parser.add_argument("--split", type=str, default="train", choices=["train", "val"])
    parser.add_argument("--devices", type=str, default="0")
    parser.add_argument("--batch_scenes", type=int, default=256)
    parser.add_argument("--init_mean_std", type=float, default=0.0)
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--broadcast_mu_var", action="store_true")
    parser.add_argument("--raw_cache", type=str, default=None, help=argparse.SUPPRESS)  # ignored since round 5
    parser.add_argument("--loader_threads", type=int, default=-1)
    parser.add_argument("--loader_procs", type=int, default=-1, help=argparse.SUPPRESS)  # ignored since round 5
    parser.add_argument("--farm", type=str, default="queue", choices=["queue", "lpt", "roundrobin"])
    parser.add_argument("--worker_rank", type=int, default=-1, help=argparse.SUPPRESS)
    parser.add_argument("--claim_dir", type=str, default=None, help=argparse.SUPPRESS)
    parser.add_argument("--job_dir", type=str, default=None, help=argparse.SUPPRESS)
    parser.add_argument("--dry_run", action="store_true", help=argparse.SUPPRESS)
    args = parser.parse_args(argv)

    os.makedirs(args.save_folder, exist_ok=True)
    filenames = sorted(glob(osp.join(args.data_root, args.split, "*_inst_nostuff.pth")))  # :27-32
    devices = [int(d) for d in args.devices.split(",") if d != ""]
    args.n_workers = len(devices)
    if args.raw_cache not in (None, "none", "None", "") or args.loader_procs > 0:
        # rounds 1-4 worked around Python's unpickling with loader processes and a memory-mapped raw scene cache; with the
        # native feeder neither has a use (the cache silently wrote a second copy of the dataset)
        print("[gen_ps] note: --raw_cache / --loader_procs are accepted for old command lines and ignored: scenes are "
              "read by the library's native feeder threads", file=sys.stderr)
    if args.worker_rank >= 0 or len(devices) == 1:
        r = max(args.worker_rank, 0)
        # independent scenes, no collective: the shared queue hands out the common list; the static farms shard it
        if args.claim_dir or len(devices) == 1:
            mine = filenames
        elif args.farm == "roundrobin":
            mine = shard_scenes(filenames, r, len(devices))
        else:  # "lpt", or "queue" without a claim directory (a worker started by hand)
            mine = shard_scenes_lpt(filenames, r, len(devices))
        # the loader threads of this process call BLAS concurrently: one BLAS / torch thread each while the worker runs
        # (restored afterwards: tests and notebooks call main() in-process)
        _loader_init()
        try:
            result = run_worker_dry(mine, args, r) if args.dry_run else run_worker(mine, args, devices[r])
        finally:
            while _BLAS_LIMIT:
                lim = _BLAS_LIMIT.pop()
                try:
                    lim.restore_original_limits()
                except Exception:  # noqa: BLE001 - older threadpoolctl
                    pass
        if args.worker_rank >= 0 and args.job_dir:  # a child of the farm: the parent prints the summary
            return 3 if result["failed"] else 0
        return finish_run(args, [result])
    import json
    import shutil
    import subprocess

    # a fresh job directory per run: the workers' claim files (queue farm) and their result files (mean IoU, failed
    # scans); the OUTPUT files are what a restart skips
    job_dir = osp.join(args.save_folder, ".job.%d.%d" % (os.getpid(), int(time.time())))
    os.makedirs(job_dir)
    extra = ["--job_dir", job_dir]
    if args.farm == "queue":
        claim_dir = osp.join(job_dir, "claims")
        os.makedirs(claim_dir)
        extra += ["--claim_dir", claim_dir]
    procs, results, crashed = [], [], []
    try:
        for r in range(len(devices)):
            cmd = [sys.executable, "-m", "gapro_amd.gen_ps"] + (argv if argv is not None else sys.argv[1:]) + \
                  ["--worker_rank", str(r)] + extra
            procs.append(subprocess.Popen(cmd))
        rc = [p.wait() for p in procs]
        for r, code in enumerate(rc):
            path = osp.join(job_dir, "result.%d.json" % r)
            if code in (0, 3) and osp.exists(path):
                with open(path) as fh:
                    results.append(json.load(fh))
            else:
                crashed.append("%d (device %d, exit status %r)" % (r, devices[r], code))
    finally:
        shutil.rmtree(job_dir, ignore_errors=True)
    return finish_run(args, results, crashed)


if __name__ == "__main__":
    _EXIT_AFTER_MAIN[0] = True
    _rc = main()
    # Every label file is on disk and renamed, every result file written: what is left is handing 20-odd GB of device memory,
    # the pinned pool and the HIP runtime back piece by piece -- 0.3 .. 0.4 s of a worker whose share of an eight-GPU job is
    # 0.7 s of work (tools/share_probe.py).  The process ends here and the driver reclaims everything at once.
    if os.environ.get("GAPRO_T0"):  # tools/share_probe.py: the parent's clock at the moment it started this process
        print("[gen_ps] since the parent started this process: modules imported at %.3f s, leaving at %.3f s"
              % (_T_IMPORT - float(os.environ["GAPRO_T0"]), time.time() - float(os.environ["GAPRO_T0"])))
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(int(_rc or 0))
