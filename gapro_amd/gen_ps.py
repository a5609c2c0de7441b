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
    --raw_cache DIR        raw scene cache (opt-in with one device; ON by default, at <save_folder>.raw_cache, with
                           several: unpickling caps the whole host below what ONE GPU takes, `--raw_cache none`
                           refuses): the first run writes, next to nothing else, one flat file per scene
                           (<DIR>/<scan>.gaproraw: the arrays of read_scene exactly as they are uploaded, 64-byte
                           aligned, with the sizes and mtimes of the source files in its header); later runs map it
                           (np.memmap) and upload straight from the page cache -- no unpickling, no loader process,
                           no shared-memory copy.  A cache whose source files changed is rebuilt.  Unpickling the
                           ScanNet .pth tuples is what caps the loaders at ~350 scenes/s per host
    --loader_threads T     threads that read scenes from disk two batches ahead, upload them and write the results
                           (default -1 = min(16, physical cores / (2 W)) per worker for W workers, at least 4).
                           Round 4: the scene and label files go through the native reader / writer (gapro_pth_*:
                           no unpickling, no pickling, GIL released), so plain threads of the worker process feed a
                           GPU and NO loader process is started by default.
    --loader_procs P       read and write in P loader PROCESSES instead (default -1 = none with the native reader; the
                           round-1..3 pool of min(16, physical cores / (2 W)) processes when GAPRO_NATIVE_PTH=0:
                           unpickling a ScanNet .pth holds the GIL, so threads top out near one core; processes hand
                           the arrays over in POSIX shared memory).  The pool is started before the worker touches
                           the GPU.

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
import torch

from . import pth_io
from .dist_utils import ClaimQueue, pending_scenes, shard_scenes, shard_scenes_lpt
from .gen_ps_utils import getInstanceInfo, getInstanceInfo_device, getInstanceInfo_native
from .pipeline import Pipeline, make_job
from .scannet_planes import get_wall_boxes, read_axis_align_matrix


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


_SHM_KEYS = ("coords_float", "mask_feats", "spp", "semantic_label", "instance_label", "wall_box", "wall_box_volume")
_INFO_KEYS = ("instance_cls", "instance_box", "instance_box_volume")


_RAW_MAGIC = b"GAPRORAW1\n"


def _source_stamp(filename, data_root, use_deepfeat=False, deepfeat_folder=None):
    """(size, mtime_ns) of every file read_scene reads for this scene: a cache written from other bytes is stale."""
    scan_name = filename.split("/")[-1][:12]
    files = [filename, osp.join(data_root, "superpoints", scan_name + ".pth"),
             osp.join(data_root, "scans_transform", scan_name, scan_name + ".txt"),
             osp.join(data_root, "scannet_planes", scan_name + ".json")]
    if use_deepfeat:
        files.append(osp.join(deepfeat_folder, scan_name + ".pth"))
    stamp = []
    for fn in files:
        try:
            st = os.stat(fn)
            stamp.append([os.path.basename(fn), int(st.st_size), int(st.st_mtime_ns)])
        except OSError:
            stamp.append([os.path.basename(fn), -1, -1])
    return stamp


def raw_cache_path(cache_dir, filename):
    return osp.join(cache_dir, filename.split("/")[-1][:12] + ".gaproraw")


def write_raw_cache(path, sc, stamp):
    """One flat file: magic | u64 header length | JSON header | arrays (64-byte aligned), written atomically."""
    import json

    arrs = {k: np.ascontiguousarray(np.asarray(sc[k])) for k in _SHM_KEYS}
    fields, off = [], 0
    for k in _SHM_KEYS:
        a = arrs[k]
        fields.append([k, a.dtype.str, list(a.shape), off])
        off += (a.nbytes + 63) // 64 * 64
    header = json.dumps(dict(scan_name=sc["scan_name"], stamp=stamp, fields=fields, nbytes=off)).encode()
    pre = len(_RAW_MAGIC) + 8 + len(header)
    pad = (-pre) % 64
    tmp = path + ".tmp.%d" % os.getpid()
    with open(tmp, "wb") as fh:
        fh.write(_RAW_MAGIC)
        fh.write(np.uint64(len(header) + pad).tobytes())
        fh.write(header + b" " * pad)
        for (k, _, _, o) in fields:
            a = arrs[k]
            fh.write(a.tobytes())
            fh.write(b"\0" * ((-a.nbytes) % 64))
    os.replace(tmp, path)


def read_raw_cache(path, stamp):
    """The cached scene as a dict of read-only memory-mapped arrays, or None (absent, stale or damaged)."""
    import json

    try:
        with open(path, "rb") as fh:
            if fh.read(len(_RAW_MAGIC)) != _RAW_MAGIC:
                return None
            hlen = int(np.frombuffer(fh.read(8), dtype=np.uint64)[0])
            header = json.loads(fh.read(hlen).decode())
        base = len(_RAW_MAGIC) + 8 + hlen
        if header["stamp"] != stamp or os.path.getsize(path) < base + header["nbytes"]:
            return None
        sc = dict(scan_name=header["scan_name"])
        for (k, dt, shape, off) in header["fields"]:
            n = int(np.prod(shape)) if len(shape) else 1
            sc[k] = np.memmap(path, dtype=np.dtype(dt), mode="r", offset=base + off, shape=tuple(shape)) if n else []
        return sc
    except (OSError, ValueError, KeyError):
        return None


_BLAS_LIMIT = []


def _loader_init():
    """Loader process / worker start-up: single-threaded BLAS / torch (the pool is the parallelism).  Not a nicety: the
    axis alignment of read_scene is a [N, 4] x [4, 4] product, which an 8-thread OpenBLAS takes 66 ms for (fork-join
    over a 1 ms job; worse with the 128+ threads of the GPU host) against 1.1 ms on one thread -- with the same bits."""
    torch.set_num_threads(1)
    try:
        from threadpoolctl import threadpool_limits

        _BLAS_LIMIT.append(threadpool_limits(1))  # kept alive: the limit is process-wide until this object is dropped
    except Exception:  # noqa: BLE001 - optional
        pass
    # A scene is ~30 MB of short-lived NumPy arrays (decoded tuple, features, aligned coordinates).  glibc serves those
    # with mmap / munmap or trims the heap after every free: page faults and address-space locking that 8 .. 16 loader
    # threads of one process serialise on (8 threads: 134 -> 205 scenes/s of host work with the heap kept).  Keep freed
    # memory in the process: M_MMAP_THRESHOLD (-3) at its 32 MiB maximum, M_TRIM_THRESHOLD (-1) and M_TOP_PAD (-2) large.
    try:
        import ctypes

        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-3, 32 << 20)
        libc.mallopt(-1, 0x7FFFFFFF)
        libc.mallopt(-2, 256 << 20)
    except Exception:  # noqa: BLE001 - not glibc
        pass


def _read_scene_shm(filename, data_root, use_deepfeat=False, deepfeat_folder=None, raw_cache=None):
    """Loader process: read_scene, then the arrays go into ONE POSIX shared-memory block (64-byte aligned
    fields) and only its name and layout travel back through the pipe.  With a raw cache directory the scene is
    also written there, so that the next run maps it instead of unpickling."""
    from multiprocessing import shared_memory

    sc = read_scene(filename, data_root, use_deepfeat, deepfeat_folder)
    if raw_cache:
        try:
            write_raw_cache(raw_cache_path(raw_cache, filename), sc,
                            _source_stamp(filename, data_root, use_deepfeat, deepfeat_folder))
        except OSError as e:  # a full disk must not lose the scene
            print("[gen_ps] raw cache not written for %s: %r" % (filename, e), file=sys.stderr)
    arrs = {k: np.ascontiguousarray(np.asarray(sc[k])) for k in _SHM_KEYS}
    # GT boxes here, on the host arrays (see read_upload in run_worker); None = a scene without instances
    info = add_instance_info(dict(sc, coords_float=arrs["coords_float"]), "host")
    info = {k: info[k] for k in _INFO_KEYS} if info is not None else None
    layout, off = [], 0
    for k in _SHM_KEYS:
        a = arrs[k]
        layout.append((k, a.dtype.str, a.shape, off))
        off += (a.nbytes + 63) // 64 * 64
    try:
        shm = shared_memory.SharedMemory(create=True, size=max(off, 64))
    except OSError:
        return dict(scan_name=sc["scan_name"], shm=None, arrays=arrs, info=info)
    try:
        try:  # reserve the pages now: a full /dev/shm must be an error here, not a SIGBUS in the copy below
            os.posix_fallocate(shm._fd, 0, max(off, 64))
        except OSError:  # no room (container with a small /dev/shm): this scene travels through the pipe instead
            shm.unlink()
            return dict(scan_name=sc["scan_name"], shm=None, arrays=arrs, info=info)
        for (k, _, _, o) in layout:
            a = arrs[k]
            if a.nbytes:
                np.ndarray(a.shape, a.dtype, buffer=shm.buf, offset=o)[...] = a
        name = shm.name
    finally:
        shm.close()
    return dict(scan_name=sc["scan_name"], shm=name, layout=layout, info=info)


def _scene_from_shm(msg, device, stager=None):
    """Worker side of _read_scene_shm: map the block, upload the per-point arrays from it (through the thread's
    pinned stager when given), keep the tiny wall arrays, release the block."""
    from multiprocessing import shared_memory

    if msg["shm"] is None:  # the pipe fallback of _read_scene_shm
        sc = dict(msg["arrays"], scan_name=msg["scan_name"])
        for k in ("wall_box", "wall_box_volume"):
            if not len(sc[k]):
                sc[k] = []
        return scene_to_device(sc, device) if device is not None else sc
    shm = shared_memory.SharedMemory(name=msg["shm"])
    sc = dict(scan_name=msg["scan_name"])
    try:
        staged = {}
        for (k, dt, shape, off) in msg["layout"]:
            view = np.ndarray(shape, np.dtype(dt), buffer=shm.buf, offset=off)
            if k in _DEVICE_DTYPES and device is not None:
                if stager is not None:
                    staged[k] = view
                else:
                    sc[k] = torch.from_numpy(view).to(device=device, dtype=_DEVICE_DTYPES[k])  # synchronous copy
            else:
                sc[k] = view.copy() if view.size else []
            del view
        if staged:
            sc.update(stager.upload(staged, _DEVICE_DTYPES))
            staged.clear()
    finally:
        shm.close()
        shm.unlink()
    return sc


_DEVICE_DTYPES = {"coords_float": torch.float64, "mask_feats": torch.float32, "spp": torch.int64,
                  "semantic_label": torch.float64, "instance_label": torch.float64}


class PinnedStager:
    """Host arrays -> device through a pinned buffer this object owns (one per uploading thread).

    `torch.from_numpy(a).to(device)` on pageable memory (a shared-memory block, a memory-mapped cache file) is a
    synchronous copy through the runtime's own small staging buffer, and a dtype change is done by the CPU first:
    ~25 ms per 150k-point scene and thread.  Here every array of a scene is copied once into the pinned buffer (numpy
    releases the GIL for it), leaves as one asynchronous DMA in its SOURCE dtype, and is converted on the device.
    The stream is synchronised before the call returns: the buffer is free again and the tensors are ready for any
    stream."""

    def __init__(self, device):
        self.device = device
        self.buf = None

    def upload(self, arrays, dtypes):
        """arrays: {key: ndarray}; returns {key: device tensor of dtypes[key]} (current stream of the thread)."""
        views, off = [], 0
        for k, a in arrays.items():
            a = np.asarray(a)
            views.append((k, a, off))
            off += (a.nbytes + 63) // 64 * 64
        if self.buf is None or self.buf.numel() < off:
            self.buf = torch.empty(max(off, 1) * 5 // 4, dtype=torch.uint8).pin_memory()
        host = self.buf.numpy()
        out = {}
        for k, a, o in views:
            if a.size == 0:
                out[k] = torch.empty(a.shape, dtype=dtypes[k], device=self.device)
                continue
            dst = host[o:o + a.nbytes].view(a.dtype).reshape(a.shape)
            np.copyto(dst, a)
            tdt = torch.from_numpy(np.empty(0, a.dtype)).dtype
            t = self.buf[o:o + a.nbytes].view(tdt).reshape(a.shape).to(self.device, non_blocking=True)
            out[k] = t if t.dtype == dtypes[k] else t.to(dtypes[k])
        torch.cuda.current_stream(self.device).synchronize()
        return out


def scene_to_device(sc, device):
    """Upload the per-point arrays of a read_scene dict once; add_instance_info and make_job then share them."""
    sc = dict(sc)
    for k, dt in _DEVICE_DTYPES.items():
        if not isinstance(sc[k], torch.Tensor):
            sc[k] = torch.from_numpy(np.ascontiguousarray(np.asarray(sc[k]))).to(device=device, dtype=dt)
    return sc


def _save_arrays(save_path, arrays, spp_inv=None):
    """Loader process: the file write of save_scene from host arrays."""
    sem, ins, prob, mu, var = arrays
    if spp_inv is not None:
        mu, var = mu[spp_inv], var[spp_inv]
    write_label_file(save_path, (sem, ins, prob, mu, var))


_T_IMPORT = time.time()
_T0_PERF = [0.0]


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


def run_worker_dry(filenames, args, rank):
    """`--dry_run` (hidden; tools/host_ceiling.py, bench.py --dry-run, tests/test_dist_cpu.py): the HOST side of one
    worker with no GPU in it -- the same scene list / claim queue / loader threads / native reader, the host
    preprocessing of read_scene, one pass over the arrays in place of the copy into the pinned staging buffer, all-zero
    stand-in outputs of the right shapes through the same writer, the same result file for the parent.  Measures what
    the host of a W-GPU node can feed and drain, and lets the farm's control path run on a box without a GPU.  The
    label files go to <save_folder>.DRY_RUN, never to --save_folder: nothing here is a pseudo-label."""
    import concurrent.futures as cf

    out_folder = osp.normpath(args.save_folder) + ".DRY_RUN"
    os.makedirs(out_folder, exist_ok=True)
    print("[gen_ps] WARNING: --dry_run: NO pseudo-labels are generated; all-zero stand-in outputs go to %s"
          % out_folder, file=sys.stderr)
    n_workers = max(1, int(getattr(args, "n_workers", 1)))
    phys = max(1, (os.cpu_count() or 2) // 2)
    n_threads = int(getattr(args, "loader_threads", -1))
    if n_threads <= 0:
        n_threads = min(16, max(4, phys // (2 * n_workers)))
    queue = ClaimQueue(filenames, args.claim_dir) if getattr(args, "claim_dir", None) else None
    pool = cf.ThreadPoolExecutor(max_workers=n_threads)
    read_args = (args.data_root, args.use_deepfeat, args.deepfeat_folder)
    t0 = time.time()
    done, failed_names = 0, []

    import threading

    tls = threading.local()

    def one(fn):
        if not hasattr(tls, "scratch"):
            tls.scratch = SceneScratch()
        sc = read_scene(fn, *read_args, scratch=tls.scratch)
        n = int(len(sc["spp"]))
        for k in _DEVICE_DTYPES:  # the upload's pass over the arrays (host -> the thread's staging buffer)
            a = np.asarray(sc[k])
            np.copyto(tls.scratch.get("stage_" + k, a.shape, a.dtype), a)
        s = max(1, n // 50)
        arrays = (tls.scratch.get("o0", (n,), np.int32), tls.scratch.get("o1", (n,), np.int32),
                  tls.scratch.get("o2", (n,), np.float32), tls.scratch.get("o3", (s,), np.float32),
                  tls.scratch.get("o4", (s,), np.float32))
        arrays[0][:] = 0
        arrays[1][:] = 0
        arrays[2][:] = 1.0
        arrays[3][:] = -100.0
        arrays[4][:] = -100.0
        _save_arrays(osp.join(out_folder, sc["scan_name"] + ".pth"), arrays, None)
        return sc["scan_name"]

    ahead = []
    chunks = _chunks(filenames, args, queue)
    for _ in range(2):
        c = next(chunks, None)
        if c:
            ahead.append([(fn, pool.submit(one, fn)) for fn in c])
    while ahead:
        futs = ahead.pop(0)
        c = next(chunks, None)
        if c:
            ahead.append([(fn, pool.submit(one, fn)) for fn in c])
        for fn, f in futs:
            try:
                f.result()
                done += 1
            except Exception as e:  # noqa: BLE001
                print("[gen_ps] %s: failed: %r" % (fn, e), file=sys.stderr)
                failed_names.append(fn.split("/")[-1][:12])
    pool.shutdown()
    dt = time.time() - t0
    print("[gen_ps] device %d: %d scenes written, %d skipped/failed, %.2f s (%.2f scenes/s)"
          % (rank, done, len(failed_names), dt, done / dt if dt > 0 else 0.0))
    print("[gen_ps] device %d: start-up %.1f s (process start -> generator ready), first batch out after %.1f s more, "
          "%d loader threads, %d loader processes, %s file I/O, dry run (no GPU)"
          % (rank, t0 - _T_IMPORT, 0.0, n_threads, 0,
             "native (gapro_pth_*)" if pth_io.native_enabled() else "torch.load / torch.save"))
    result = dict(done=done, failed=sorted(failed_names), miou={}, seconds=dt, startup_seconds=t0 - _T_IMPORT,
                  first_batch_seconds=0.0, timeout_retries=0, dry_run=True)
    if getattr(args, "job_dir", None):
        import json

        tmp = osp.join(args.job_dir, "result.%d.json.tmp" % device_index_rank(args))
        with open(tmp, "w") as fh:
            json.dump(result, fh)
        os.replace(tmp, tmp[:-4])
    return result


def run_worker(filenames, args, device_index):
    """One GPU: scenes are read from disk by a pool of loader threads or processes (two batches ahead), go
    through the software-pipelined generator batch by batch (Pipeline.run_stream), and are written by the
    same pool."""
    import concurrent.futures as cf
    import threading

    n_procs = int(getattr(args, "loader_procs", 0))
    n_workers = max(1, int(getattr(args, "n_workers", 1)))  # GPU workers sharing this host (--devices)
    native = pth_io.native_enabled()
    phys = max(1, (os.cpu_count() or 2) // 2)
    n_threads = int(getattr(args, "loader_threads", -1))
    if n_threads <= 0:  # auto: with the native reader the threads ARE the loaders
        n_threads = min(16, max(4, phys // (2 * n_workers))) if native else 4
    if n_procs < 0 and native:
        n_procs = 0  # nothing left that holds the GIL for long: no loader process, no shared-memory hand-over
    if n_procs < 0:
        # auto: at most 16 per worker (more only adds start-up time: 16 -> 32 -> 64 loaders: 223 -> 200 -> 159 scenes/s),
        # and all workers' loaders together at most half the physical cores -- eight workers x 16 loaders on one host
        # oversubscribe it long before eight GPUs are fed
        n_procs = min(16, max(2, phys // (2 * n_workers)), (os.cpu_count() or 1) // 4)
        if n_procs < 2:  # a small host: one loader process is no faster than the threads
            n_procs = 0
    procs = None
    if n_procs:  # start the loader processes BEFORE this process initialises the GPU
        import multiprocessing as mp

        procs = mp.get_context("spawn").Pool(n_procs, initializer=_loader_init)
    pipe = Pipeline(device=device_index, training_iter=50, init_mean_std=args.init_mean_std, seed=args.seed)
    pipe.strict = False  # a scene that cannot be processed is reported and skipped, the rest of its batch is written
    if os.environ.get("GAPRO_DRIVER_TIMES"):
        pipe.trace = []  # host-side stage timeline of the pipeline (printed at the end)
    # The fit workspace is allocated NOW, by a helper thread, while the first scenes are read: its size is only known
    # after the first batch has been scheduled (~2 s in), and the hipMalloc + clear of ~25 GB then takes another
    # 1.2 .. 2.4 s before the first launch.  Estimate: ~330 bytes of workspace per point of a batch of ScanNet-like
    # scenes (14 GB per 256 scenes of the train-split mix) with the usual 30 % headroom, ~91 bytes of scene file per
    # point; a batch that needs more grows it as before, a smaller one just leaves memory unused (288 GB per GPU).
    try:
        sizes = sorted((os.path.getsize(fn) for fn in filenames), reverse=True)
        nb = max(1, min(int(args.batch_scenes), len(sizes)))
        # the claim queue hands out the largest files first; a static list comes in name order (a mix: mean + margin)
        batch_bytes = sum(sizes[:nb]) if getattr(args, "claim_dir", None) else 1.2 * nb * sum(sizes) / max(1, len(sizes))
        est = int(batch_bytes / 91.0 * 330.0 * 1.3)
        if not os.environ.get("GAPRO_NO_PREALLOC") and est > (1 << 28):
            threading.Thread(target=pipe.prealloc_workspace, args=(min(est, 64 << 30),), daemon=True).start()
    except OSError:
        pass
    dev = pipe.device
    _T0_PERF[0] = time.perf_counter()
    done = failed = 0
    failed_names = []  # scan names of every scene this worker could not write (exit status 3)
    miou = {}          # --eval_pslabel: scan name -> per-instance IoUs (float32), gen_ps.py:116-124
    t0 = time.time()
    queue = ClaimQueue(filenames, args.claim_dir) if getattr(args, "claim_dir", None) else None

    def chunk_iter():
        return _chunks(filenames, args, queue)

    pool = cf.ThreadPoolExecutor(max_workers=max(1, n_threads))
    meta = []  # per yielded batch: (scenes, jobs)
    read_args = (args.data_root, args.use_deepfeat, args.deepfeat_folder)
    raw_cache = getattr(args, "raw_cache", None)
    if raw_cache:
        os.makedirs(raw_cache, exist_ok=True)
    cache_hits = [0]
    spent = dict(wait=0.0, upload=0.0, boxes=0.0, jobs=0.0, export=0.0)  # main-thread seconds, GAPRO_DRIVER_TIMES=1

    tls = threading.local()

    def side_stream():
        if not hasattr(tls, "stream"):  # one stream per pool thread: its copies stay off the default stream
            tls.stream = torch.cuda.Stream(dev)
        return tls.stream

    def stager():
        if not hasattr(tls, "stager"):
            tls.stager = PinnedStager(dev)
        return tls.stager

    def with_job(sc):
        """Pool thread: the SceneJob of an uploaded scene (tensor wrapping and argument checks: ~1 ms of Python per scene,
        a quarter of a second per 256-scene batch when the main thread did it between two launches)."""
        if sc is not None:
            sc["job"] = make_job(sc["coords_float"], sc["mask_feats"], sc["spp"], sc["instance_cls"], sc["instance_box"],
                                 sc["instance_box_volume"], sc["wall_box"], sc["wall_box_volume"],
                                 instance_classes=18, ground_h=0.1, thresh_spp_occu=0.999, device=dev,  # :106-110
                                 scene_key=zlib.crc32(sc["scan_name"].encode()))
        return sc

    def upload(r):
        """Pool thread: map the loader's block, upload from it, GT boxes of the scene (gapro_instance_info only
        touches the buffers it is given, so it may run beside the generator)."""
        msg = r.get(600)  # a loader that died (e.g. killed for memory) must not hang the run: the scene is skipped
        with torch.cuda.stream(side_stream()):
            dev_sc = _scene_from_shm(msg, dev, stager())  # (always: the shared-memory block must be released)
            if msg.get("info") is None:
                return None
            dev_sc.update(msg["info"])
            return with_job(dev_sc)

    def export(path, job, o, ready):
        """Pool thread: device -> host on the thread's stream; pickling and the file write go to a loader process."""
        st = side_stream()
        with torch.cuda.stream(st):
            st.wait_event(ready)
            arrays = (o[0].int().cpu().numpy(), o[1].int().cpu().numpy()) + tuple(t.cpu().numpy() for t in o[2:])
            inv = job.spp_inv.cpu().numpy() if args.broadcast_mu_var else None
        return procs.apply_async(_save_arrays, (path, arrays, inv))

    def upload_cached(fn, sc):
        """Pool thread: a raw-cache hit -- the arrays are memory-mapped files, uploaded straight from the page cache."""
        info = add_instance_info(dict(sc), "host")  # on the mapped host arrays, before the upload (see read_upload)
        if info is None:
            return None
        with torch.cuda.stream(side_stream()):
            dev_sc = dict(scan_name=sc["scan_name"], **{k: info[k] for k in _INFO_KEYS})
            for k in _SHM_KEYS:
                if k not in _DEVICE_DTYPES:
                    dev_sc[k] = np.array(sc[k]) if len(sc[k]) else []
            dev_sc.update(stager().upload({k: sc[k] for k in _SHM_KEYS if k in _DEVICE_DTYPES}, _DEVICE_DTYPES))
            return with_job(dev_sc)

    def scratch():
        if not hasattr(tls, "scratch"):
            tls.scratch = SceneScratch()
        return tls.scratch

    def read_and_cache(fn):
        sc = read_scene(fn, *read_args, scratch=scratch())  # consumed (uploaded) by this thread before its next read
        if raw_cache:
            try:
                write_raw_cache(raw_cache_path(raw_cache, fn), sc, _source_stamp(fn, *read_args))
            except OSError as e:
                print("[gen_ps] raw cache not written for %s: %r" % (fn, e), file=sys.stderr)
        return sc

    def read_upload(fn):
        """Pool thread, no loader process: native read (GIL released while the payloads are transcoded), host
        preprocessing in NumPy, upload through the thread's pinned stager, GT boxes on the device."""
        sc = read_and_cache(fn)
        # GT boxes on the HOST arrays, before the upload (gapro_scene_instance_boxes, one native pass): the device form
        # is a kernel, and a short kernel is not dispatched beside a running fit launch -- every loader thread waited for
        # the launch to drain once per scene (115 scenes/s of loading beside a launch against 850 without)
        if add_instance_info(sc, "host") is None:
            return None
        with torch.cuda.stream(side_stream()):
            dev_sc = dict(scan_name=sc["scan_name"], **{k: sc[k] for k in _INFO_KEYS})
            for k in _SHM_KEYS:
                if k not in _DEVICE_DTYPES:
                    dev_sc[k] = sc[k]
            dev_sc.update(stager().upload({k: sc[k] for k in _DEVICE_DTYPES}, _DEVICE_DTYPES))
            return with_job(dev_sc)

    def export_native(path, job, o, ready):
        """Pool thread, no loader process: device -> host on the thread's stream, then the native writer."""
        st = side_stream()
        with torch.cuda.stream(st):
            st.wait_event(ready)
            arrays = (o[0].int().cpu().numpy(), o[1].int().cpu().numpy()) + tuple(t.cpu().numpy() for t in o[2:])
            inv = job.spp_inv.cpu().numpy() if args.broadcast_mu_var else None
        _save_arrays(path, arrays, inv)

    def submit(chunk):
        out, misses = [], []
        for fn in chunk:
            sc = read_raw_cache(raw_cache_path(raw_cache, fn), _source_stamp(fn, *read_args)) if raw_cache else None
            if sc is not None:
                cache_hits[0] += 1
                out.append((fn, pool.submit(upload_cached, fn, sc), True))
            else:
                misses.append(fn)
        if procs is not None:  # read in a loader process; a pool thread maps the block and uploads from it
            reads = [(fn, procs.apply_async(_read_scene_shm, (fn,) + read_args + (raw_cache,))) for fn in misses]
            out += [(fn, pool.submit(upload, r), True) for fn, r in reads]
        else:
            out += [(fn, pool.submit(read_upload, fn), True) for fn in misses]
        order = {fn: i for i, fn in enumerate(chunk)}
        return sorted(out, key=lambda e: order[e[0]])

    def fetch(fut, on_device):
        t = time.time()
        got = fut.result()
        t1 = time.time()
        if on_device:
            spent["wait"] += t1 - t
            return got
        sc = scene_to_device(got, dev)
        t2 = time.time()
        sc = add_instance_info(sc, dev)
        spent["wait"] += t1 - t
        spent["upload"] += t2 - t1
        spent["boxes"] += time.time() - t2
        return sc

    def batches():
        nonlocal failed
        chunks = chunk_iter()
        ahead = []
        for _ in range(2):  # the disk reads of two batches are in flight
            c = next(chunks, None)
            if c:
                ahead.append(submit(c))
        while ahead:
            futs = ahead.pop(0)
            c = next(chunks, None)
            if c:
                ahead.append(submit(c))
            scenes = []
            for fn, fut, on_device in futs:
                try:
                    sc = fetch(fut, on_device)
                    if sc is None:
                        print("[gen_ps] %s: no instances, skipped" % fn, file=sys.stderr)
                        failed += 1
                        failed_names.append(fn.split("/")[-1][:12])
                        continue
                    scenes.append(sc)
                except Exception as e:  # noqa: BLE001 - one bad scene must not kill the run
                    print("[gen_ps] %s: load failed: %r" % (fn, e), file=sys.stderr)
                    failed += 1
                    failed_names.append(fn.split("/")[-1][:12])
            if not scenes:
                continue
            if pipe.trace is not None:
                pipe.trace.append((time.perf_counter(), -1, "%d scenes of a batch fetched" % len(scenes)))
            t = time.time()
            jobs = [s["job"] if "job" in s else with_job(s)["job"] for s in scenes]
            spent["jobs"] += time.time() - t
            meta.append((scenes, jobs))
            yield jobs

    def host_only_stream(batch_iter):
        """GAPRO_DRIVER_HOST_ONLY=1 (tools/bench_driver.py --host-only): everything the host does for a scene -- read,
        unpickle, upload, GT boxes, device -> host, pickle, write -- with the generation replaced by zero outputs of
        the right shapes.  Measures what the host can feed W workers, whatever the GPUs do; never a product mode."""
        for jobs in batch_iter:
            outs = []
            for j in jobs:
                n = int(j.coords.shape[0])
                outs.append((torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev),
                             torch.ones(n, dtype=torch.float32, device=dev),
                             torch.full((max(1, n // 50),), -100.0, dtype=torch.float32, device=dev),
                             torch.full((max(1, n // 50),), -100.0, dtype=torch.float32, device=dev)))
            yield outs

    host_only = bool(os.environ.get("GAPRO_DRIVER_HOST_ONLY"))
    out_folder = args.save_folder
    if host_only:  # a measurement mode must never leave files that look like pseudo-labels
        out_folder = osp.normpath(args.save_folder) + ".HOST_ONLY_MEASUREMENT"
        os.makedirs(out_folder, exist_ok=True)
        print("[gen_ps] WARNING: GAPRO_DRIVER_HOST_ONLY is set: NO pseudo-labels are generated; the all-zero stand-in "
              "outputs go to %s, never to --save_folder" % out_folder, file=sys.stderr)
    writes = []
    t_first = None
    n_first = 0
    for outs in (host_only_stream(batches()) if host_only else pipe.run_stream(batches())):
        if t_first is None:
            t_first = time.time()
            n_first = len(outs)
        scenes, jobs = meta.pop(0)
        t_exp = time.time()
        ready = torch.cuda.current_stream(dev).record_event()  # run_stream ordered the outputs on this stream
        for s, job, o in zip(scenes, jobs, outs):
            if o is None:  # Pipeline.strict = False: this scene could not be processed, the others could
                print("[gen_ps] warning: %s skipped: %s" % (s["scan_name"], job.error), file=sys.stderr)
                failed += 1
                failed_names.append(s["scan_name"])
                continue
            if args.eval_pslabel:
                from .eval_ps_labels import get_miou_scene

                sem_gt = s["semantic_label"].int()
                ins_gt = s["instance_label"].int()
                sem_gt[sem_gt != -100] -= 2  # :119-120
                sem_gt[(sem_gt == -1) | (sem_gt == -2)] = 18
                ious = get_miou_scene(sem_gt.long(), ins_gt.long(), o[0].long(), o[1].long())
                print("miou", ious)
                miou[s["scan_name"]] = ious.float().cpu().numpy()  # :125 ious_arr.append(ious)
            path = osp.join(out_folder, s["scan_name"] + ".pth")
            if procs is not None:
                writes.append(pool.submit(export, path, job, o, ready))
            else:  # device -> host and the file write on a pool thread (native writer: no GIL)
                writes.append(pool.submit(export_native, path, job, o, ready))
            done += 1
        spent["export"] += time.time() - t_exp
    try:
        for w in writes:
            try:
                w.result().get() if procs is not None else w.result()
            except Exception as e:  # noqa: BLE001 - a failed write loses that scene, not the run
                print("[gen_ps] a label file could not be written: %r" % (e,), file=sys.stderr)
                failed += 1
                done -= 1
                failed_names.append("<write failed: %r>" % (e,))
    finally:  # queued writes are flushed and the loader processes released whatever happened above
        pool.shutdown()
        if procs is not None:
            procs.close()
            procs.join()
    dt = time.time() - t0
    print("[gen_ps] device %d: %d scenes written, %d skipped/failed, %.2f s (%.2f scenes/s)%s"
          % (device_index, done, failed, dt, done / dt if dt > 0 else 0.0,
             ", %d from the raw cache" % cache_hits[0] if raw_cache else ""))
    # start-up is reported apart from the rate: at ~300 scenes/s a 1201-scene split is a few seconds of work, and the
    # spawned interpreters / library loads of a worker and its loaders take longer than that
    print("[gen_ps] device %d: start-up %.1f s (process start -> generator ready), first batch out after %.1f s more, "
          "%d loader threads, %d loader processes, %s file I/O%s"
          % (device_index, t0 - _T_IMPORT, (t_first - t0) if t_first else 0.0, n_threads, n_procs,
             "native (gapro_pth_*)" if native else "torch.load / torch.save",
             ", host-only measurement mode" if host_only else ""))
    # the first batch carries the one-off costs of a worker (code objects, the fit workspaces of every pipeline slot:
    # tens of GB of hipMalloc + fill, the first reads with an empty pipeline); what follows it is the steady state
    t_end = t0 + dt
    if t_first is not None and done > n_first and t_end > t_first:
        print("[gen_ps] device %d: steady state %.2f scenes/s (%d scenes in the %.2f s after the first batch of %d)"
              % (device_index, (done - n_first) / (t_end - t_first), done - n_first, t_end - t_first, n_first))
    if os.environ.get("GAPRO_DRIVER_TIMES"):
        print("[gen_ps] main-thread seconds: " + ", ".join("%s %.2f" % kv for kv in spent.items()))
        ids = {}
        for t, bid, name in (pipe.trace or [])[:60]:
            print("[gen_ps]   %8.3f s  batch %d  %s" % (t - _T0_PERF[0], ids.setdefault(bid, len(ids)), name))
    result = dict(done=done, failed=sorted(failed_names), miou={k: [float(x) for x in v] for k, v in miou.items()},
                  seconds=dt, startup_seconds=t0 - _T_IMPORT, first_batch_seconds=(t_first - t0) if t_first else 0.0,
                  timeout_retries=int(getattr(pipe, "timeout_retries", 0)))
    job_dir = getattr(args, "job_dir", None)
    if job_dir:  # several workers: the parent aggregates (mean IoU over all scenes, failed scans, exit status)
        import json

        tmp = osp.join(job_dir, "result.%d.json.tmp" % device_index_rank(args))
        with open(tmp, "w") as fh:
            json.dump(result, fh)
        os.replace(tmp, tmp[:-4])
    return result


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
    parser.add_argument("--raw_cache", type=str, default=None)
    parser.add_argument("--loader_threads", type=int, default=-1)
    parser.add_argument("--loader_procs", type=int, default=-1)
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
    if len(devices) > 1 and args.raw_cache is None:
        # several workers on one host: unpickling the .pth files caps the HOST at ~350 scenes/s whatever the number of
        # loaders, one GPU alone takes ~300 -- the memory-mapped raw cache (first pass writes it, every later pass and
        # every restart reads it) is what keeps more than one GPU fed.  `--raw_cache none` switches it off.
        args.raw_cache = osp.normpath(args.save_folder) + ".raw_cache"  # next to the label folder, never inside it
    if args.raw_cache in ("none", "None", ""):
        args.raw_cache = None
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
        prev_threads = torch.get_num_threads()
        _loader_init()
        try:
            result = run_worker_dry(mine, args, r) if args.dry_run else run_worker(mine, args, devices[r])
        finally:
            torch.set_num_threads(prev_threads)
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
    sys.exit(main())
