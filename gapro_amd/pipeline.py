"""Host orchestration of the pseudo-label path on one GPU: enumerate -> batch-fit -> ordered merge.

Mirrors what reference gapro/gen_ps_utils.py:293-482 does for one scene, but staged so that any
number of scenes share each device launch:

  stage A  gapro_partition_prepare   scene stats + dense superpoint ranks        (device)
  stage B  gapro_partition_pool      membership + pooling in one pass            (device)
  stage C  gapro_schedule_build      static pair schedule                         (host, C++)
  stage D  gapro_svgp_fit_batch      every GP fit of every scene in ONE launch    (device)
  stage E  gapro_schedule_merge      ordered merge, fallback, label tables        (host, C++)
  stage F  gapro_broadcast_labels    superpoint -> point                          (device)

All arithmetic happens in libgapro_hip.so.  Device memory, streams and events come from a backend (devmem.py): torch's
(default: the Python API shims take and return torch tensors) or the library's own arena ("native": the gen_ps workers,
which then never import torch -- round 6).
"""
from __future__ import annotations

import os
import sys
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import Context, FitDesc, SceneHeader, SceneTask, ScheduleCounts


# A fit is "soft" when its outputs move by more than this (sigma^2 relative, p absolute) when the jitter on K_ZZ's diagonal
# is scaled by (1 + 1e-11) (Pipeline.reproducibility_probe): fifty Adam steps amplify last-bit differences ~1e9-fold on
# such a fit, in ANY float64 implementation.  Measured on the S3DIS-shaped test scene's 66 fits: the
# two soft ones move by 6.4e-5 / 6.8e-4, every other fit by 8e-8 .. 1.2e-6
REPRO_SOFT = 1.0e-5


def _ptr(t) -> C.c_void_p:
    if t is None:
        return C.c_void_p(0)
    if hasattr(t, "data_ptr"):  # torch.Tensor / devmem buffer
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)


def _to_np(x, dtype):
    if hasattr(x, "detach"):  # torch.Tensor
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(x), dtype=dtype)


class LazyViews(dict):
    """Per-scene views into the batch-wide buffers, created on first access: a batch of 256 scenes would otherwise
    pay thousands of tensor-view constructions on the host between two fit launches; the kernels only need the
    addresses, which are plain integer arithmetic."""

    def __init__(self):
        super().__init__()
        self._specs = {}

    def spec(self, name, buf, off, nbytes, dtype, shape):
        self._specs[name] = (buf, off, nbytes, dtype, shape)
        self.pop(name, None)
        return buf.data_ptr() + off

    def __missing__(self, name):
        buf, off, nbytes, dtype, shape = self._specs[name]
        v = buf[off:off + nbytes].view(dtype).view(*shape)
        self[name] = v
        return v

    def __contains__(self, name):
        return dict.__contains__(self, name) or name in self._specs


@dataclass
class SceneJob:
    """Inputs of one gen_pseudo_label_gaussian_process call (reference gen_ps_utils.py:293-307)."""
    coords: object  # f64[N,3] device (torch.Tensor or devmem.DevBuf)
    feats: object  # f32[N,D] device
    spp: object  # i64[N] device
    instance_cls: np.ndarray  # i64[Bi]
    instance_box: np.ndarray  # f32[Bi,6]
    instance_box_volume: np.ndarray  # f32[Bi]
    wall_box: np.ndarray  # f32[Bw,6] (may be empty)
    wall_box_volume: np.ndarray  # f32[Bw]
    instance_classes: int = 18
    ground_h: float = 0.1
    thresh_spp_occu: float = 0.8
    # --- filled by the stages ---
    header: Optional[SceneHeader] = None
    n_spps: int = 0
    boxes: Optional[np.ndarray] = None
    boxes_cls: Optional[np.ndarray] = None
    boxes_volume: Optional[np.ndarray] = None
    feats_row_base: int = 0
    dev: dict = field(default_factory=LazyViews)
    host: dict = field(default_factory=LazyViews)
    schedule: Optional[C.c_void_p] = None
    counts: Optional[ScheduleCounts] = None
    fit_base: int = 0
    out_base: int = 0
    outputs: Optional[tuple] = None
    error: Optional[Exception] = None  # why the scene could not be processed (Pipeline.strict = False)
    scene_key: int = 0  # stable id of the scene (e.g. crc32 of the scan name): seeds the optional initial-mean noise

    @property
    def spp_inv(self):
        """i32[N] dense superpoint rank of every point (view into the batch-wide buffer, created on demand)."""
        return self.dev["spp_inv"] if "spp_inv" in self.dev else None

    @property
    def n_points(self):
        return int(self.coords.shape[0])

    @property
    def n_boxes(self):
        return int(self.boxes.shape[0])


def make_job(coords_float, mask_feats, spp, instance_cls, instance_box, instance_box_volume, wall_box,
             wall_box_volume, instance_classes=18, ground_h=0.1, thresh_spp_occu=0.8, device=None,
             scene_key: int = 0, backend=None) -> SceneJob:
    """backend: a devmem.NativeBackend (arrays are its device buffers or host arrays; no torch), else torch."""
    if backend is not None and backend.name == "native":
        def dev(x, dtype):
            if backend.is_device_array(x):
                if x.dtype != np.dtype(dtype):
                    raise ValueError("device buffer of dtype %s where %s is expected" % (x.dtype, np.dtype(dtype)))
                return x
            return backend.from_numpy(np.ascontiguousarray(np.asarray(x), dtype=dtype))

        f64, f32, i64 = np.float64, np.float32, np.int64
    else:
        import torch

        device = torch.device(device if device is not None else "cuda:0")

        def dev(x, dtype):
            t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
            return t.to(device=device, dtype=dtype, non_blocking=True).contiguous()

        f64, f32, i64 = torch.float64, torch.float32, torch.int64
    coords = dev(coords_float, f64)
    feats = dev(mask_feats, f32)  # mask_feats.float()  gen_ps_utils.py:315
    if feats.dim() != 2 or coords.dim() != 2 or coords.shape[1] != 3 or feats.shape[0] != coords.shape[0]:
        raise ValueError("coords_float must be [N,3] and mask_feats [N,D]")
    sp = dev(spp, i64).reshape(-1)
    if sp.shape[0] != coords.shape[0]:
        raise ValueError("spp must have one id per point")
    ibox = _to_np(instance_box, np.float32).reshape(-1, 6)
    has_wall = wall_box is not None and len(wall_box) > 0
    return SceneJob(coords, feats, sp, _to_np(instance_cls, np.int64).reshape(-1), ibox,
                    _to_np(instance_box_volume, np.float32).reshape(-1),
                    _to_np(wall_box, np.float32).reshape(-1, 6) if has_wall else np.zeros((0, 6), np.float32),
                    _to_np(wall_box_volume, np.float32).reshape(-1) if has_wall else np.zeros((0,), np.float32),
                    int(instance_classes), float(ground_h), float(thresh_spp_occu), scene_key=int(scene_key))


def _desc_table(descs, n_fits: int) -> np.ndarray:
    return np.frombuffer(descs, dtype=np.int32, count=n_fits * (C.sizeof(FitDesc) // 4)).reshape(n_fits, -1)


def fit_flops_each(descs, n_fits: int, feat_dim: int, training_iter: int) -> np.ndarray:
    """Algorithmic FLOPs of every fit of a batch, SURVEY.md section 8(d):
    F_fit = I (8.33 M^3 + 12 D M^2) + (M^3/3 + 2 M^2 T + 2 D (M^2 + M T)),  M = m1 + m2."""
    if n_fits <= 0:
        return np.zeros(0)
    raw = _desc_table(descs, n_fits)
    m = (raw[:, 0] + raw[:, 1]).astype(np.float64)
    t = raw[:, 2].astype(np.float64)
    d = float(feat_dim)
    return training_iter * (8.33 * m**3 + 12.0 * d * m * m) + (m**3 / 3.0 + 2.0 * m * m * t + 2.0 * d * (m * m + m * t))


def fit_flops(descs, n_fits: int, feat_dim: int, training_iter: int) -> float:
    return float(fit_flops_each(descs, n_fits, feat_dim, training_iter).sum())


class FitTiming:
    """Device-side timing of one fit launch (gapro_fit_timing): HIP events recorded by the library on the
    streams its kernels run on.  read() blocks until the launch has finished."""

    def __init__(self, ctx, handle, flops_strip, flops_staged, flops_small, m, flops_cluster=0.0, flops_wave=0.0):
        self.ctx, self.handle = ctx, handle
        self.flops_strip, self.flops_staged, self.flops_small, self.m = flops_strip, flops_staged, flops_small, m
        self.flops_cluster, self.flops_wave = flops_cluster, flops_wave
        self.ms = None

    @property
    def flops(self):
        return self.flops_strip + self.flops_staged + self.flops_small + self.flops_cluster + self.flops_wave

    def read(self):
        """(staged kernel ms, strip kernel ms, first start -> last end ms, small-fit strip kernel ms, cluster kernel
        ms, wave-per-fit kernels ms); the span covers all of them"""
        if self.ms is None:
            out = (C.c_float * 5)()
            self.ctx.check(self.ctx.lib.gapro_fit_timing_read(self.ctx.handle, self.handle, out))
            w = C.c_float(0.0)
            self.ctx.check(self.ctx.lib.gapro_fit_timing_read_wave(self.ctx.handle, self.handle, C.byref(w)))
            self.ms = (float(out[0]), float(out[1]), float(out[2]), float(out[3]), float(out[4]), float(w.value))
        return self.ms

    def cluster_info(self):
        """(clusters of more than one workgroup, those whose members did not share an XCD, member workgroups) of this
        launch's cluster kernel; read before 64 further launches."""
        out = (C.c_int32 * 3)()
        self.ctx.check(self.ctx.lib.gapro_fit_timing_cluster_info(self.ctx.handle, self.handle, out))
        return int(out[0]), int(out[1]), int(out[2])

    def offsets(self, ref):
        """(first kernel start, last kernel end) of this launch in ms after the start of launch `ref`."""
        out = (C.c_float * 2)()
        self.ctx.check(self.ctx.lib.gapro_fit_timing_offsets(self.ctx.handle, ref.handle, self.handle, out))
        return float(out[0]), float(out[1])

    def close(self):
        if self.handle is not None:
            self.ctx.lib.gapro_fit_timing_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


class Pipeline:
    def __init__(self, device=0, training_iter=50, init_mean_std=0.0, seed=0, eval_stale_chol=False,
                 spp_range_cap=None, force_staged=False, precision="f64", cluster_all=False, backend="torch"):
        from .devmem import make_backend

        if not isinstance(device, int):
            import torch

            device = torch.device(device).index or 0
        # backend: "torch" (tensors in, tensors out) or "native" (the library's arena, no torch in the process);
        # either raises when there is no HIP device -- there is no CPU fallback
        self.be = make_backend(backend, device)
        self.device = self.be.device
        self.ctx = self.be.ctx
        self.lib = self.ctx.lib
        self.opt = _lib.default_fit_options(training_iter)
        self.opt.eval_stale_chol = 1 if eval_stale_chol else 0
        # reserved: 0 default dispatch (strip-streaming kernel for M_p <= 128, LDS-staged kernel up to 512, generic
        # kernel beyond); 1 = never the strip kernel (A/B runs, tests)
        self.opt.reserved = 1 if force_staged else 0
        # precision: "f64" (default, float64 throughout) or "mixed" (the reference's float32 / float64 split, run by
        # the cluster kernel; fits routed elsewhere stay float64).  cluster_all: the cluster kernel for every fit it
        # can take (precision sweeps, A/B runs)
        if precision not in ("f64", "mixed"):
            raise ValueError("precision must be 'f64' or 'mixed'")
        self.opt.precision = 1 if precision == "mixed" else 0
        if cluster_all:
            self.opt.reserved |= 16
        # A/B runs and tests only: further debug bits of gapro_fit_options.reserved (include/gapro_hip.h) from the
        # environment.  Never silent: a leaked variable changes the routing of every fit.
        env_flags = os.environ.get("GAPRO_FIT_FLAGS", "").strip()
        if env_flags:
            try:
                bits = int(env_flags, 0)
            except ValueError:
                raise ValueError("GAPRO_FIT_FLAGS=%r is not an integer (debug bits of gapro_fit_options.reserved)"
                                 % env_flags) from None
            if bits:
                print("[gapro_amd] GAPRO_FIT_FLAGS=0x%x: debug routing bits are ORed into every fit launch of this "
                      "process (measurement / test setting, not a product mode)" % bits, file=sys.stderr)
            self.opt.reserved |= bits
        # a fit that comes back GAPRO_ERR_TIMEOUT (a cluster member was not resident in time: transient, unlike a failed
        # factorisation) is run once more on the single-workgroup route before its scene is given up
        self.retry_timeouts = True
        self.timeout_retries = 0  # fits that went through the retry, over the life of this object

        self.init_mean_std = float(init_mean_std)
        self.seed = int(seed)
        self.spp_range_cap = spp_range_cap
        # strict: a scene that cannot be processed (non-finite input, id range beyond the rank table, a GP fit that
        # fails after the jitter retries) raises, as the reference would.  The gen_ps driver clears it: the scene's
        # outputs come back as None with job.error set, and the other scenes of the batch are unaffected.
        self.strict = True
        self.last_stats = {}
        self.trace = None  # set to a list to collect (time, batch id, stage) host timestamps
        self.workspace_headroom = 1.3  # growth factor of the fit workspaces over the need that triggers it
        self.serialize_fits = not os.environ.get("GAPRO_OVERLAP_FITS")  # fit launches never overlap (see fit_launch)
        self._last_fit_done = None
        import threading

        self._ws_lock = threading.Lock()
        # optional HIP-event timing of the fit launches (bench.py): list of (start_event, end_event, flops)
        self.profile_fit = False
        self.fit_events = []
        # with profile_fit: torch events (current stream = the stream the kernels are launched on) around the batched
        # partition calls: dicts with 'prepare', 'pool', 'broadcast' -> (start, end) and the points they cover
        self.part_events = []
        self.profile_stages = False  # bench.py --stage-times: synchronising per-stage wall clock
        self.stage_times = {}
        self._pin_cache = {}
        self._pin_events = {}
        self._ws = {}  # slot -> fit workspace (device, float64)
        self._keep = {}

    def _sh(self) -> C.c_void_p:
        """raw handle of the backend's current stream"""
        return C.c_void_p(self.be.current_stream().cuda_stream)

    # ------------------------------------------------------------------ batched partition plumbing
    def _task_array(self, jobs: Sequence[SceneJob]):
        """One gapro_scene_task per scene (host ctypes array + its device mirror), created once per batch and
        refilled stage by stage."""
        tasks = (SceneTask * len(jobs))()
        for t, job in zip(tasks, jobs):
            t.n_points = job.n_points
            t.coords, t.feats, t.spp = job.coords.data_ptr(), job.feats.data_ptr(), job.spp.data_ptr()
        d_tasks = self.be.empty(len(jobs) * C.sizeof(SceneTask))
        return tasks, d_tasks

    @staticmethod
    def _carve(sizes, align=256):
        """Offsets of consecutive `align`-aligned regions of the given byte sizes, and the total."""
        offs, tot = [], 0
        for sz in sizes:
            offs.append(tot)
            tot += (int(sz) + align - 1) // align * align
        return offs, max(tot, align)

    def _part_event(self, jobs, name):
        if not self.profile_fit:
            return None
        e0 = self.be.event(enable_timing=True)
        e0.record(self.be.current_stream())
        return dict(name=name, start=e0, points=sum(j.n_points for j in jobs), feat_dim=int(jobs[0].feats.shape[1]))

    def _part_event_end(self, ev):
        if ev is not None:
            ev["end"] = self.be.event(enable_timing=True)
            ev["end"].record(self.be.current_stream())
            self.part_events.append(ev)

    # ------------------------------------------------------------------ stage A
    def _prepare_finish(self, job: SceneJob, hdr: SceneHeader):
        if hdr.status != 0:
            msg = "scene has a non-finite coordinate or feature" if hdr.status == _lib.GAPRO_ERR_NOT_FINITE else \
                "superpoint id range [%d, %d] exceeds the rank table (%d slots)" % (hdr.spp_min, hdr.spp_max,
                                                                                    job.host["range_cap"])
            job.error = _lib.GaproError(int(hdr.status), msg)
            if self.strict:
                raise job.error
            return
        job.header = hdr
        job.n_spps = int(hdr.n_spps)
        job.dev.pop("prep_ws", None)
        # boxes = cat(instance, wall, floor) with torch's dtype promotion (gen_ps_utils.py:317-345); the instance
        # and wall rows do not depend on the scene statistics and are assembled once per job
        st = job.host.get("_static")
        if st is None:
            nw = len(job.wall_box)
            st = (np.concatenate([job.instance_box.astype(np.float64), job.wall_box.astype(np.float64),
                                  np.zeros((1, 6))], 0),
                  np.ascontiguousarray(np.concatenate([job.instance_cls,
                                                       np.full(nw + 1, job.instance_classes, dtype=np.int64)])),
                  np.concatenate([job.instance_box_volume.astype(np.float64), job.wall_box_volume.astype(np.float64),
                                  np.zeros(1)]))
            dict.__setitem__(job.host, "_static", st)
        mn, mx = hdr.coord_min, hdr.coord_max
        boxes, vol = st[0].copy(), st[2].copy()
        boxes[-1] = (mn[0], mn[1], mn[2], mx[0], mx[1], mn[2] + job.ground_h)
        floor = boxes[-1]
        vol[-1] = max(floor[3] - floor[0], 0.001) * max(floor[4] - floor[1], 0.001) * max(floor[5] - floor[2], 0.001)
        job.boxes, job.boxes_cls, job.boxes_volume = boxes, st[1], vol

    def _prepare(self, job: SceneJob):
        """Single-scene, blocking form (tests)."""
        self._prepare_all([job])

    def _prepare_all(self, jobs: Sequence[SceneJob]):
        """Scene statistics + dense superpoint ranks of every scene: one launch per kernel, one sync."""
        lib, be = self.lib, self.be
        tasks, d_tasks = self._task_array(jobs)
        D = int(jobs[0].feats.shape[1])
        caps = [int(self.spp_range_cap) if self.spp_range_cap else max(4 * j.n_points, 1 << 20) for j in jobs]
        ws_off, ws_tot = self._carve([lib.gapro_partition_prepare_workspace_bytes(j.n_points, c)
                                      for j, c in zip(jobs, caps)])
        prep_ws = be.empty(ws_tot)
        inv_off, inv_tot = self._carve([4 * j.n_points for j in jobs])
        spp_inv_all = be.empty(inv_tot)
        hsz = C.sizeof(SceneHeader)
        d_headers = be.empty(len(jobs) * hsz)
        pinned = self._pinned("headers%x" % int(be.current_stream().cuda_stream), len(jobs) * hsz)
        for t, job, cap, wo, io in zip(tasks, jobs, caps, ws_off, inv_off):
            t.spp_inv = job.dev.spec("spp_inv", spp_inv_all, io, 4 * job.n_points, be.i32, (job.n_points,))
            dict.__setitem__(job.host, "range_cap", cap)
            t.prepare_ws, t.spp_range_cap = prep_ws.data_ptr() + wo, cap
        ev = self._part_event(jobs, "prepare")
        self.ctx.check(lib.gapro_partition_prepare_batch(
            self.ctx.handle, self._sh(), len(jobs), D, C.cast(tasks, C.c_void_p), _ptr(d_tasks),
            _ptr(d_headers), _ptr(pinned)))
        self._part_event_end(ev)
        be.current_stream().synchronize()  # one sync for the whole batch
        raw = pinned.numpy()
        for i, job in enumerate(jobs):
            self._prepare_finish(job, SceneHeader.from_buffer_copy(raw[i * hsz:(i + 1) * hsz].tobytes()))
        good = [i for i, job in enumerate(jobs) if job.error is None]
        if len(good) != len(jobs):  # non-strict: the rejected scenes leave the batch here
            tasks2 = (SceneTask * max(len(good), 1))()
            for k, i in enumerate(good):
                tasks2[k] = tasks[i]
            tasks = tasks2
            d_tasks = be.empty(max(len(good), 1) * C.sizeof(SceneTask))
        return tasks, d_tasks

    # ------------------------------------------------------------------ stage B
    def _pool(self, job: SceneJob, feats_spp_all, stage=None, off: int = 0):
        """Single-scene form (tests)."""
        tasks, d_tasks = self._task_array([job])
        tasks[0].spp_inv = job.spp_inv.data_ptr()
        self._pool_all([job], tasks, d_tasks, feats_spp_all, stage)

    def _pool_all(self, jobs: Sequence[SceneJob], tasks, d_tasks, feats_spp_all, stage=None):
        """Fused membership + pooling of every scene: one launch per kernel; the two small tables the host
        scheduler needs (occ_bits, n_bbs) of all scenes come back in ONE device-to-host copy (no sync here)."""
        lib, be = self.lib, self.be
        D = int(jobs[0].feats.shape[1])
        # boxes of every scene: one pinned staging buffer, one upload
        box_off, box_tot = self._carve([j.boxes.nbytes for j in jobs], 64)
        slot = "%x" % int(be.current_stream().cuda_stream)
        h_boxes = self._pinned("boxes" + slot, box_tot)
        hb = h_boxes.numpy()
        for job, bo in zip(jobs, box_off):
            hb[bo:bo + job.boxes.nbytes] = job.boxes.view(np.uint8).reshape(-1)
        d_boxes = be.empty(box_tot)
        d_boxes.copy_(h_boxes[:box_tot], non_blocking=True)
        # host-visible tables [occ_bits | n_bbs] per scene, laid out exactly like the pinned staging area
        tab_sizes = [j.n_spps * (((j.n_boxes + 63) // 64) * 8 + 4) for j in jobs]
        tab_off, tab_tot = self._carve(tab_sizes, 16)
        d_tables = be.empty(tab_tot)
        # integer tallies [feat_sum i64 | occ_count i32 | point_count i32] per scene
        tal_sizes = [j.n_spps * (8 * D + 4 * j.n_boxes + 4) for j in jobs]
        tal_off, tal_tot = self._carve(tal_sizes, 16)
        d_tallies = be.empty(tal_tot)
        if stage is None:
            stage = be.pinned(tab_tot)
        fbase, fstride = feats_spp_all.data_ptr(), 4 * D
        for t, job, bo, to, ao in zip(tasks, jobs, box_off, tab_off, tal_off):
            S, B = job.n_spps, job.n_boxes
            W = (B + 63) // 64
            d = job.dev
            t.boxes = d.spec("boxes", d_boxes, bo, job.boxes.nbytes, be.f64, (B, 6))
            t.feat_sum = d.spec("feat_sum", d_tallies, ao, 8 * S * D, be.i64, (S, D))
            t.occ_count = d.spec("occ_count", d_tallies, ao + 8 * S * D, 4 * S * B, be.i32, (S, B))
            t.point_count = d.spec("point_count", d_tallies, ao + 8 * S * D + 4 * S * B, 4 * S, be.i32, (S,))
            t.occ_bits = d.spec("occ_bits", d_tables, to, 8 * S * W, be.i64, (S, W))
            t.n_bbs = d.spec("n_bbs", d_tables, to + 8 * S * W, 4 * S, be.i32, (S,))
            t.feats_spp = fbase + job.feats_row_base * fstride
            dict.__setitem__(d, "feats_spp", None)
            d._specs["feats_spp"] = (feats_spp_all.view(be.u8).view(-1), job.feats_row_base * fstride, S * fstride,
                                     be.f32, (S, D))
            dict.pop(d, "feats_spp", None)
            t.n_boxes, t.n_spps = B, S
            t.fixed_shift, t.thresh_spp_occu = int(job.header.fixed_shift), float(job.thresh_spp_occu)
            h = job.host
            h.spec("occ_bits_pin", stage, to, 8 * S * W, be.i64, (S, W))
            h.spec("n_bbs_pin", stage, to + 8 * S * W, 4 * S, be.i32, (S,))
        ev = self._part_event(jobs, "pool")
        self.ctx.check(lib.gapro_partition_pool_batch(self.ctx.handle, self._sh(), len(jobs), D,
                                                      C.cast(tasks, C.c_void_p), _ptr(d_tasks)))
        self._part_event_end(ev)
        stage[:tab_tot].copy_(d_tables, non_blocking=True)

    # ------------------------------------------------------------------ stage C
    def _host_threads(self):
        if getattr(self, "_host_pool", None) is None:
            import concurrent.futures as cf

            self._host_pool = cf.ThreadPoolExecutor(max_workers=max(2, min(16, (os.cpu_count() or 4) // 8)))
        return self._host_pool

    def _schedule(self, job: SceneJob):
        h = job.host
        h["occ_bits"] = np.ascontiguousarray(h["occ_bits_pin"].numpy().view(np.uint64))
        h["n_bbs"] = np.ascontiguousarray(h["n_bbs_pin"].numpy())
        sched = C.c_void_p()
        rc = self.lib.gapro_schedule_build(job.n_spps, job.n_boxes, _ptr(job.boxes), _ptr(h["occ_bits"]),
                                           _ptr(h["n_bbs"]), C.byref(sched))
        if rc != 0:
            raise _lib.GaproError(rc, "gapro_schedule_build")
        job.schedule = sched
        cnt = ScheduleCounts()
        self.lib.gapro_schedule_get_counts(sched, C.byref(cnt))
        job.counts = cnt

    # ------------------------------------------------------------------ run
    def run(self, jobs: Sequence[SceneJob], keep_debug: bool = False):
        """Process a batch of scenes; fills job.outputs = (sem i32[N], inst i32[N], prob f32[N], mu f32[S],
        var f32[S]) as device tensors (same lengths as the reference returns)."""
        return self._finish(self._start(jobs, keep_debug))

    kSlots = 3  # pipeline streams = slots of per-batch buffers (see _stream_batches)

    def _ensure_streams(self):
        if not hasattr(self, "_streams"):
            self._streams = [self.be.new_stream() for _ in range(self.kSlots)]
            for st in self._streams:
                self._ws.setdefault("s%x" % int(st.cuda_stream), None)

    def run_pipelined(self, batches: Sequence[Sequence[SceneJob]]):
        """Several batches back to back, software-pipelined.  The fit workgroups fill every CU for the whole
        launch and the short partition / broadcast kernels cannot be dispatched beside them (measured: they
        wait until CUs drain, whatever the stream priorities), so the order is built around that (_stream_batches):

            partition(i+1) in the tail of fit(i-1) -> launch fit(i) -> while it runs, on the host: schedule(i+1),
            merge(i-1); broadcast(i-1) is enqueued without waiting for it

        Same results as run() batch by batch."""
        self._ensure_streams()
        outs = []
        # inputs produced on the caller's stream are ordered before both pipeline streams ONCE: an event on
        # the (legacy default) stream recorded per batch would also wait for every blocking stream
        ready = self.be.current_stream().record_event()
        for st in self._streams:
            st.wait_event(ready)
        outs = list(self._stream_batches(iter(batches)))
        for st in self._streams:
            self.be.current_stream().wait_stream(st)
        return outs

    def warmup(self):
        """One small launch through every fit kernel family and the result path (a driver calls this while its first
        scenes are being read): the first launch of a process pays for code-object loading, kernel attributes and the
        allocator's first blocks -- ~0.2 s that would otherwise sit in front of the first batch."""
        rng = np.random.default_rng(0)
        sizes = [4, 12, 20, 30, 50, 75, 150, 280]
        feats = self.be.from_numpy(rng.normal(size=(2 * sum(sizes), 6)).astype(np.float32))
        descs = (FitDesc * len(sizes))()
        idx, io, oo, base = [], 0, 0, 0
        for k, m in enumerate(sizes):
            d = descs[k]
            d.m1, d.m2, d.t, d.b1, d.b2, d.scene = m, m, 4, 0, 1, 0
            d.idx_offset, d.out_offset, d.ws_offset = io, oo, 0
            idx.append(np.arange(base, base + 2 * m, dtype=np.int32))
            idx.append(np.arange(base, base + 4, dtype=np.int32))
            io += 2 * m + 4
            oo += 4
            base += 2 * m
        it, prof = self.opt.training_iter, self.profile_fit
        self.opt.training_iter, self.profile_fit = 2, False
        try:
            self.fit_descs(feats, descs, len(sizes), np.concatenate(idx), oo, raise_on_failure=False)
        finally:
            self.opt.training_iter, self.profile_fit = it, prof

    def prealloc_workspace(self, n_bytes: int, slot: str = "shared"):
        """Allocate the fit workspace ahead of its first use (a driver calls this from a helper thread while the first
        scenes are still being read: the hipMalloc + clear of ~25 GB takes 1.2 .. 2.4 s, a third of what a worker needs
        for the whole ScanNet train split).  A later need beyond this size grows it as usual."""
        with self.be.device_ctx():
            self._workspace(slot, max(1, int(n_bytes) // 8), headroom=1.0)

    def _workspace(self, slot: str, n_doubles: int, headroom: Optional[float] = None):
        """Grow-only fit workspace per pipeline slot.  Every registered slot grows together and new memory is
        touched once here: a slot first used inside a timed region would otherwise pay the allocation and the
        first-touch mapping of several GB inside its fit kernel."""
        with self._ws_lock:
            return self._workspace_locked(slot, n_doubles, headroom)

    def _workspace_locked(self, slot: str, n_doubles: int, headroom: Optional[float]):
        self._ws.setdefault(slot, None)
        cur = self._ws[slot]
        if cur is None or cur.numel() < n_doubles:
            if self._last_fit_done is not None:
                # a launch that may still be running works in the memory about to be released (the fit kernels run on
                # library-owned streams the caching allocator knows nothing about)
                self._last_fit_done.synchronize()
            # 30 % headroom: the need of a batch of 256 scenes varies by ~10 %, and growing means a hipMalloc of
            # 10 .. 20 GB per slot plus its fill behind the running fit kernels -- 2 s on a freshly booted box, inside
            # whatever step first exceeds the old size (bench.py's sporadic 980 ms steps against 760 ms launches)
            size = int(n_doubles * (self.workspace_headroom if headroom is None else headroom)) + 1024
            if slot.endswith("retry"):  # a handful of timed-out fits: exactly their need, for one launch
                self._ws[slot] = self.be.zeros(n_doubles + 1024, self.be.f64)
                self.be.current_stream().synchronize()
                return self._ws[slot][:n_doubles]
            if self.trace is not None:
                import time as _time
                self.trace.append((_time.perf_counter(), 0, "workspace grows to %.1f GB per slot" % (size * 8 / 1e9)))
            for k in list(self._ws):  # the requested slot, and every slot that is in use (they grow together)
                if k != slot and (self._ws[k] is None or k.endswith("retry")):
                    continue
                if self._ws[k] is None or self._ws[k].numel() < size:
                    self._ws[k] = None  # release before growing
                    self._ws[k] = self.be.zeros(size, self.be.f64)
            # the fill runs on the allocating thread's stream; the fit kernels that use this memory run on other
            # streams (another pipeline slot's, the library's): nobody may get the tensor before the fill is done
            self.be.current_stream().synchronize()
            if self.trace is not None:
                import time as _time
                self.trace.append((_time.perf_counter(), 0, "workspace ready"))
        return self._ws[slot][:n_doubles]

    def run_stream(self, batches):
        """run_pipelined for an ITERATOR of batches (e.g. a dataset being read from disk): yields the outputs of
        every batch, in order, one batch behind the one being launched.  The consumer may use the yielded
        tensors on the current stream right away (they are ordered after the pipeline's streams)."""
        self._ensure_streams()
        ready = self.be.current_stream().record_event()
        for st in self._streams:
            st.wait_event(ready)
        for i, out in enumerate(self._stream_batches(iter(batches))):
            self.be.current_stream().wait_stream(self._streams[i % self.kSlots])
            yield out

    def _stream_batches(self, it):
        """The software pipeline of run_pipelined / run_stream over an iterator (lookahead of one batch).

        Batch k lives on stream / buffer slot k % kSlots.  Iteration i, entered while fit(i-1) runs:
            partition(i+1)   the host blocks on its two round trips; the kernels cannot be dispatched beside a fit that
                             holds every CU: by the host-side trace (bench.py --trace) they complete when the last
                             workgroups of fit(i-1) end, also on a stream that is not ordered behind that fit (three
                             slots; with two, batch i+1 shared the stream of batch i-1, which gapro_svgp_fit_batch
                             joins its kernels back into) and with more hardware queues (GPU_MAX_HW_QUEUES = 8 / 16:
                             slower); ~10 ms of partition kernels and round trips stay between two fits
            launch fit(i)
            pull batch i+2, schedule(i+1), finish(i-1): host work while fit(i) runs
        Queueing fit(i+1) behind fit(i) instead (two fits in flight, five slots) starves the partition kernels until
        BOTH have drained: measured, 64 ms of idle GPU every second step, 287 vs 300 scenes/s."""
        S = self.kSlots

        def on(k, fn, *a):
            with self.be.stream(self._streams[k % S]):
                return fn(*a)

        # a batch is pulled from the iterator under the pipeline stream that will process it: whatever device work
        # building its jobs enqueues (a dtype cast, .contiguous() of a strided input in make_job) is then ordered
        # before its partition kernels, for every batch and not only the first
        cur = on(0, next, it, None)
        if cur is None:
            return
        i = 0
        cur_state = on(0, self._partition, cur, False)
        on(0, self._schedule_all, cur_state)
        # The first fit launch goes out before the second batch is even asked for (round 5): with the native feeder the
        # iterator hands the first batch over ~0.3 s into a run and blocks ~0.2 s for the second -- time the GPU would
        # sit idle in the plain order below, which partitions batch i + 1 before it launches batch i.  (Round 4 tried the
        # same and measured nothing: the Python loaders, not the order, were what the first launch waited for.)
        on(0, self._launch, cur_state)
        nxt = on(1, next, it, None)
        prev_state = None
        finish_late = bool(os.environ.get("GAPRO_FINISH_LATE"))  # A/B: round 5's order (merge after the pull)
        while cur_state is not None:
            nxt_state = on(i + 1, self._partition, nxt, False) if nxt is not None else None
            if i > 0:
                on(i, self._launch, cur_state)
            # everything below is host work that runs while the fit just launched occupies the GPU: the merge of the
            # previous batch, fetching the batch after next from the iterator (building jobs, reading / uploading
            # scenes), the schedule of the next batch.  The merge comes FIRST (round 6): fit(i-1) has ended -- the
            # partition kernels above only ran once it had drained -- and the pull can block for a whole batch of
            # reads, which in a worker's first second kept finished labels waiting for 0.3 .. 0.5 s
            if prev_state is not None and not finish_late:
                yield on(i - 1, self._finish, prev_state, False)
            nxt = on(i + 2, next, it, None) if nxt is not None else None
            if nxt_state is not None:
                on(i + 1, self._schedule_all, nxt_state)
            if prev_state is not None and finish_late:
                yield on(i - 1, self._finish, prev_state, False)
            prev_state, cur_state = cur_state, nxt_state
            i += 1
        yield on(i - 1, self._finish, prev_state, True)

    def _pinned(self, tag: str, nbytes: int):
        """Growable page-locked staging buffers, reused across batches (hipHostMalloc is slow)."""
        buf = self._pin_cache.get(tag)
        if buf is None or buf.numel() < nbytes:
            buf = self.be.pinned(max(nbytes, 1 << 16))
            self._pin_cache[tag] = buf
        return buf

    def _start(self, jobs: Sequence[SceneJob], keep_debug: bool = False):
        """Stages A-D: everything up to and including the (asynchronous) fit launch."""
        state = self._partition(jobs, keep_debug)
        self._schedule_all(state)
        self._launch(state)
        return state

    def _marker(self, jobs):
        import time as _time
        t = [_time.perf_counter()]

        def _mark(name):
            if self.trace is not None:  # host-side timeline, no synchronisation added
                self.trace.append((_time.perf_counter(), id(jobs) & 0xFFFF, name))
            if self.profile_stages:
                self.be.synchronize()
                now = _time.perf_counter()
                self.stage_times[name] = self.stage_times.get(name, 0.0) + (now - t[0])
                t[0] = now

        return _mark

    def _partition(self, jobs: Sequence[SceneJob], keep_debug: bool = False):
        """Stages A-B on the current stream: superpoint ids, pooled features, occupancy tables on the host."""
        _mark = self._marker(jobs)
        stream = self.be.current_stream()
        slot = "s%x" % int(stream.cuda_stream)
        _mark("start")
        all_jobs = list(jobs)
        tasks, d_tasks = self._prepare_all(jobs)
        jobs = [j for j in all_jobs if j.error is None]
        _mark("A prepare")
        if not jobs:
            return dict(jobs=[], all_jobs=all_jobs, stream=stream, keep_debug=keep_debug, mark=_mark, slot=slot,
                        pending=None, n_fits=0, n_out=0, tasks=tasks, d_tasks=d_tasks, feats_spp_all=None)
        D = int(jobs[0].feats.shape[1])
        base = 0
        for job in jobs:
            if int(job.feats.shape[1]) != D:
                raise ValueError("all scenes of a batch must share the feature dimension")
            job.feats_row_base = base
            base += job.n_spps
        feats_spp_all = self.be.empty_typed((base, D), self.be.f32)
        # one pinned staging area for the tables the host scheduler needs from every scene
        need = sum(job.n_spps * (((job.n_boxes + 63) // 64) * 8 + 4) + 16 for job in jobs)
        stage = self._pinned(slot + "tables", need)
        self._pool_all(jobs, tasks, d_tasks, feats_spp_all, stage)
        stream.synchronize()  # one sync: pooled tables of every scene are on the host
        _mark("B pool")
        return dict(jobs=jobs, all_jobs=all_jobs, stream=stream, keep_debug=keep_debug, mark=_mark,
                    feats_spp_all=feats_spp_all, slot=slot, pending=None, n_fits=0, n_out=0, tasks=tasks, d_tasks=d_tasks)

    def _schedule_all(self, state):
        """Stage C (host only): static pair schedule of every scene, fit descriptors of the whole batch."""
        lib = self.lib
        jobs, _mark = state["jobs"], state["mark"]
        if not jobs:
            return
        # one host thread per scene (gapro_schedule_build is host C++ behind ctypes: no GIL): 256 scenes took 0.45 s on
        # the main thread, more than half of what a fit launch of that batch lasts -- the product path's steady state
        # was bound by this thread, not by the GPU
        if len(jobs) >= 8:
            list(self._host_threads().map(self._schedule, jobs))
        else:
            for job in jobs:
                self._schedule(job)
        _mark("C schedule")
        n_fits = sum(j.counts.n_fits for j in jobs)
        n_idx = sum(j.counts.n_fit_idx for j in jobs)
        n_out = sum(j.counts.n_fit_out for j in jobs)
        descs = (FitDesc * max(n_fits, 1))()
        h_idx = np.zeros(max(n_idx, 1), dtype=np.int32)
        fo = io = oo = 0
        for si, job in enumerate(jobs):
            job.fit_base, job.out_base = fo, oo
            if job.counts.n_fits:
                rc = lib.gapro_schedule_export_fits(
                    job.schedule, job.feats_row_base, io, oo, si,
                    C.cast(C.byref(descs, fo * C.sizeof(FitDesc)), C.c_void_p), _ptr(h_idx[io:]))
                if rc != 0:
                    raise _lib.GaproError(rc, "gapro_schedule_export_fits")
            fo += job.counts.n_fits
            io += job.counts.n_fit_idx
            oo += job.counts.n_fit_out
        _mark("C export")
        state.update(n_fits=n_fits, n_out=n_out, descs=descs, h_idx=h_idx)

    def _launch(self, state):
        """Stage D: one (asynchronous) launch for every fit of every scene."""
        if state["n_fits"]:
            state["pending"] = self.fit_launch(state["feats_spp_all"], state["descs"], state["n_fits"], state["h_idx"],
                                               state["n_out"], keep_debug=state["keep_debug"], slot=state["slot"],
                                               scene_keys=[j.scene_key for j in state["jobs"]])
        state["mark"]("D launched")

    def _finish(self, state, sync: bool = True):
        """Stages E-F: wait for the fit results, ordered merge on the host, broadcast on the device.  With
        sync=False the broadcast kernels are only enqueued (the outputs are ordered on the current stream)."""
        lib, ctx, be = self.lib, self.ctx, self.be
        jobs, keep_debug, _mark = state["jobs"], state["keep_debug"], state["mark"]
        _mark("finish")
        if not jobs:  # every scene of the batch was rejected (non-strict mode)
            self.last_stats = dict(n_fits=0, n_fit_out=0, fit=None)
            return [None for _ in state["all_jobs"]]
        res = self.fit_collect(state["pending"], raise_on_failure=False) if state["pending"] is not None else None
        if res is not None:
            res = self._retry_timeouts(res, state["feats_spp_all"], state["descs"], state["h_idx"], state["n_out"],
                                       slot=state["slot"], scene_keys=[j.scene_key for j in jobs])
        _mark("D fit")
        if res is not None and (res["status"] != 0).any():
            # per-fit status -> per-scene failure: only the scenes that own a failed fit are lost
            for job in jobs:
                st = res["status"][job.fit_base:job.fit_base + job.counts.n_fits]
                if (st != 0).any():
                    k = int(np.nonzero(st)[0][0])
                    job.error = _lib.GaproError(int(st[k]), "GP fit %d of %d of the scene failed (%d failed in all)"
                                                % (k, job.counts.n_fits, int((st != 0).sum())))
            if self.strict:
                raise next(j.error for j in jobs if j.error is not None)
        tot_s = sum(job.n_spps for job in jobs)
        tables = self._pinned(state["slot"] + "labels", tot_s * 20)
        busy = self._pin_events.pop(state["slot"] + "labels", None)
        if busy is not None:
            busy.synchronize()  # the previous batch of this slot has uploaded its label tables
        d_tables = be.empty(tot_s * 20)
        tab_np = tables.numpy()
        views, off = [], 0
        offs = []
        for job in jobs:
            offs.append(off)
            off += 20 * job.n_spps

        def merge_one(arg):
            job, off = arg
            S = job.n_spps
            sem_spp = tab_np[off:off + 4 * S].view(np.int32)
            inst_spp = tab_np[off + 4 * S:off + 8 * S].view(np.int32)
            prob_spp = tab_np[off + 8 * S:off + 12 * S].view(np.float32)
            mu_spp = tab_np[off + 12 * S:off + 16 * S].view(np.float32)
            var_spp = tab_np[off + 16 * S:off + 20 * S].view(np.float32)
            if job.counts.n_fits:
                a, b = job.out_base, job.out_base + job.counts.n_fit_out
                pn, lb, mu, var = (res["probs_new"][a:b], res["labels"][a:b], res["mu"][a:b], res["var"][a:b])
            else:
                pn = lb = mu = var = None
            rc = lib.gapro_schedule_merge(job.schedule, _ptr(pn), _ptr(lb), _ptr(mu), _ptr(var),
                                          _ptr(job.boxes_cls), _ptr(job.boxes_volume), len(job.instance_box),
                                          job.instance_classes, _ptr(sem_spp), _ptr(inst_spp), _ptr(prob_spp),
                                          _ptr(mu_spp), _ptr(var_spp))
            if rc != 0:
                raise _lib.GaproError(rc, "gapro_schedule_merge")
            job.host.update(sem_spp=sem_spp.copy(), inst_spp=inst_spp.copy(), prob_spp=prob_spp.copy())

        # the ordered merge of a scene is host C++ on that scene's own slices: one host thread per scene
        if len(jobs) >= 8:
            list(self._host_threads().map(merge_one, zip(jobs, offs)))
        else:
            for arg in zip(jobs, offs):
                merge_one(arg)
        views = offs
        d_tables.copy_(tables[:tot_s * 20], non_blocking=True)  # one H2D copy for the whole batch
        self._pin_events[state["slot"] + "labels"] = be.current_stream().record_event()
        tasks, d_tasks = state["tasks"], state["d_tasks"]
        out_off, out_tot = self._carve([12 * job.n_points for job in jobs], 16)
        d_out = be.empty(out_tot)  # [sem | inst | prob] per scene
        for t, job, off, oo in zip(tasks, jobs, views, out_off):
            S, n = job.n_spps, job.n_points
            sem = d_out[oo:oo + 4 * n].view(be.i32)
            ins = d_out[oo + 4 * n:oo + 8 * n].view(be.i32)
            prb = d_out[oo + 8 * n:oo + 12 * n].view(be.f32)
            t.sem_spp = d_tables.data_ptr() + off
            t.inst_spp = d_tables.data_ptr() + off + 4 * S
            t.prob_spp = d_tables.data_ptr() + off + 8 * S
            t.sem, t.inst, t.prob = sem.data_ptr(), ins.data_ptr(), prb.data_ptr()
            job.outputs = (sem, ins, prb, d_tables[off + 12 * S:off + 16 * S].view(be.f32),
                           d_tables[off + 16 * S:off + 20 * S].view(be.f32))
            if not keep_debug:
                lib.gapro_schedule_free(job.schedule)
                job.schedule = None
        ev = self._part_event(jobs, "broadcast")
        ctx.check(lib.gapro_broadcast_labels_batch(ctx.handle, self._sh(), len(jobs),
                                                   C.cast(tasks, C.c_void_p), _ptr(d_tasks)))
        self._part_event_end(ev)
        # the task array must outlive the (possibly delayed) upload enqueued above
        self._keep[state["slot"]] = (tasks, d_tasks)
        if sync:
            be.current_stream().synchronize()
        _mark("E+F merge/broadcast")
        self.last_stats = dict(n_fits=state["n_fits"], n_fit_out=state["n_out"], fit=res)
        for j in jobs:
            if j.error is not None:  # merged from a failed fit's garbage: not a result
                j.outputs = None
        return [j.outputs if j.error is None else None for j in state["all_jobs"]]

    # ------------------------------------------------------------------ stage D
    def fit_descs(self, feats_spp, descs, n_fits: int, h_idx: np.ndarray, n_out: int,
                  init_mean: Optional[np.ndarray] = None, keep_debug: bool = False, raise_on_failure: bool = True):
        """Launch a batch of fits and wait for the results (numpy arrays)."""
        res = self.fit_collect(self.fit_launch(feats_spp, descs, n_fits, h_idx, n_out, init_mean, keep_debug),
                               raise_on_failure=False)
        res = self._retry_timeouts(res, feats_spp, descs, h_idx, n_out, init_mean=init_mean)
        if raise_on_failure and (res["status"] != 0).any():
            bad = int(np.nonzero(res["status"])[0][0])
            raise _lib.GaproError(int(res["status"][bad]), "fit %d of %d failed" % (bad, n_fits))
        return res

    def reproducibility_probe(self, feats_spp, descs, n_fits: int, h_idx: np.ndarray, n_out: int, res=None,
                              init_mean: Optional[np.ndarray] = None, rel: float = 1e-11):
        """How far do a fit's outputs move when nothing but rounding-sized quantities change?  The same fits once more
        with the jitter on K_ZZ's diagonal scaled by (1 + rel) -- 1e-15 absolute on entries of size 0.7, a few ulps --
        compared with `res` (the unperturbed run; computed here when not given).  Returns (dv, dp) per fit: the largest
        relative change of sigma^2 and absolute change of p over the fit's test superpoints.  Well-behaved fits come
        back at 1e-7 .. 1e-6 (float32 output rounding and a little more); a fit beyond REPRO_SOFT amplifies last-bit differences ~1e9-fold over
        its fifty Adam steps, and NO float64 implementation reproduces its sigma^2 to north_star's 1e-4 -- the reference
        against itself on another BLAS included (on the S3DIS-shaped test scene: 2 of 66 fits, the
        two the oracle's own implementations disagree on).  Twice the work of the fits: a caller's choice
        (fit_gp_spp_batch(..., reproducibility_probe=True)), not the default.  (Perturbing the initial variational mean
        instead is useless as a probe: at 1e-13 it moves EVERY fit's sigma^2 by 1e-4 .. 9e-2 -- the optimisation is
        that sensitive to its starting point, which is why the reference's unseeded 1e-3 * randn start is switched off
        by default -- LABNOTES R6.)"""
        if res is None:
            res = self.fit_descs(feats_spp, descs, n_fits, h_idx, n_out, init_mean=init_mean, raise_on_failure=False)
        jit = float(self.opt.jitter)
        self.opt.jitter = jit * (1.0 + rel)
        try:
            r2 = self.fit_descs(feats_spp, descs, n_fits, h_idx, n_out, init_mean=init_mean, raise_on_failure=False)
        finally:
            self.opt.jitter = jit
        dv, dp = np.zeros(n_fits), np.zeros(n_fits)
        for k in range(n_fits):
            d = descs[k]
            a, b = int(d.out_offset), int(d.out_offset) + int(d.t)
            if b > a:
                v1, v2 = res["var"][a:b].astype(np.float64), r2["var"][a:b].astype(np.float64)
                dv[k] = float(np.max(np.abs(v1 - v2) / np.maximum(np.abs(v1), 1e-30)))
                dp[k] = float(np.max(np.abs(res["probs"][a:b].astype(np.float64) - r2["probs"][a:b].astype(np.float64))))
        return dv, dp

    def _retry_timeouts(self, res, feats_spp, descs, h_idx, n_out, init_mean=None, slot="s0", scene_keys=None):
        """Fits whose status is GAPRO_ERR_TIMEOUT are launched once more with the cluster kernel switched off (debug
        bit 3 of gapro_fit_options.reserved: the LDS-staged kernel up to M_p = 512, the generic kernel beyond -- one
        workgroup each, no cross-workgroup barrier that could time out) and their outputs, status and loss replace the
        first attempt's.  A timeout says that a member of the fit's cluster was not given a CU within
        GAPRO_CLUSTER_BARRIER_TIMEOUT_MS; the arithmetic never ran to an end, so there is nothing deterministic about
        the failure and the second attempt computes what the first one would have (the per-fit result does not
        depend on the kernel beyond float64 round-off: tests/test_fit_gpu.py).  Same descriptors, same index array,
        same output offsets; only the workspace is the retry's own."""
        bad = np.nonzero(res["status"] == _lib.GAPRO_ERR_TIMEOUT)[0]
        if not len(bad) or not self.retry_timeouts:
            return res
        sub = (FitDesc * len(bad))()
        for k, i in enumerate(bad):
            C.memmove(C.byref(sub, k * C.sizeof(FitDesc)), C.byref(descs, int(i) * C.sizeof(FitDesc)), C.sizeof(FitDesc))
        print("[gapro_amd] %d GP fit(s) timed out at a cluster barrier; retrying them on one workgroup each"
              % len(bad), file=sys.stderr)
        old, prof = self.opt.reserved, self.profile_fit
        self.opt.reserved = (int(old) | 8) & ~32768
        self.profile_fit = False  # the retry is not a step of whoever is timing the launches
        try:
            r2 = self.fit_collect(self.fit_launch(feats_spp, sub, len(bad), h_idx, n_out, init_mean, slot=slot + "retry",
                                                  scene_keys=scene_keys), raise_on_failure=False)
        finally:
            self.opt.reserved, self.profile_fit = old, prof
            with self._ws_lock:
                self._ws.pop(slot + "retry", None)  # its workspace goes back to the allocator
        for k, i in enumerate(bad):
            d = descs[int(i)]
            a, b = int(d.out_offset), int(d.out_offset) + int(d.t)
            for key in ("probs", "probs_new", "mu", "var", "labels"):
                res[key][a:b] = r2[key][a:b]
            res["status"][i] = r2["status"][k]
            res["loss"][i] = r2["loss"][k]
            res["cond"][i] = r2["cond"][k]
        self.timeout_retries += len(bad)
        res["retried"] = [int(i) for i in bad]
        return res

    def fit_launch(self, feats_spp, descs, n_fits: int, h_idx: np.ndarray, n_out: int,
                   init_mean: Optional[np.ndarray] = None, keep_debug: bool = False, slot: str = "s0",
                   scene_keys: Optional[Sequence[int]] = None):
        lib, ctx, be = self.lib, self.ctx, self.be
        D = int(feats_spp.shape[1])
        ws_bytes = int(lib.gapro_fit_plan_workspace(C.cast(descs, C.c_void_p), n_fits, D))
        if init_mean is None and self.init_mean_std > 0.0:
            # gpytorch: variational mean <- 0 + mean_init_std * randn on first call.  The noise of a
            # fit is a function of (seed, scene key, box pair) only, so a scene's result depends neither on the
            # batch it travels in nor on its position there, and a resumed run reproduces it.
            init_mean = np.zeros(len(h_idx))
            for k in range(n_fits):
                d = descs[k]
                key = int(scene_keys[d.scene]) if scene_keys is not None else int(d.scene)
                rng = np.random.default_rng([self.seed, key & 0xFFFFFFFF, int(d.b1), int(d.b2)])
                m = int(d.m1 + d.m2)
                init_mean[d.idx_offset:d.idx_offset + m] = self.init_mean_std * rng.standard_normal(m)
        d_descs = be.empty(n_fits * C.sizeof(FitDesc))
        d_idx = be.from_numpy(h_idx)
        d_init = be.from_numpy(np.ascontiguousarray(init_mean, dtype=np.float64)) if init_mean is not None else None
        # ONE workspace for all pipeline slots while launches are serialised (round 4): launch i + 1 starts on the device
        # when launch i has ended, nothing on the host reads a workspace (results leave through `out` / `stat`), and every
        # kernel initialises what it reads -- slots were already reused by other fits every third launch.  A worker's
        # first batches no longer pay three hipMallocs of ~18 GB (~2 s each: the driver clears VRAM on allocation; a
        # 1201-scene job is 3.5 s of GPU work), and a 256-scene pipeline holds 14 GB of workspace instead of 42.
        # Tests that inspect trained parameters (keep_debug) and overlapping launches keep a workspace per slot.
        ws_slot = "shared" if (self.serialize_fits and not keep_debug and not slot.endswith("retry")) else slot
        ws = self._workspace(ws_slot, ws_bytes // 8)
        no = max(n_out, 1)
        # per-test-superpoint outputs in one device block: probs f32 | probs_new f32 | mu f32 | var f32 | labels u8
        out = be.empty(no * 17)
        probs = out[0:4 * no].view(be.f32)
        probs_new = out[4 * no:8 * no].view(be.f32)
        mu = out[8 * no:12 * no].view(be.f32)
        var = out[12 * no:16 * no].view(be.f32)
        labels = out[16 * no:17 * no]
        stat = be.empty(n_fits * 20)  # loss f64[n] | cond f64[n] | status i32[n]
        loss = stat[0:8 * n_fits].view(be.f64)
        cond = stat[8 * n_fits:16 * n_fits].view(be.f64)
        status = stat[16 * n_fits:20 * n_fits].view(be.i32)
        if self.profile_fit:
            tm = C.c_void_p()
            ctx.check(lib.gapro_fit_timing_create(ctx.handle, C.byref(tm)))
            ctx.check(lib.gapro_fit_timing_arm(ctx.handle, tm))
        # One fit launch at a time.  The software pipeline launches fit(i) when the partition kernels of batch i+1 have
        # completed, and those normally complete when fit(i-1) drains -- but now and then they slip in earlier, fit(i)
        # is enqueued with most of fit(i-1) still to run, and the members of its cluster fits become resident one by
        # one as CUs free up, spinning at their first barrier for hundreds of ms on CUs that fit(i-1) could use
        # (bench.py --steps 20: cluster kernels of 1.2 .. 1.5 s beside the usual 0.23 s, launches 12 % longer).
        prev = self._last_fit_done if self.serialize_fits else None
        if prev is not None:
            be.current_stream().wait_event(prev)
        # (_ex: the per-fit conditioning figure travels with the status)
        ctx.check(lib.gapro_svgp_fit_batch_ex(
            ctx.handle, self._sh(), n_fits, D, _ptr(feats_spp), _ptr(d_idx), C.cast(descs, C.c_void_p),
            _ptr(d_descs), _ptr(d_init),
            C.byref(self.opt), _ptr(ws), ws_bytes, _ptr(probs), _ptr(probs_new), _ptr(labels), _ptr(mu), _ptr(var),
            _ptr(status), _ptr(loss), _ptr(cond)))
        if self.profile_fit:
            each = fit_flops_each(descs, n_fits, D, int(self.opt.training_iter))
            raw = _desc_table(descs, n_fits)
            m = (raw[:, 0] + raw[:, 1]).copy()
            route = {int(v): int(lib.gapro_fit_route(int(v), D)) for v in np.unique(m)}
            flags = int(self.opt.reserved)
            r = np.array([route[int(v)] for v in m])
            if flags & 16:
                for v in np.unique(m):
                    mp = int(lib.gapro_fit_padded_m(int(v), D))
                    if mp >= 64 and mp % 32 == 0 and D <= 32:
                        r[m == v] = 4
            if flags & 8:  # no cluster kernel: those fits run where they ran in round 1
                for v in np.unique(m[r == 4]):
                    r[m == v] = 1 if int(v) <= 512 and D <= 32 else 2
            if flags & (1 << 20):  # no wave-per-fit kernel
                r[r == 5] = 3
            if flags & 4:
                r[r == 3] = 0
            if flags & 1:
                r[(r == 0) | (r == 3)] = 1
            is_strip, is_small, is_clus, is_wave = r == 0, r == 3, r == 4, r == 5
            self.fit_events.append(FitTiming(ctx, tm, float(each[is_strip].sum()),
                                             float(each[~(is_strip | is_small | is_clus | is_wave)].sum()),
                                             float(each[is_small].sum()), m, float(each[is_clus].sum()),
                                             float(each[is_wave].sum())))
            self.last_fit_m = m
        # the next launch is ordered behind THIS launch's kernels only, not behind the copies below
        kern_done = be.event()
        kern_done.record(be.current_stream())
        # results travel to pinned host memory on the same stream; nobody waits here
        h_out = self._pinned(slot + "fit_out", no * 17)
        h_stat = self._pinned(slot + "fit_stat", n_fits * 20)
        h_out[:no * 17].copy_(out, non_blocking=True)
        h_stat[:n_fits * 20].copy_(stat, non_blocking=True)
        done = be.event()
        done.record(be.current_stream())
        # The next launch stays ordered behind the result COPIES of this one, as in round 3.  ADVICE r03 suggested the
        # kernels' end instead (so that the host blocks for less inside the next gapro_svgp_fit_batch); measured, same
        # box, alternating, 10 steps: 322.8 / 327.3 scenes/s with it against 331.3 / 336.7 without (-2.7 %): the next
        # launch's descriptor upload and first workgroups then compete with this launch's copies and the host work
        # that follows them.  GAPRO_SERIALIZE_ON_KERNELS=1 selects the variant (A/B only).
        self._last_fit_done = kern_done if os.environ.get("GAPRO_SERIALIZE_ON_KERNELS") else done
        keep = (d_descs, d_idx, d_init, ws, out, stat, feats_spp)  # alive until the launch has finished
        return dict(done=done, h_out=h_out, h_stat=h_stat, no=no, n_fits=n_fits, ws_bytes=ws_bytes, keep=keep,
                    ws=ws if keep_debug else None, descs=descs if keep_debug else None)

    def fit_collect(self, p, raise_on_failure: bool = True):
        """Results of a launch.  The library reports a gapro_status per fit and never fails the batch; with
        raise_on_failure (the fit_gp_spp API: gpytorch raises NotPSDError / NanError there) the first failed fit
        raises, otherwise the caller reads res["status"]."""
        p["done"].synchronize()
        no, n_fits = p["no"], p["n_fits"]
        raw = p["h_out"].numpy()
        st_raw = p["h_stat"].numpy()
        st = st_raw[16 * n_fits:20 * n_fits].view(np.int32).copy()
        if raise_on_failure and (st != 0).any():
            bad = int(np.nonzero(st)[0][0])
            raise _lib.GaproError(int(st[bad]), "fit %d of %d failed" % (bad, n_fits))
        res = dict(probs=raw[0:4 * no].view(np.float32).copy(), probs_new=raw[4 * no:8 * no].view(np.float32).copy(),
                   mu=raw[8 * no:12 * no].view(np.float32).copy(), var=raw[12 * no:16 * no].view(np.float32).copy(),
                   labels=raw[16 * no:17 * no].copy(), loss=st_raw[0:8 * n_fits].view(np.float64).copy(), status=st,
                   # (max L_jj / min L_jj)^2 of every fit's last factorisation (a diagnostic: see reproducibility_probe)
                   cond=st_raw[8 * n_fits:16 * n_fits].view(np.float64).copy(), ws_bytes=p["ws_bytes"])
        if p["ws"] is not None:
            res["workspace"] = p["ws"].clone()  # the slot's workspace is reused by the next launch
            res["descs"] = p["descs"]
        p["keep"] = None
        return res
