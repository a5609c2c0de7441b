"""ctypes face of the native batch feeder (csrc/feeder.hip, include/gapro_hip.h `gapro_feed_*`).

The gen_ps driver's per-scene host work -- reference gen_ps.py:36-87 on the way in, :126-132 on the way out -- runs on
the library's own threads; Python sees batches: ``submit`` paths, ``take`` the next scenes that are loaded (records +
one device slab with their arrays), ``export`` label files, ``finish``.
"""
from __future__ import annotations

import ctypes as C
import os.path as osp
from typing import List, Optional, Sequence

import numpy as np

from . import _lib


class FeedScene(C.Structure):
    """gapro_feed_scene"""
    _fields_ = [("status", C.c_int32), ("n_points", C.c_int32), ("n_instances", C.c_int32), ("feat_dim", C.c_int32),
                ("off_coords", C.c_int64), ("off_feats", C.c_int64), ("off_spp", C.c_int64), ("off_sem", C.c_int64),
                ("off_inst", C.c_int64), ("inst_box", C.POINTER(C.c_double)), ("inst_cls", C.POINTER(C.c_double)),
                ("inst_vol", C.POINTER(C.c_double)), ("host_image", C.c_void_p)]


class FeedOut(C.Structure):
    """gapro_feed_out"""
    _fields_ = [("d_sem", C.c_void_p), ("d_inst", C.c_void_p), ("d_prob", C.c_void_p), ("d_mu", C.c_void_p),
                ("d_var", C.c_void_p), ("n_points", C.c_int64), ("n_mu", C.c_int64), ("path", C.c_char_p)]


def scene_paths(filename: str, data_root: str, use_deepfeat: bool = False, deepfeat_folder: Optional[str] = None):
    """(scene, superpoints, alignment, features or None) files of a scene, as gen_ps.py:37-64 names them."""
    scan = filename.split("/")[-1][:12]
    return (filename, osp.join(data_root, "superpoints", scan + ".pth"),
            osp.join(data_root, "scans_transform", scan, scan + ".txt"),
            osp.join(deepfeat_folder, scan + ".pth") if use_deepfeat else None)


class TakenScene:
    """One scene of a batch handed out by the feed: status, sizes, offsets into the batch's slab, GT boxes (copied)."""
    __slots__ = ("filename", "status", "n_points", "n_instances", "feat_dim", "off", "instance_cls", "instance_box",
                 "instance_box_volume", "host_image")


class NativeFeeder:
    def __init__(self, device: int, n_threads: int, budget_bytes: int = 6 << 30, feat_dim: int = 6):
        self.lib = _lib.load()
        h = C.c_void_p()
        rc = self.lib.gapro_feed_create(int(device), int(n_threads), int(budget_bytes), int(feat_dim), C.byref(h))
        if rc != 0:
            raise _lib.GaproError(rc, "gapro_feed_create")
        self.handle = h
        self.device = int(device)
        self.names: List[str] = []  # submitted file names, in order
        self.taken = 0
        self.exported = 0

    def _check(self, rc):
        if rc != 0:
            raise _lib.GaproError(rc, (self.lib.gapro_feed_last_error(self.handle) or b"").decode())

    def submit(self, filenames: Sequence[str], data_root: str, use_deepfeat=False, deepfeat_folder=None):
        n = len(filenames)
        if not n:
            return
        cols = list(zip(*[scene_paths(fn, data_root, use_deepfeat, deepfeat_folder) for fn in filenames]))
        arrs = []
        for col in cols[:3]:
            a = (C.c_char_p * n)(*[p.encode() for p in col])
            arrs.append(a)
        feats = (C.c_char_p * n)(*[p.encode() for p in cols[3]]) if use_deepfeat else None
        self._check(self.lib.gapro_feed_submit(self.handle, n, arrs[0], arrs[1], arrs[2], feats))
        self.names += list(filenames)

    def close(self):
        self._check(self.lib.gapro_feed_close(self.handle))

    def poll(self, min_ready: int, max_scenes: int, timeout_ms: int = -1):
        n, b = C.c_int32(0), C.c_int64(0)
        self._check(self.lib.gapro_feed_poll(self.handle, int(min_ready), int(max_scenes), int(timeout_ms), C.byref(n),
                                             C.byref(b)))
        return int(n.value), int(b.value)

    def upload(self, n: int, d_slab: int, slab_bytes: int, slab_stream: int = 0):
        """The next n loaded scenes into the device slab at address d_slab (allocated on stream `slab_stream`: the copies
        wait for what is queued there); returns (batch id, [TakenScene])."""
        recs = (FeedScene * n)()
        bid = C.c_int64(0)
        self._check(self.lib.gapro_feed_upload(self.handle, n, C.c_void_p(d_slab), int(slab_bytes),
                                               C.c_void_p(slab_stream) if slab_stream else None, recs, C.byref(bid)))
        out = []
        for k in range(n):
            r = recs[k]
            t = TakenScene()
            t.filename = self.names[self.taken + k]
            t.status, t.n_points, t.n_instances, t.feat_dim = int(r.status), int(r.n_points), int(r.n_instances), int(r.feat_dim)
            t.off = (int(r.off_coords), int(r.off_feats), int(r.off_spp), int(r.off_sem), int(r.off_inst))
            t.host_image = r.host_image
            i = t.n_instances
            if t.status == 0 and i > 0:  # copied now: the feed's arrays live until its next upload
                # the dtypes add_instance_info hands to make_job (gen_ps.py:71-77 + the driver's casts)
                t.instance_cls = np.ctypeslib.as_array(r.inst_cls, (i,)).astype(np.int64)
                t.instance_box = np.ctypeslib.as_array(r.inst_box, (i, 6)).astype(np.float32)
                t.instance_box_volume = np.ctypeslib.as_array(r.inst_vol, (i,)).astype(np.float32)
            else:
                t.instance_cls = t.instance_box = t.instance_box_volume = None
            out.append(t)
        self.taken += n
        return int(bid.value), out

    def batch_wait(self, batch_id: int, stream_handle: int):
        self._check(self.lib.gapro_feed_batch_wait(self.handle, int(batch_id), C.c_void_p(stream_handle)))

    def release_batch(self, batch_id: int):
        self._check(self.lib.gapro_feed_release_batch(self.handle, int(batch_id)))

    def export(self, items, ready_event: int = 0):
        """items: [(path, d_sem, d_inst, d_prob, d_mu, d_var, n_points, n_mu)] with raw addresses."""
        n = len(items)
        if not n:
            return
        arr = (FeedOut * n)()
        for k, (path, a, b, c, d, e, npts, nmu) in enumerate(items):
            arr[k] = FeedOut(a, b, c, d, e, int(npts), int(nmu), path.encode())
        self._check(self.lib.gapro_feed_export(self.handle, n, arr, C.c_void_p(ready_event) if ready_event else None))
        self.exported += n

    def export_wait(self, until_done: int = -1, timeout_ms: int = -1):
        d, f = C.c_int64(0), C.c_int64(0)
        self._check(self.lib.gapro_feed_export_wait(self.handle, int(until_done), int(timeout_ms), C.byref(d), C.byref(f)))
        return int(d.value), int(f.value)

    def export_errors(self, n_failed: int):
        out = []
        buf = C.create_string_buffer(1024)
        for i in range(n_failed):
            if self.lib.gapro_feed_export_error(self.handle, i, buf, 1024) == 0:
                out.append(buf.value.decode(errors="replace"))
        return out

    def stats(self):
        out = (C.c_double * 8)()
        self._check(self.lib.gapro_feed_stats(self.handle, out))
        return dict(pin_s=out[0], pin_blocks=int(out[1]), pin_GB=out[2] / 1e9, load_s=out[3], loads=int(out[4]),
                    write_s=out[5], writes=int(out[6]), first_loaded_s=out[7])

    def destroy(self, process_is_exiting: bool = False):
        if self.handle:
            if process_is_exiting:
                self.lib.gapro_feed_detach(self.handle)
            else:
                self.lib.gapro_feed_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass
