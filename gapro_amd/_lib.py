"""ctypes binding of libgapro_hip.so (the C ABI declared in include/gapro_hip.h).

The library is the product: there is no Python/NumPy/torch fallback for the hot path.  If the
shared object is missing or cannot be loaded this module raises at import of the first symbol.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgapro_hip.so")

GAPRO_OK = 0
GAPRO_ERR_NOT_FINITE = -4
GAPRO_ERR_CHOLESKY = -5
STATUS_NAMES = {0: "OK", -1: "BAD_ARG", -2: "OOM", -3: "HIP", -4: "NOT_FINITE", -5: "CHOLESKY", -6: "SPP_RANGE",
                -7: "WORKSPACE", -8: "TIMEOUT", -9: "IO", -10: "UNSUPPORTED"}
GAPRO_ERR_TIMEOUT = -8
GAPRO_ERR_IO = -9
GAPRO_ERR_UNSUPPORTED = -10


class GaproError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("libgapro_hip: %s (%d) %s" % (STATUS_NAMES.get(code, "?"), code, msg))
        self.code = code


class SceneHeader(C.Structure):
    _fields_ = [("coord_min", C.c_double * 3), ("coord_max", C.c_double * 3), ("spp_min", C.c_int64),
                ("spp_max", C.c_int64), ("feat_absmax", C.c_float), ("fixed_shift", C.c_int32),
                ("n_spps", C.c_int32), ("status", C.c_int32)]


class ScheduleCounts(C.Structure):
    _fields_ = [("n_events", C.c_int32), ("n_fits", C.c_int32), ("n_event_idx", C.c_int64),
                ("n_fit_idx", C.c_int64), ("n_fit_out", C.c_int64), ("max_m", C.c_int32), ("max_t", C.c_int32)]


class FitDesc(C.Structure):
    _fields_ = [("m1", C.c_int32), ("m2", C.c_int32), ("t", C.c_int32), ("b1", C.c_int32), ("b2", C.c_int32),
                ("scene", C.c_int32), ("slot", C.c_int32), ("reserved", C.c_int32), ("idx_offset", C.c_int64),
                ("out_offset", C.c_int64),
                ("ws_offset", C.c_int64)]


class SceneTask(C.Structure):
    """gapro_scene_task: the device pointers of one scene of a batched partition call."""
    _fields_ = [("n_points", C.c_int64), ("coords", C.c_void_p), ("feats", C.c_void_p), ("spp", C.c_void_p),
                ("spp_inv", C.c_void_p), ("prepare_ws", C.c_void_p), ("spp_range_cap", C.c_int64),
                ("boxes", C.c_void_p), ("n_boxes", C.c_int32), ("n_spps", C.c_int32), ("fixed_shift", C.c_int32),
                ("thresh_spp_occu", C.c_float), ("feat_sum", C.c_void_p), ("occ_count", C.c_void_p),
                ("point_count", C.c_void_p), ("feats_spp", C.c_void_p), ("occ_bits", C.c_void_p),
                ("n_bbs", C.c_void_p), ("sem_spp", C.c_void_p), ("inst_spp", C.c_void_p), ("prob_spp", C.c_void_p),
                ("sem", C.c_void_p), ("inst", C.c_void_p), ("prob", C.c_void_p)]


class PthArray(C.Structure):
    """gapro_pth_array: one NumPy array of a torch.save()d file."""
    _fields_ = [("kind", C.c_int32), ("itemsize", C.c_int32), ("ndim", C.c_int32), ("encoded", C.c_int32),
                ("shape", C.c_int64 * 4), ("nbytes", C.c_int64)]


class InstanceHeader(C.Structure):
    _fields_ = [("instance_num", C.c_int32), ("n_boxes", C.c_int32), ("status", C.c_int32), ("reserved", C.c_int32)]


class EvalHeader(C.Structure):
    _fields_ = [("n_gt", C.c_int32), ("n_ps", C.c_int32), ("status", C.c_int32), ("reserved", C.c_int32)]


class FitOptions(C.Structure):
    _fields_ = [("training_iter", C.c_int32), ("lr", C.c_double), ("jitter", C.c_double),
                ("min_variance", C.c_double), ("eval_stale_chol", C.c_int32), ("reserved", C.c_int32),
                ("psd_retries", C.c_int32), ("precision", C.c_int32), ("psd_jitter", C.c_double)]


# name -> (restype, argtypes); every symbol of include/gapro_hip.h
_P = C.c_void_p
SIGNATURES = {
    "gapro_version": (C.c_int, []),
    "gapro_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "gapro_ctx_destroy": (None, [_P]),
    "gapro_last_error": (C.c_char_p, [_P]),
    "gapro_partition_prepare_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "gapro_partition_prepare": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, _P, _P, C.c_int64, _P, C.c_size_t, _P,
                                          C.POINTER(SceneHeader)]),
    "gapro_partition_prepare_async": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, _P, _P, C.c_int64, _P, C.c_size_t,
                                                _P, _P]),
    "gapro_partition_pool": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                       _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gapro_broadcast_labels": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P, _P, _P, _P, _P]),
    "gapro_partition_prepare_batch": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "gapro_partition_pool_batch": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P]),
    "gapro_broadcast_labels_batch": (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    "gapro_instance_info_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "gapro_instance_info": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P, C.c_int32, C.c_int32, _P, C.c_size_t, _P, _P, _P, _P,
                                      _P, _P]),
    "gapro_eval_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "gapro_eval_miou": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P, _P, C.c_int32, C.c_int32, _P, C.c_size_t, _P, _P, _P,
                                  _P]),
    "gapro_eval_sem_confusion": (C.c_int, [_P, _P, C.c_int64, _P, _P, C.c_int32, _P]),
    "gapro_label_heuristic_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "gapro_label_heuristic": (C.c_int, [_P, _P, C.c_int64, _P, _P, C.c_int32, C.c_int32, _P, _P, _P, C.c_int32, C.c_int32,
                                        C.c_int32, _P, C.c_size_t, _P, _P]),
    "gapro_label_pool_mean": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gapro_weighted_bce_with_logits": (C.c_int, [_P, _P, C.c_int32, C.c_int64, _P, _P, _P, C.c_float, _P, _P, _P]),
    "gapro_kl_gp_loss": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, _P, _P, _P,
                                   _P]),
    "gapro_schedule_build": (C.c_int, [C.c_int32, C.c_int32, _P, _P, _P, C.POINTER(_P)]),
    "gapro_schedule_free": (None, [_P]),
    "gapro_schedule_get_counts": (C.c_int, [_P, C.POINTER(ScheduleCounts)]),
    "gapro_schedule_export_fits": (C.c_int, [_P, C.c_int32, C.c_int64, C.c_int64, C.c_int32, _P, _P]),
    "gapro_schedule_export_events": (C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "gapro_schedule_merge": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "gapro_fit_options_default": (None, [C.POINTER(FitOptions)]),
    "gapro_fit_workspace_doubles": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "gapro_fit_plan_workspace": (C.c_int64, [_P, C.c_int32, C.c_int32]),
    "gapro_svgp_fit_batch": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, C.POINTER(FitOptions), _P,
                                       C.c_size_t, _P, _P, _P, _P, _P, _P, _P]),
    "gapro_svgp_fit_batch_ex": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, C.POINTER(FitOptions), _P,
                                          C.c_size_t, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gapro_fit_workspace_layout": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _P]),
    "gapro_fit_route": (C.c_int, [C.c_int32, C.c_int32]),
    "gapro_fit_padded_m": (C.c_int, [C.c_int32, C.c_int32]),
    "gapro_fit_timing_create": (C.c_int, [_P, C.POINTER(C.c_void_p)]),
    "gapro_fit_timing_destroy": (None, [_P]),
    "gapro_fit_timing_arm": (C.c_int, [_P, _P]),
    "gapro_fit_timing_read": (C.c_int, [_P, _P, C.POINTER(C.c_float)]),
    "gapro_fit_timing_read_wave": (C.c_int, [_P, _P, C.POINTER(C.c_float)]),
    "gapro_fit_timing_offsets": (C.c_int, [_P, _P, _P, C.POINTER(C.c_float)]),
    "gapro_fit_timing_cluster_info": (C.c_int, [_P, _P, C.POINTER(C.c_int32)]),
    "gapro_pth_open": (C.c_int, [C.c_char_p, C.POINTER(_P)]),
    "gapro_pth_count": (C.c_int, [_P]),
    "gapro_pth_is_sequence": (C.c_int, [_P]),
    "gapro_pth_info": (C.c_int, [_P, C.c_int32, C.POINTER(PthArray)]),
    "gapro_pth_read": (C.c_int, [_P, C.c_int32, _P, C.c_int64]),
    "gapro_pth_close": (None, [_P]),
    "gapro_pth_write": (C.c_int, [C.c_char_p, C.c_int32, C.POINTER(PthArray), C.POINTER(_P), C.c_int32]),
    "gapro_pth_last_error": (C.c_char_p, []),
    "gapro_pth_decoder": (C.c_char_p, []),
    "gapro_pth_encoder": (C.c_char_p, []),
    "gapro_pth_crc": (C.c_char_p, []),
    "gapro_pth_crc32": (C.c_uint32, [_P, C.c_int64]),
    "gapro_pth_encode_latin1": (C.c_int64, [_P, C.c_int64, _P, C.c_int64]),
    # batch feeder of the gen_ps driver (gapro_amd/feeder.py holds the structs)
    "gapro_dev_alloc": (C.c_int, [_P, C.c_size_t, _P, C.POINTER(C.c_void_p)]),
    "gapro_dev_free": (C.c_int, [_P, _P]),
    "gapro_dev_trim": (C.c_int, [_P]),
    "gapro_dev_stats": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                  C.POINTER(C.c_int64)]),
    "gapro_host_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(C.c_void_p)]),
    "gapro_host_free": (C.c_int, [_P, _P]),
    "gapro_stream_create": (C.c_int, [_P, C.POINTER(C.c_void_p)]),
    "gapro_stream_destroy": (C.c_int, [_P, _P]),
    "gapro_stream_sync": (C.c_int, [_P, _P]),
    "gapro_device_sync": (C.c_int, [_P]),
    "gapro_event_create": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_void_p)]),
    "gapro_event_destroy": (C.c_int, [_P, _P]),
    "gapro_event_record": (C.c_int, [_P, _P, _P]),
    "gapro_stream_wait_event": (C.c_int, [_P, _P, _P]),
    "gapro_event_sync": (C.c_int, [_P, _P]),
    "gapro_event_query": (C.c_int, [_P, _P]),
    "gapro_event_elapsed_ms": (C.c_int, [_P, _P, _P, C.POINTER(C.c_float)]),
    "gapro_memcpy_async": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_int32, _P]),
    "gapro_memset_async": (C.c_int, [_P, _P, C.c_int32, C.c_size_t, _P]),
    "gapro_feed_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_void_p)]),
    "gapro_feed_destroy": (None, [_P]),
    "gapro_feed_detach": (None, [_P]),
    "gapro_feed_last_error": (C.c_char_p, [_P]),
    "gapro_feed_submit": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P]),
    "gapro_feed_close": (C.c_int, [_P]),
    "gapro_feed_poll": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "gapro_feed_upload": (C.c_int, [_P, C.c_int32, _P, C.c_int64, _P, _P, C.POINTER(C.c_int64)]),
    "gapro_feed_batch_wait": (C.c_int, [_P, C.c_int64, _P]),
    "gapro_feed_release_batch": (C.c_int, [_P, C.c_int64]),
    "gapro_feed_export": (C.c_int, [_P, C.c_int32, _P, _P]),
    "gapro_feed_export_wait": (C.c_int, [_P, C.c_int64, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gapro_feed_export_error": (C.c_int, [_P, C.c_int32, _P, C.c_int32]),
    "gapro_feed_stats": (C.c_int, [_P, _P]),
    "gapro_scene_default_feats": (C.c_int, [_P, _P, C.c_int64, _P]),
    "gapro_scene_instance_boxes": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P, _P,
                                             C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
}

# libgapro_hip_debug.so (include/gapro_hip_debug.h): measurement / self-test entry points, loaded on request only
DEBUG_LIB_PATH = os.path.join(_HERE, "libgapro_hip_debug.so")
DEBUG_SIGNATURES = {
    "gapro_debug_mfma_tn": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32]),
    "gapro_debug_stream": (C.c_int, [_P, _P, C.c_int64, _P, _P, C.c_int32]),
    "gapro_debug_fit_math": (C.c_int, [_P, _P, C.c_int64, _P, _P, C.c_int32]),
    "gapro_debug_mfma_peak": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.POINTER(C.c_double)]),
    "gapro_debug_mfma_clock": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.POINTER(C.c_double),
                                         C.POINTER(C.c_double)]),
    "gapro_debug_product_bench": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P,
                                            C.POINTER(C.c_float)]),
    "gapro_debug_wgloop": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.POINTER(C.c_double)]),
}

_lib: Optional[C.CDLL] = None
_dbg: Optional[C.CDLL] = None


def source_build_id() -> str:
    """Identity of the DEVICE KERNELS this tree holds (sha256 over the kernel sources and the headers they include, 16
    hex digits): bench.py prints it and the committed PMC passes carry it, so that a traffic figure is only ever
    attached to a bench line of the SAME kernels.  Host-only sources (file I/O, feeder, arena,
    scheduler) are left out: they move no byte of a kernel's traffic."""
    import glob
    import hashlib

    h = hashlib.sha256()
    files = []
    for pat in ("csrc/svgp_fit*.hip", "csrc/partition.hip", "csrc/labels.hip", "csrc/consumer.hip", "csrc/*.h"):
        files += glob.glob(os.path.join(_HERE, pat))
    for fn in sorted(files, key=os.path.basename):
        h.update(os.path.basename(fn).encode())
        with open(fn, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load_debug() -> C.CDLL:
    """Load libgapro_hip_debug.so (tools, bench.py's peak_measured, the MFMA lane-map test); the product never does."""
    global _dbg
    if _dbg is not None:
        return _dbg
    if not os.path.exists(DEBUG_LIB_PATH):
        raise ImportError("libgapro_hip_debug.so not built: run gapro_amd/csrc/build.sh; expected at " + DEBUG_LIB_PATH)
    lib = C.CDLL(DEBUG_LIB_PATH)
    for name, (res, args) in DEBUG_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _dbg = lib
    return lib


_TORCH_FIRST = [True]  # was torch already in the process when the library was loaded (or never imported)?


def load() -> C.CDLL:
    """Load libgapro_hip.so; raises (never falls back) if it is absent or lacks a symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libgapro_hip.so not built: run gapro_amd/csrc/build.sh (or __graft_entry__.build()); "
                          "expected at " + LIB_PATH)
    _TORCH_FIRST[0] = "torch" in sys.modules
    lib = C.CDLL(LIB_PATH)
    default = os.path.join(_HERE, "libgapro_hip.so")
    host_only = None  # A/B tools point LIB_PATH at a variant build that may predate a host-side helper
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)  # AttributeError if the symbol is missing
        except AttributeError:
            if os.path.abspath(LIB_PATH) == os.path.abspath(default) or not name.startswith(("gapro_pth_", "gapro_scene_", "gapro_feed_")):
                raise
            if host_only is None:
                host_only = C.CDLL(default)
            fn = getattr(host_only, name)
            setattr(lib, name, fn)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class Context:
    """One gapro_ctx per (process, device)."""

    _cache = {}

    def __init__(self, device: int = 0):
        self.lib = load()
        h = _P()
        rc = self.lib.gapro_ctx_create(int(device), C.byref(h))
        if rc != GAPRO_OK:
            raise GaproError(rc, "gapro_ctx_create(device=%d): no usable HIP device%s" % (device, (
                " (torch was imported AFTER libgapro_hip.so was loaded: the process now holds two HIP runtimes, torch's "
                "bundled one and /opt/rocm's -- import torch first, or use the torch-free backend)"
                if ("torch" in sys.modules and not _TORCH_FIRST[0]) else "")))
        self.handle = h
        self.device = int(device)

    @property
    def dbg(self) -> C.CDLL:
        """The debug library's entry points (they take this context's handle)."""
        return load_debug()

    @classmethod
    def get(cls, device: int = 0) -> "Context":
        ctx = cls._cache.get(device)
        if ctx is None:
            ctx = cls._cache[device] = cls(device)
        return ctx

    def check(self, rc: int):
        if rc != GAPRO_OK:
            raise GaproError(rc, (self.lib.gapro_last_error(self.handle) or b"").decode())

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.gapro_ctx_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def default_fit_options(training_iter: int = 50) -> FitOptions:
    opt = FitOptions()
    load().gapro_fit_options_default(C.byref(opt))
    opt.training_iter = int(training_iter)
    return opt
