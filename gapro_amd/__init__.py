"""gapro_amd -- MI355X-native implementation of GaPro's Gaussian-Process pseudo-label generator.

Drop-in for the reference's gapro/gen_ps_utils.py + gapro/gaussian_process_utils.py entry points
(same names, arguments, return order, dtypes and lengths); the arithmetic runs in hand-written HIP
kernels behind the C ABI of include/gapro_hip.h (libgapro_hip.so).  No CPU fallback.

The names below are resolved on first use (PEP 562): `python -m gapro_amd.gen_ps --devices 0,..,7` runs this file in
its parent process, which starts the workers and needs none of them -- importing torch there was 1.2 s of every job.
"""
import importlib

__version__ = "0.1.0"

_EXPORTS = {
    "gen_ps_utils": ("batch_giou_cross", "gen_pseudo_label", "gen_pseudo_label_box2mask",
                     "gen_pseudo_label_gaussian_process", "gen_pseudo_label_gaussian_process_batch", "getInstanceInfo",
                     "getInstanceInfo_device", "is_box1_in_box2"),
    "gaussian_process_utils": ("fit_gp_spp", "fit_gp_spp_batch"),
    "scannet_planes": ("get_wall_boxes",),
}
__all__ = [n for names in _EXPORTS.values() for n in names]


def __getattr__(name):
    for mod, names in _EXPORTS.items():
        if name in names:
            value = getattr(importlib.import_module("." + mod, __name__), name)
            globals()[name] = value
            return value
    if name in _EXPORTS or name in ("gen_ps", "pipeline", "synth", "_lib", "pth_io", "feeder", "dist_utils",
                                    "consumer_ops", "eval_ps_labels"):
        return importlib.import_module("." + name, __name__)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))


def __dir__():
    return sorted(list(globals()) + __all__)
