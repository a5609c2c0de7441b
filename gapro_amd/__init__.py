"""gapro_amd -- MI355X-native implementation of GaPro's Gaussian-Process pseudo-label generator.

Drop-in for the reference's gapro/gen_ps_utils.py + gapro/gaussian_process_utils.py entry points
(same names, arguments, return order, dtypes and lengths); the arithmetic runs in hand-written HIP
kernels behind the C ABI of include/gapro_hip.h (libgapro_hip.so).  No CPU fallback.
"""
from .gen_ps_utils import (batch_giou_cross, gen_pseudo_label, gen_pseudo_label_box2mask,  # noqa: F401
                           gen_pseudo_label_gaussian_process, gen_pseudo_label_gaussian_process_batch,
                           getInstanceInfo, getInstanceInfo_device, is_box1_in_box2)
from .gaussian_process_utils import fit_gp_spp, fit_gp_spp_batch  # noqa: F401
from .scannet_planes import get_wall_boxes  # noqa: F401

__version__ = "0.1.0"
