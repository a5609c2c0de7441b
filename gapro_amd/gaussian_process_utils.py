"""Same-name mirror of reference gapro/gaussian_process_utils.py for its only live fitter.

``fit_gp_spp`` keeps the reference signature (gaussian_process_utils.py:382) and return order (:445):
(pred_probs f32[T], pred_probs_new f32[T], pred_labels bool[T], pred_mu f32[T], pred_variance f32[T]).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import FitDesc
from .gen_ps_utils import _pick_device, _pipeline


def fit_gp_spp_batch(feats_spp, problems, training_iter=50, init_mean=None, device=None, keep_debug=False,
                     return_status=False, reproducibility_probe=False, **pipe_kw):
    """Fit many independent GPs in one launch.

    feats_spp  f32[S,D] (torch or numpy); problems = list of (b1_inds, b2_inds, intersect_inds).
    init_mean  optional list of per-problem initial variational means (length m1+m2 each).
    Returns a list of 5-tuples of numpy arrays in the reference's order, plus (if keep_debug) the raw
    result dict as a second value.  A fit that fails (non-finite input, K_ZZ not positive definite after the
    jitter retries of gpytorch's psd_safe_cholesky) raises GaproError, as gpytorch raises there; with
    ``return_status=True`` nothing is raised and the per-fit gapro_status array (0 = ok) comes back as the last
    value: a failed fit does not affect the other fits of the launch.
    The raw result dict carries ``cond``, a per-fit conditioning figure of the last Cholesky factor (a diagnostic), and
    with ``reproducibility_probe=True`` (implies keep_debug; twice the work) ``repro_dv`` / ``repro_dp``: how far each
    fit's sigma^2 (relative) and p (absolute) move when the jitter on K_ZZ is scaled by (1 + 1e-11) -- beyond
    pipeline.REPRO_SOFT (1e-5) a fit's variances are reproducible by no float64 implementation to 1e-4.
    """
    dev = _pick_device(feats_spp, device)
    pipe = _pipeline(dev, training_iter, **pipe_kw)
    f = feats_spp if isinstance(feats_spp, torch.Tensor) else torch.from_numpy(np.asarray(feats_spp))
    f = f.to(device=dev, dtype=torch.float32).contiguous()
    n = len(problems)
    descs = (FitDesc * max(n, 1))()
    idx, init = [], []
    io = oo = 0
    for i, (b1, b2, it) in enumerate(problems):
        b1 = np.asarray(b1.cpu() if isinstance(b1, torch.Tensor) else b1, dtype=np.int32).reshape(-1)
        b2 = np.asarray(b2.cpu() if isinstance(b2, torch.Tensor) else b2, dtype=np.int32).reshape(-1)
        it = np.asarray(it.cpu() if isinstance(it, torch.Tensor) else it, dtype=np.int32).reshape(-1)
        if len(b1) == 0 or len(b2) == 0:
            raise ValueError("fit_gp_spp needs at least one superpoint on each side")
        d = descs[i]
        d.m1, d.m2, d.t, d.b1, d.b2, d.scene = len(b1), len(b2), len(it), 0, 1, i
        d.idx_offset, d.out_offset, d.ws_offset = io, oo, 0
        idx += [b1, b2, it]
        if init_mean is not None:
            im = np.zeros(len(b1) + len(b2) + len(it))
            im[:len(b1) + len(b2)] = np.asarray(init_mean[i], dtype=np.float64)
            init.append(im)
        io += len(b1) + len(b2) + len(it)
        oo += len(it)
    h_idx = np.ascontiguousarray(np.concatenate(idx)) if idx else np.zeros(1, np.int32)
    h_init = np.concatenate(init) if init else None
    keep_debug = keep_debug or reproducibility_probe
    res = pipe.fit_descs(f, descs, n, h_idx, oo, init_mean=h_init, keep_debug=keep_debug,
                         raise_on_failure=not return_status)
    if reproducibility_probe:
        res["repro_dv"], res["repro_dp"] = pipe.reproducibility_probe(f, descs, n, h_idx, oo, res=res, init_mean=h_init)
    outs = []
    for i in range(n):
        a, b = descs[i].out_offset, descs[i].out_offset + descs[i].t
        outs.append((res["probs"][a:b], res["probs_new"][a:b], res["labels"][a:b].astype(bool), res["mu"][a:b],
                     res["var"][a:b]))
    if return_status:
        return (outs, res, res["status"]) if keep_debug else (outs, res["status"])
    return (outs, res) if keep_debug else outs


def fit_gp_spp(coords_float_spp, feats_spp, b1_inds, b2_inds, intersect_inds, training_iter=50, *,
               init_mean=None, device=None, **pipe_kw):
    """Reference gaussian_process_utils.py:382-445.  ``coords_float_spp`` is accepted and unused, as in
    the reference.  Returns torch tensors on the device of ``feats_spp``."""
    dev = _pick_device(feats_spp, device)
    out = fit_gp_spp_batch(feats_spp, [(b1_inds, b2_inds, intersect_inds)], training_iter,
                           init_mean=None if init_mean is None else [init_mean], device=dev, **pipe_kw)[0]
    keep_cpu = isinstance(feats_spp, torch.Tensor) and not feats_spp.is_cuda
    tens = tuple(torch.from_numpy(np.ascontiguousarray(o)) for o in out)
    return tens if keep_cpu else tuple(t.to(dev) for t in tens)
