#!/usr/bin/env python3
"""Mixed-precision mode vs the oracle's restatements, step by step (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gapro_amd.gaussian_process_utils import fit_gp_spp_batch
from gapro_amd.synth import make_gp_problem
from oracle import svgp_oracle as so

m1, m2, t = 32, 32, 40
feats, b1, b2, it = make_gp_problem(4100 + m1, m1, m2, t, 6)
X = np.concatenate([feats[b1], feats[b2]]).astype(np.float64)
y = np.r_[-np.ones(m1), np.ones(m2)]
Xt = feats[it].astype(np.float64)
for iters in (0, 1, 2, 3, 5, 10, 50):
    k64 = fit_gp_spp_batch(feats, [(b1, b2, it)], training_iter=iters, precision="f64", cluster_all=True)[0]
    kmx = fit_gp_spp_batch(feats, [(b1, b2, it)], training_iter=iters, precision="mixed", cluster_all=True)[0]
    o64 = so.svgp_fit_predict_autograd(X, y, Xt, iters, "f64")
    omx = so.svgp_fit_predict_autograd(X, y, Xt, iters, "mixed")
    rel = lambda a, b: float(np.max(np.abs(np.asarray(a, np.float64) - b) / np.abs(b)))
    print("iters %2d  var: k64-o64 %.1e  kmx-o64 %.1e  omx-o64 %.1e  kmx-omx %.1e | mu(abs): kmx-o64 %.1e omx-o64 %.1e kmx-omx %.1e"
          % (iters, rel(k64[4], o64[1]), rel(kmx[4], o64[1]), rel(omx[1], o64[1]), rel(kmx[4], omx[1]),
             np.max(np.abs(kmx[3] - o64[0])), np.max(np.abs(omx[0] - o64[0])), np.max(np.abs(kmx[3] - omx[0]))), flush=True)
