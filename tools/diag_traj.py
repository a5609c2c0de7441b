#!/usr/bin/env python3
"""HIP vs oracle trajectories on one small problem: python tools/diag_traj.py m1 m2 t d"""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from gapro_amd.gaussian_process_utils import fit_gp_spp_batch
from gapro_amd.synth import make_gp_problem
from oracle import svgp_oracle as so

m1, m2, t, d = [int(v) for v in sys.argv[1:5]]
feats, b1, b2, it = make_gp_problem(7 + m1, m1, m2, t, d)
X = np.concatenate([feats[b1], feats[b2]]).astype(np.float64)
y = np.r_[-np.ones(m1), np.ones(m2)]
Xt = feats[it].astype(np.float64)
for iters in (1, 2, 3, 4, 6, 8, 12, 20, 30, 50):
    out = fit_gp_spp_batch(feats, [(b1, b2, it)], training_iter=iters)[0]
    a = so.svgp_fit_predict_autograd(X, y, Xt, iters, "f64")
    m = so.svgp_fit_predict_manual(X, y, Xt, iters)
    print("iters %2d  hip-vs-autograd var %.2e mu %.2e | manual-vs-autograd var %.2e mu %.2e | var %s" % (
        iters, np.max(np.abs(out[4] - a[1]) / a[1]), np.max(np.abs(out[3] - a[0])), np.max(np.abs(m[1] - a[1]) / a[1]),
        np.max(np.abs(m[0] - a[0])), a[1][:2]))
