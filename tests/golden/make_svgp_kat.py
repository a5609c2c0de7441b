#!/usr/bin/env python3
"""Freeze the float64 GP oracle: known-answer vectors tests/golden/svgp_kat_*.npz.

    python tests/golden/make_svgp_kat.py

The GP arithmetic of the reference lives in gpytorch, which cannot be run here (parity unpinned), so
the target the HIP kernels are held to is oracle/svgp_oracle.py.  These files pin THAT target: inputs (pooled
features and the three index vectors of fit_gp_spp, reference gaussian_process_utils.py:382) and the oracle's
50-step results (mu, sigma^2, p, the ELBO loss of every step, the trained scalars) as computed by the torch-autograd
float64 implementation at the commit that wrote them.  A later edit of the oracle that moves any of them fails
tests/test_svgp_kat.py instead of silently moving the target; the GPU tests compare the HIP kernels with the same
files.  Data only: no reference source is involved.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from gapro_amd.synth import make_gp_problem  # noqa: E402
from oracle import svgp_oracle as so  # noqa: E402

# name -> (seed, m1, m2, t, d, std): every kernel route of both feature widths (gapro_fit_route)
CASES = {
    "m7_d6": (901, 3, 4, 5, 6, 1.0),
    "m50_d6": (902, 20, 30, 10, 6, 1.0),
    "m100_d6": (903, 40, 60, 33, 6, 1.0),
    "m150_d6": (904, 70, 80, 100, 6, 1.0),
    "m256_d6": (905, 120, 136, 40, 6, 1.0),
    "m370_d6": (909, 185, 185, 20, 6, 1.0),   # staged kernel, 64 x 64 wave tiles (M_p = 384)
    "m530_d6": (910, 265, 265, 15, 6, 1.0),   # cluster kernel over 4 workgroups (M_p = 544)
    "m22_d32": (906, 10, 12, 75, 32, 0.3),
    "m90_d32": (907, 40, 50, 33, 32, 0.3),
    "m190_d32": (908, 90, 100, 20, 32, 0.3),
    "m44_d32": (911, 20, 24, 30, 32, 0.3),    # round 6: M_p = 48, the small-fit strip kernel at D = 32 (m22_d32 now runs
                                              # on the wave-per-fit kernel, which takes M_p <= 32 at this width)
}


def main():
    import torch

    torch.set_num_threads(1)  # one summation order
    only = set(sys.argv[1:])  # names: write only these (the others stay as committed)
    for name, (seed, m1, m2, t, d, std) in CASES.items():
        if only and name not in only:
            continue
        feats, b1, b2, it = make_gp_problem(seed, m1, m2, t, d, std=std)
        X = np.concatenate([feats[b1], feats[b2]]).astype(np.float64)
        y = np.r_[-np.ones(m1), np.ones(m2)]
        (mu, var, p), st = so.svgp_fit_predict_autograd(X, y, feats[it].astype(np.float64), 50, "f64",
                                                        return_trace=True)
        (mu_s, var_s, p_s) = so.svgp_fit_predict_autograd(X, y, feats[it].astype(np.float64), 50, "f64",
                                                          eval_chol="stale")
        np.savez_compressed(os.path.join(HERE, "svgp_kat_%s.npz" % name), feats=feats, b1=b1, b2=b2, it=it,
                            mu=mu, var=var, p=p, loss=np.asarray(st["loss"]), c=st["c"], rho_s=st["rho_s"],
                            rho_l=st["rho_l"], m=st["m"], mu_stale=mu_s, var_stale=var_s)
        print(name, "loss %.9f -> %.9f" % (st["loss"][0], st["loss"][-1]), "var range %.3e..%.3e" % (var.min(), var.max()))


if __name__ == "__main__":
    main()
