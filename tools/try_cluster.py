#!/usr/bin/env python3
"""Quick check of the cluster fit kernel against the float64 oracle (diagnostic)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gapro_amd.gaussian_process_utils import fit_gp_spp_batch
from gapro_amd.synth import make_gp_problem
from gapro_amd import _lib
from oracle import svgp_oracle as so
import torch

def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    sizes = [(200, 215, 40), (260, 270, 90), (340, 360, 50)] if len(sys.argv) < 3 else [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]]
    for (m1, m2, t) in sizes:
        feats, b1, b2, it = make_gp_problem(40 + m1, m1, m2, t, 6)
        print("M", m1 + m2, "route", _lib.load().gapro_fit_route(m1 + m2, 6), flush=True)
        t0 = time.time()
        out = fit_gp_spp_batch(feats, [(b1, b2, it)], training_iter=iters)[0]
        torch.cuda.synchronize()
        t1 = time.time()
        out = fit_gp_spp_batch(feats, [(b1, b2, it)], training_iter=iters)[0]
        t2 = time.time()
        X = np.concatenate([feats[b1], feats[b2]]).astype(np.float64)
        y = np.r_[-np.ones(m1), np.ones(m2)]
        mu, var, p = so.svgp_fit_predict_autograd(X, y, feats[it].astype(np.float64), iters, "f64")
        print("  gpu %.3f s (first %.3f)  dv %.2e dmu %.2e dp %.2e" % (t2 - t1, t1 - t0, np.max(np.abs(out[4] - var) / var),
              np.max(np.abs(out[3] - mu)), np.max(np.abs(out[0] - p))), flush=True)

if __name__ == "__main__":
    main()
