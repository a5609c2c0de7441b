#!/usr/bin/env python3
"""VERDICT r05 item 2b, at the level of the arithmetic (CPU, float64 NumPy): what would the two candidate cures for the
kernels' explicit triangular inverse buy on the two ill-conditioned fits of the S3DIS-shaped test scene?

The kernels form LI = L^-1 (16 x 16 diagonal-block inverses + block forward substitution) and MULTIPLY -- A = LI K_ZX,
G_KX = LI^T G_A, G_Kzz = LI^T Pm LI -- where the oracle (and gpytorch) SOLVE triangular systems.  This tool runs the
NumPy oracle (oracle/svgp_oracle.py, the hand-derived backward the kernels implement) with its two solve hooks replaced by

    solve      scipy.linalg.solve_triangular (the oracle as it is)
    inverse    LI = L^-1 by the kernels' recipe (diagonal-block inverses, block forward substitution), then LI @ B
    newton     the same LI followed by one Newton step  LI <- LI (2 I - L LI)  (two more triangular products per step)
    blocked    no LI at all: blocked forward / backward substitution that only uses the 16 x 16 diagonal-block inverses
               (same M^3 as the multiply, and no inverse phase)

and prints each variant's sigma^2 / p deviation from the float64 AUTOGRAD oracle on fits 5 (M = 58) and 46 (M = 144),
next to a well-conditioned fit.  Arithmetic only: sums are NumPy's, not MFMA order.

    python tools/inverse_cures.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def block_inverse(L, nb=16):
    """L^-1 the kernels' way: Dinv_k = inverse of the k-th diagonal block (forward substitution), then
    LI_ik = -Dinv_i sum_{j=k}^{i-1} L_ij LI_jk."""
    from scipy.linalg import solve_triangular

    M = L.shape[0]
    blocks = [(a, min(a + nb, M)) for a in range(0, M, nb)]
    LI = np.zeros_like(L)
    dinv = [solve_triangular(L[a:b, a:b], np.eye(b - a), lower=True) for a, b in blocks]
    for k, (ka, kb) in enumerate(blocks):
        LI[ka:kb, ka:kb] = dinv[k]
        for i in range(k + 1, len(blocks)):
            ia, ib = blocks[i]
            acc = L[ia:ib, ka:ia] @ LI[ka:ia, ka:kb]
            LI[ia:ib, ka:kb] = -dinv[i] @ acc
    return LI


def blocked_solve(L, B, nb=16, trans=False):
    """L X = B (or L^T X = B) by block substitution that multiplies with the diagonal-block inverses only."""
    from scipy.linalg import solve_triangular

    M = L.shape[0]
    blocks = [(a, min(a + nb, M)) for a in range(0, M, nb)]
    dinv = [solve_triangular(L[a:b, a:b], np.eye(b - a), lower=True) for a, b in blocks]
    X = np.array(B, dtype=np.float64, copy=True)
    if not trans:
        for i, (a, b) in enumerate(blocks):
            X[a:b] = dinv[i] @ (X[a:b] - L[a:b, :a] @ X[:a])
    else:
        for i in reversed(range(len(blocks))):
            a, b = blocks[i]
            X[a:b] = dinv[i].T @ (X[a:b] - L[b:, a:b].T @ X[b:])
    return X


def main():
    import torch
    torch.set_num_threads(8)
    from gapro_amd.gen_ps_utils import getInstanceInfo
    from gapro_amd.synth import make_scene
    from oracle import gen_ps_oracle as O
    from oracle import svgp_oracle as so

    sc = make_scene(seed=7, n_points=1_000_000, n_objects=40, with_walls_json=False, obj_patch=60, plane_patch=400)
    xyz = sc.aligned_xyz()
    _, cls, box, vol, _ = getInstanceInfo(xyz, sc.inst, sc.sem)
    kw = dict(coords_float=xyz, mask_feats=sc.default_feats().astype(np.float32), spp=sc.spp,
              instance_cls=cls.astype(np.int64), instance_box=box.astype(np.float32),
              instance_box_volume=vol.astype(np.float32), wall_box=[], wall_box_volume=[], instance_classes=13,
              ground_h=0.1, training_iter=50, thresh_spp_occu=0.999)

    def stub(f, b1, b2, it):  # the schedule does not depend on the GP outputs: any outputs will do
        n = len(it)
        return (np.full(n, 0.5, np.float32), np.full(n, 0.5, np.float32), np.zeros(n, bool), np.zeros(n, np.float32),
                np.ones(n, np.float32))

    _, dbg = O.gen_pseudo_label_gaussian_process(**kw, fit_fn=stub, return_debug=True)
    fits = [e for e in dbg["events"] if e.kind == "fit"]
    f = dbg["part"].feats_spp
    hooks = (so._tril_solve, so._triu_solve_T)
    variants = {
        "solve": hooks,
        "inverse": (lambda L, B: block_inverse(L) @ B, lambda L, B: block_inverse(L).T @ B),
        "newton": None,
        "blocked": (lambda L, B: blocked_solve(L, B), lambda L, B: blocked_solve(L, B, trans=True)),
    }

    def newton_li(L):
        LI = block_inverse(L)
        return np.tril(LI @ (2.0 * np.eye(L.shape[0]) - L @ LI))

    variants["newton"] = (lambda L, B: newton_li(L) @ B, lambda L, B: newton_li(L).T @ B)
    print("deviation from the float64 autograd oracle, sigma^2 relative | p absolute")
    for k in (5, 46, 10):
        e = fits[k]
        X = np.concatenate([f[e.b1_inds], f[e.b2_inds]]).astype(np.float64)
        y = np.r_[-np.ones(len(e.b1_inds)), np.ones(len(e.b2_inds))]
        Xt = f[e.intersect_inds].astype(np.float64)
        a = so.svgp_fit_predict_autograd(X, y, Xt, 50, "f64")
        row = []
        for name, hk in variants.items():
            so._tril_solve, so._triu_solve_T = hk
            try:
                m = so.svgp_fit_predict_manual(X, y, Xt, 50)
            finally:
                so._tril_solve, so._triu_solve_T = hooks
            row.append("%s %.1e | %.1e" % (name, np.max(np.abs(a[1] - m[1]) / a[1]), np.max(np.abs(a[2] - m[2]))))
        print("fit %2d (M = %3d): " % (k, len(X)) + "   ".join(row), flush=True)


if __name__ == "__main__":
    main()
