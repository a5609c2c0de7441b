#!/usr/bin/env python3
"""Stage-by-stage comparison of the HIP fit kernel with the NumPy oracle (run on the GPU box).

python tools/diag_fit.py [m1 m2 t d]
"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch  # noqa: E402

from gapro_amd import _lib  # noqa: E402
from gapro_amd.gaussian_process_utils import fit_gp_spp_batch  # noqa: E402
from gapro_amd.synth import make_gp_problem  # noqa: E402
from oracle import svgp_oracle as so  # noqa: E402

MATS = ["LS", "LST", "MLS", "VLS", "GLS", "L", "LT", "LI", "U", "KX", "A", "AT", "BM", "BMT", "GA", "GKX", "GKXT"]
VECS = ["Y", "M", "MM", "VM", "GM", "MU", "VAR", "GMU", "GV"]


def grab(res, m, t, d):
    lay = (C.c_int64 * 8)()
    _lib.load().gapro_fit_workspace_layout(m, t, d, C.cast(lay, C.c_void_p))
    Mp, mat, vec, xz, xt, dinv, scal, total = [int(v) for v in lay]
    ws = res["workspace"].cpu().numpy()
    out = {"Mp": Mp}
    for i, n in enumerate(MATS):
        out[n] = ws[mat + i * Mp * Mp: mat + (i + 1) * Mp * Mp].reshape(Mp, Mp)
    for i, n in enumerate(VECS):
        out[n] = ws[vec + i * Mp: vec + (i + 1) * Mp]
    for i, n in enumerate(["X", "Z", "mZ", "vZ", "gZ"]):
        out[n] = ws[xz + i * Mp * d: xz + (i + 1) * Mp * d].reshape(Mp, d)
    out["scal"] = ws[scal: scal + 16]
    return out


def err(name, a, b):
    a, b = np.asarray(a), np.asarray(b)
    den = max(np.abs(b).max(), 1e-300)
    print("  %-10s max|diff| %.3e  (rel to max %.3e)  ref max %.3e" % (name, np.abs(a - b).max(), np.abs(a - b).max() / den, den))


def main():
    m1, m2, t, d = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (20, 30, 10, 6)
    M = m1 + m2
    feats, b1, b2, it = make_gp_problem(3, m1, m2, t, d)
    X = np.concatenate([feats[b1], feats[b2]]).astype(np.float64)
    y = np.r_[-np.ones(m1), np.ones(m2)]
    Xt = feats[it].astype(np.float64)
    print("== problem M=%d T=%d D=%d" % (M, t, d))

    # ---- iter 0: Cholesky, inverse, forward products, prediction with initial parameters
    outs, res = fit_gp_spp_batch(feats, [(b1, b2, it)], training_iter=0, keep_debug=True)
    w = grab(res, M, t, d)
    s = ell = np.log(2.0)
    d2 = ((X[:, None] - X[None]) ** 2).sum(-1)
    Kzz = s * np.exp(-0.5 * d2 / ell**2) + 1e-4 * np.eye(M)
    L = np.linalg.cholesky(Kzz)
    LI = np.linalg.inv(L)
    print("-- iter 0")
    err("L", w["L"][:M, :M], L)
    err("LT", w["LT"][:M, :M], L.T)
    err("LI", w["LI"][:M, :M], LI)
    err("U", w["U"][:M, :M], LI.T)
    mu, var, p = so.svgp_predict(Xt, X, np.zeros(M), np.eye(M), 0.0, 0.0, 0.0)
    err("mu0", outs[0][3], mu)
    err("var0", outs[0][4], var)
    err("p0", outs[0][0], p)
    d2t = ((X[:, None] - Xt[None]) ** 2).sum(-1)
    At = LI @ (s * np.exp(-0.5 * d2t / ell**2))
    err("A(test)", w["A"][:M, :t], At)
    err("AT(test)", w["AT"][:t, :M], At.T)
    err("BM(test)", w["BM"][:M, :t], At)

    # ---- iter 1 with the stale factor: gradients of the first step
    outs, res = fit_gp_spp_batch(feats, [(b1, b2, it)], training_iter=1, keep_debug=True, eval_stale_chol=True)
    w = grab(res, M, t, d)
    loss, G = so.svgp_loss_and_grads(X, y, X.copy(), np.zeros(M), np.eye(M), 0.0, 0.0, 0.0)
    print("-- iter 1 (gradients at the initial point); loss hip %.12f oracle %.12f" % (res["loss"][0], loss))
    err("G_LS", w["MLS"][:M, :M] / 0.1, G["LS"])  # Adam first moment after one step = 0.1 g
    err("G_Z", w["gZ"][:M], G["Z"])
    err("G_m", w["GM"][:M], G["m"])
    sc = w["scal"]
    print("  adam m (c, rho_s, rho_l) / 0.1:", sc[3] / 0.1, sc[4] / 0.1, sc[5] / 0.1, " oracle:", G["c"], G["rho_s"], G["rho_l"])
    print("  params after 1 step (c, rho_s, rho_l):", sc[0], sc[1], sc[2])

    # ---- full runs
    for iters in (2, 5, 50):
        outs = fit_gp_spp_batch(feats, [(b1, b2, it)], training_iter=iters)
        mu, var, p = so.svgp_fit_predict_manual(X, y, Xt, iters)
        print("-- iter %d" % iters)
        err("mu", outs[0][3], mu)
        err("var", outs[0][4], var)
        err("p", outs[0][0], p)
        print("  rel var err %.3e, rel mu err %.3e" % (np.max(np.abs(outs[0][4] - var) / var), np.max(np.abs(outs[0][3] - mu) / np.abs(mu))))


if __name__ == "__main__":
    main()
