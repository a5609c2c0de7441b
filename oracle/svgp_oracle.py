"""CPU restatement of the GP that reference ``fit_gp_spp`` trains.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The arithmetic lives in gpytorch (imported at reference
gapro/gaussian_process_utils.py:1,5-7), which is neither vendored, pinned, listed in any
requirements file, nor installed here, and the reference has no test or golden vector for
``fit_gp_spp``.  This module restates gpytorch 1.x's published algorithm for exactly the
objects the reference constructs:

  GPClassificationModel (gaussian_process_utils.py:11-25)
    CholeskyVariationalDistribution(M)        m = 0 (+1e-3*randn on first call), L_S = I
    VariationalStrategy(train_x)              whitened, learn_inducing_locations=True, Z0 = train_x
    ConstantMean()                            c = 0
    ScaleKernel(RBFKernel())                  raw params 0 -> softplus -> s = l = ln 2
  BernoulliLikelihood                         E_q[log Phi(y f)], 20-point Gauss-Hermite
  VariationalELBO(likelihood, model, M)       loss = -(sum_i E_i / M - KL / M)
  Adam(lr=0.1), 50 steps                      gaussian_process_utils.py:410-423
  eval: mu, sigma^2, Phi(mu / sqrt(1 + sigma^2)) gaussian_process_utils.py:426-438

Two implementations that must agree:

* ``svgp_fit_predict_autograd`` -- torch autograd (the gradient is right by construction);
  ``dtype="mixed"`` mirrors gpytorch's precision split (float32 everywhere, float64 for the
  Cholesky factor and the triangular solve), ``dtype="f64"`` is the all-float64 ground truth.
* ``svgp_fit_predict_manual``   -- NumPy float64 with the hand-derived backward
   that the HIP kernel implements step for step.

Version-sensitive knobs are keyword arguments with the survey's defaults.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np

JITTER = 1e-4  # gpytorch settings.variational_cholesky_jitter (float32 default, >= 1.4)
MIN_VARIANCE = 1e-6  # gpytorch settings.min_variance (float32 default)
NUM_GH = 20  # gpytorch settings.num_gauss_hermite_locs
ADAM_LR, ADAM_B1, ADAM_B2, ADAM_EPS = 0.1, 0.9, 0.999, 1e-8  # torch.optim.Adam defaults + lr of :410

_GH_T, _GH_W = np.polynomial.hermite.hermgauss(NUM_GH)


def _softplus(x):
    return np.log1p(np.exp(-abs(x))) + max(x, 0.0)


def _sigmoid(x):
    return 1.0 / (1.0 + math.exp(-x))


# ------------------------------------------------------------------------------------------
# torch autograd restatement
# ------------------------------------------------------------------------------------------
PSD_JITTER, PSD_MAX_TRIES = 1e-8, 3  # gpytorch settings.cholesky_jitter (float64), settings.cholesky_max_tries


def psd_safe_cholesky(A, jitter=PSD_JITTER, max_tries=PSD_MAX_TRIES):
    """gpytorch.utils.cholesky.psd_safe_cholesky restated for one float64 matrix (torch): plain Cholesky first; on
    failure the diagonal gets jitter * 10^i added (cumulatively replacing the previous amount), i < max_tries, then
    NotPSDError (here RuntimeError).  Differentiable where it succeeds (the jitter is a constant)."""
    import torch

    L, info = torch.linalg.cholesky_ex(A)
    if int(info) == 0:
        return L
    if torch.isnan(A).any():
        raise RuntimeError("NanError: cholesky of a matrix with NaNs")
    for i in range(max_tries):
        L, info = torch.linalg.cholesky_ex(A + (jitter * 10.0 ** i) * torch.eye(A.shape[-1], dtype=A.dtype))
        if int(info) == 0:
            return L
    raise RuntimeError("NotPSDError: not positive definite after adding jitter of %g" % (jitter * 10.0 ** (max_tries - 1)))


def svgp_fit_predict_autograd(train_x, train_y, test_x, training_iter=50, dtype="f64", init_mean=None,
                              jitter=JITTER, eval_chol="fresh", lr=ADAM_LR, return_trace=False):
    """dtype: "f64" all float64 (ground truth); "mixed" the reference's split (float32 everywhere, float64 for the
    Cholesky factor and the triangular solve); tolerance-study modes of BASELINE configs[4]: "f32" float32 everywhere
    including the factorisation, "bf16in" = "mixed" with the input features rounded to bfloat16 first."""
    import torch

    if dtype == "bf16in":
        train_x = torch.as_tensor(np.asarray(train_x), dtype=torch.float32).bfloat16().float().numpy()
        test_x = torch.as_tensor(np.asarray(test_x), dtype=torch.float32).bfloat16().float().numpy()
    chol_dt = torch.float32 if dtype == "f32" else torch.float64
    T = torch.float64 if dtype == "f64" else torch.float32
    X = torch.as_tensor(np.asarray(train_x), dtype=T)
    y = torch.as_tensor(np.asarray(train_y), dtype=T)
    Xt = torch.as_tensor(np.asarray(test_x), dtype=T)
    M, D = X.shape
    Z = X.clone().requires_grad_(True)  # VariationalStrategy(self, train_x, ...) :14
    m = torch.zeros(M, dtype=T) if init_mean is None else torch.as_tensor(np.asarray(init_mean), dtype=T).clone()
    m.requires_grad_(True)
    LS = torch.eye(M, dtype=T, requires_grad=True)
    c = torch.zeros((), dtype=T, requires_grad=True)
    rho_s = torch.zeros((), dtype=T, requires_grad=True)
    rho_l = torch.zeros((), dtype=T, requires_grad=True)
    params = [Z, m, LS, c, rho_s, rho_l]
    opt = torch.optim.Adam(params, lr=lr)
    t_k = torch.as_tensor(_GH_T, dtype=T)
    w_k = torch.as_tensor(_GH_W, dtype=T)

    def sqdist(a, b):
        return ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)

    def chol_factor():
        ell = torch.nn.functional.softplus(rho_l)
        s = torch.nn.functional.softplus(rho_s)
        Kzz = s * torch.exp(-0.5 * sqdist(Z / ell, Z / ell)) + jitter * torch.eye(M, dtype=T)
        return psd_safe_cholesky(Kzz.to(chol_dt), jitter=PSD_JITTER if chol_dt == torch.float64 else 1e-6)  # K.double()

    def q_f(x, L):
        ell = torch.nn.functional.softplus(rho_l)
        s = torch.nn.functional.softplus(rho_s)
        Kzx = s * torch.exp(-0.5 * sqdist(Z / ell, x / ell))
        A = torch.linalg.solve_triangular(L, Kzx.to(chol_dt), upper=False).to(T)  # interp_term
        mean = A.t() @ m + c
        B = torch.tril(LS).t() @ A
        var = s + jitter + ((B * B) - (A * A)).sum(0)
        return mean, var.clamp_min(MIN_VARIANCE)

    trace = []
    L = None
    for _ in range(training_iter):
        L = chol_factor()
        mean, var = q_f(X, L)
        f = torch.sqrt(2.0 * var)[None, :] * t_k[:, None] + mean[None, :]
        ell_terms = (torch.special.log_ndtr(f * y[None, :]) * w_k[:, None]).sum(0) / math.sqrt(math.pi)
        LSt = torch.tril(LS)
        kl = 0.5 * ((LSt * LSt).sum() + (m * m).sum() - M - torch.log(torch.diagonal(LSt) ** 2).sum())
        loss = -(ell_terms.sum() / M - kl / M)
        opt.zero_grad()
        loss.backward()
        opt.step()
        trace.append(float(loss.item()))
    with torch.no_grad():
        if eval_chol == "fresh" or L is None:
            L = chol_factor()
        mean, var = q_f(Xt, L.detach())
        probs = torch.special.ndtr(mean / torch.sqrt(1.0 + var))  # BernoulliLikelihood.marginal
    out = (mean.double().numpy(), var.double().numpy(), probs.double().numpy())
    if return_trace:
        state = dict(Z=Z.detach().double().numpy(), m=m.detach().double().numpy(),
                     LS=LS.detach().double().numpy(), c=float(c.detach()), rho_s=float(rho_s.detach()),
                     rho_l=float(rho_l.detach()),
                     loss=trace)
        return out, state
    return out


# ------------------------------------------------------------------------------------------
# NumPy float64 restatement with the hand-derived backward (what the HIP kernel does)
# ------------------------------------------------------------------------------------------
def log_ndtr_and_ratio(z):
    """log Phi(z) and r(z) = phi(z) / Phi(z), stable for both tails (float64).

    z < 0:  Phi = erfcx(-z/sqrt2) * exp(-z^2/2) / 2.   z >= 0: Phi = 1 - erfc(z/sqrt2)/2.
    The HIP kernel uses the same two branches (ocml erfcx / erfc).
    """
    from scipy.special import erfc, erfcx

    z = np.asarray(z, dtype=np.float64)
    neg = z < 0
    zn = np.where(neg, z, 0.0)
    zp = np.where(neg, 0.0, z)
    ex = erfcx(-zn / math.sqrt(2.0))
    logphi_neg = np.log(0.5 * ex) - 0.5 * zn * zn
    r_neg = math.sqrt(2.0 / math.pi) / ex
    tail = 0.5 * erfc(zp / math.sqrt(2.0))
    logphi_pos = np.log1p(-tail)
    r_pos = np.exp(-0.5 * zp * zp) / math.sqrt(2.0 * math.pi) / (1.0 - tail)
    return np.where(neg, logphi_neg, logphi_pos), np.where(neg, r_neg, r_pos)


def _tril_solve(L, Bm):
    from scipy.linalg import solve_triangular

    return solve_triangular(L, Bm, lower=True)


def _triu_solve_T(L, Bm):
    from scipy.linalg import solve_triangular

    return solve_triangular(L, Bm, lower=True, trans="T")


def svgp_loss_and_grads(X, y, Z, m, LS, c, rho_s, rho_l, jitter=JITTER):
    """One ELBO evaluation + hand-derived gradients (float64).  Returns (loss, grads dict)."""
    M, D = Z.shape
    N = X.shape[0]
    ell = _softplus(rho_l)
    s = _softplus(rho_s)
    inv_l2 = 1.0 / (ell * ell)
    d2zz = ((Z[:, None, :] - Z[None, :, :]) ** 2).sum(-1)
    d2zx = ((Z[:, None, :] - X[None, :, :]) ** 2).sum(-1)
    Ezz = np.exp(-0.5 * d2zz * inv_l2)
    Ezx = np.exp(-0.5 * d2zx * inv_l2)
    Kzz = s * Ezz + jitter * np.eye(M)
    Kzx = s * Ezx
    L = np.linalg.cholesky(Kzz)
    A = _tril_solve(L, Kzx)  # M x N
    LSt = np.tril(LS)
    mean = A.T @ m + c
    Bm = LSt.T @ A
    var_raw = s + jitter + (Bm * Bm - A * A).sum(0)
    clamped = var_raw < MIN_VARIANCE
    var = np.where(clamped, MIN_VARIANCE, var_raw)
    sd = np.sqrt(2.0 * var)
    f = sd[None, :] * _GH_T[:, None] + mean[None, :]
    lp, r = log_ndtr_and_ratio(f * y[None, :])
    ipi = 1.0 / math.sqrt(math.pi)
    E = ipi * (lp * _GH_W[:, None]).sum(0)
    kl = 0.5 * ((LSt * LSt).sum() + (m * m).sum() - M - np.log(np.diagonal(LSt) ** 2).sum())
    loss = -(E.sum() / N - kl / N)

    # ---- backward of loss -------------------------------------------------------------
    dE_dmu = ipi * (r * _GH_W[:, None]).sum(0) * y
    dE_dvar = ipi * (r * (_GH_W * _GH_T)[:, None]).sum(0) * y / sd
    g_mu = -dE_dmu / N
    g_v = np.where(clamped, 0.0, -dE_dvar / N)
    G_m = A @ g_mu + m / N
    G_c = g_mu.sum()
    G_B = 2.0 * Bm * g_v[None, :]
    G_LS = np.tril(A @ G_B.T) + (LSt - np.diag(1.0 / np.diagonal(LSt))) / N
    G_A = np.outer(m, g_mu) + LSt @ G_B - 2.0 * A * g_v[None, :]
    G_Kzx = _triu_solve_T(L, G_A)  # L^-T G_A
    G_L = -np.tril(G_Kzx @ A.T)
    P = np.tril(L.T @ G_L)
    P[np.diag_indices(M)] *= 0.5
    G_Kzz = _triu_solve_T(L, _triu_solve_T(L, P.T).T)  # L^-T P L^-1
    G_Kzz = 0.5 * (G_Kzz + G_Kzz.T)
    Wzz = G_Kzz * (s * Ezz)
    Wzx = G_Kzx * Kzx
    G_s = g_v.sum() + (G_Kzz * Ezz).sum() + (G_Kzx * Ezx).sum()
    G_l = ((Wzz * d2zz).sum() + (Wzx * d2zx).sum()) / (ell ** 3)
    # dK/dZ_i = -K (Z_i - .) / l^2 ; Kzz appears with Z on both sides (G_Kzz symmetric)
    G_Z = -inv_l2 * (2.0 * (Wzz.sum(1)[:, None] * Z - Wzz @ Z) + (Wzx.sum(1)[:, None] * Z - Wzx @ X))
    grads = dict(Z=G_Z, m=G_m, LS=G_LS, c=G_c, rho_s=G_s * _sigmoid(rho_s), rho_l=G_l * _sigmoid(rho_l))
    return loss, grads


def svgp_predict(Xt, Z, m, LS, c, rho_s, rho_l, jitter=JITTER, L=None):
    ell = _softplus(rho_l)
    s = _softplus(rho_s)
    inv_l2 = 1.0 / (ell * ell)
    M = Z.shape[0]
    if L is None:
        d2zz = ((Z[:, None, :] - Z[None, :, :]) ** 2).sum(-1)
        L = np.linalg.cholesky(s * np.exp(-0.5 * d2zz * inv_l2) + jitter * np.eye(M))
    d2zx = ((Z[:, None, :] - Xt[None, :, :]) ** 2).sum(-1)
    A = _tril_solve(L, s * np.exp(-0.5 * d2zx * inv_l2))
    mean = A.T @ m + c
    Bm = np.tril(LS).T @ A
    var = np.maximum(s + jitter + (Bm * Bm - A * A).sum(0), MIN_VARIANCE)
    from scipy.special import ndtr

    return mean, var, ndtr(mean / np.sqrt(1.0 + var))


def svgp_fit_predict_manual(train_x, train_y, test_x, training_iter=50, init_mean=None, jitter=JITTER,
                            lr=ADAM_LR, return_trace=False):
    X = np.asarray(train_x, dtype=np.float64)
    y = np.asarray(train_y, dtype=np.float64)
    Xt = np.asarray(test_x, dtype=np.float64)
    M, D = X.shape
    P = dict(Z=X.copy(), m=np.zeros(M) if init_mean is None else np.asarray(init_mean, dtype=np.float64).copy(),
             LS=np.eye(M), c=0.0, rho_s=0.0, rho_l=0.0)
    m1 = {k: np.zeros_like(np.asarray(v, dtype=np.float64)) for k, v in P.items()}
    m2 = {k: np.zeros_like(np.asarray(v, dtype=np.float64)) for k, v in P.items()}
    trace = []
    for t in range(1, training_iter + 1):
        loss, G = svgp_loss_and_grads(X, y, P["Z"], P["m"], P["LS"], P["c"], P["rho_s"], P["rho_l"], jitter)
        trace.append(loss)
        bc1 = 1.0 - ADAM_B1 ** t
        bc2 = 1.0 - ADAM_B2 ** t
        for k in P:
            g = np.asarray(G[k], dtype=np.float64)
            m1[k] = ADAM_B1 * m1[k] + (1.0 - ADAM_B1) * g
            m2[k] = ADAM_B2 * m2[k] + (1.0 - ADAM_B2) * g * g
            denom = np.sqrt(m2[k]) / math.sqrt(bc2) + ADAM_EPS
            P[k] = P[k] - (lr / bc1) * m1[k] / denom
    out = svgp_predict(Xt, P["Z"], P["m"], P["LS"], float(P["c"]), float(P["rho_s"]), float(P["rho_l"]), jitter)
    if return_trace:
        return out, dict(P, loss=trace)
    return out


# ------------------------------------------------------------------------------------------
# fit_gp_spp restatement (reference gaussian_process_utils.py:382-445)
# ------------------------------------------------------------------------------------------
def fit_gp_spp_oracle(feats_spp, b1_inds, b2_inds, intersect_inds, training_iter=50, impl="manual",
                      dtype="f64", init_mean=None):
    feats_spp = np.asarray(feats_spp, dtype=np.float32)
    b1 = np.asarray(b1_inds, dtype=np.int64)
    b2 = np.asarray(b2_inds, dtype=np.int64)
    it = np.asarray(intersect_inds, dtype=np.int64)
    train_x = np.concatenate([feats_spp[b1], feats_spp[b2]], 0)  # :395
    train_y = np.concatenate([-np.ones(len(b1)), np.ones(len(b2))])  # :396-398
    test_x = feats_spp[it]  # :386
    if impl == "manual":
        mu, var, p = svgp_fit_predict_manual(train_x, train_y, test_x, training_iter, init_mean=init_mean)
    else:
        mu, var, p = svgp_fit_predict_autograd(train_x, train_y, test_x, training_iter, dtype=dtype,
                                               init_mean=init_mean)
    pred_probs = p.astype(np.float32)  # :432
    pred_labels = pred_probs >= np.float32(0.5)  # :433
    pred_probs_new = np.where(pred_labels, pred_probs, np.float32(1.0) - pred_probs).astype(np.float32)  # :438
    return pred_probs, pred_probs_new, pred_labels, mu.astype(np.float32), var.astype(np.float32)
