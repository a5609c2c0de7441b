"""Self-consistency of the GP oracle (CPU): hand-derived backward == torch autograd, known-answer
checks of the ELBO at initialisation, and the precision/noise study DESIGN.md quotes."""
import math

import numpy as np
import pytest

from gapro_amd.synth import make_gp_problem
from oracle import svgp_oracle as so


def _problem(seed, m1, m2, t, d=6):
    feats, b1, b2, it = make_gp_problem(seed, m1, m2, t, d)
    X = np.concatenate([feats[b1], feats[b2]]).astype(np.float64)
    y = np.r_[-np.ones(m1), np.ones(m2)]
    return X, y, feats[it].astype(np.float64)


def test_initial_elbo_is_the_analytic_value():
    # at init mu = 0, var = s + eps for every point, KL = 0: loss = -E[log Phi(sqrt(var) z)]
    X, y, Xt = _problem(0, 7, 9, 3)
    loss, _ = so.svgp_loss_and_grads(X, y, X.copy(), np.zeros(16), np.eye(16), 0.0, 0.0, 0.0)
    t, w = np.polynomial.hermite.hermgauss(20)
    from scipy.special import log_ndtr
    var = math.log(2.0) + 1e-4
    expect = -(w * log_ndtr(math.sqrt(2 * var) * t)).sum() / math.sqrt(math.pi)
    assert abs(loss - expect) < 1e-13
    assert abs(loss - 0.9079288393362) < 1e-10  # SURVEY Appendix D, probe 5


def test_manual_backward_equals_autograd():
    X, y, Xt = _problem(1, 12, 15, 6)
    (mu_a, var_a, p_a), st_a = so.svgp_fit_predict_autograd(X, y, Xt, 8, "f64", return_trace=True)
    (mu_m, var_m, p_m), st_m = so.svgp_fit_predict_manual(X, y, Xt, 8, return_trace=True)
    np.testing.assert_allclose(st_m["loss"], st_a["loss"], rtol=1e-11)
    np.testing.assert_allclose(mu_m, mu_a, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(var_m, var_a, rtol=1e-9)
    np.testing.assert_allclose(st_m["Z"], st_a["Z"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(np.tril(st_m["LS"]), np.tril(st_a["LS"]), rtol=1e-8, atol=1e-10)


def test_loss_decreases_and_labels_separate_blobs():
    feats, b1, b2, it = make_gp_problem(5, 25, 25, 40, 6, sep=4.0)
    probs, probs_new, labels, mu, var = so.fit_gp_spp_oracle(feats, b1, b2, it, 50)
    assert probs.dtype == np.float32 and labels.dtype == bool and var.dtype == np.float32
    assert (probs_new >= 0.5).all() and (var > 0).all()
    # test points were drawn between the blobs with weight w towards blob 2: most are separable
    assert 5 < labels.sum() < 35


def test_log_ndtr_branches():
    from scipy.special import log_ndtr
    z = np.array([-40.0, -12.0, -3.0, -1e-9, 0.0, 1e-9, 2.0, 9.0])
    lp, r = so.log_ndtr_and_ratio(z)
    np.testing.assert_allclose(lp, log_ndtr(z), rtol=1e-12, atol=1e-300)
    phi = np.exp(-0.5 * z * z) / math.sqrt(2 * math.pi)
    np.testing.assert_allclose(r[2:], phi[2:] / np.exp(log_ndtr(z[2:])), rtol=1e-12)
    assert abs(r[0] - 40.0) / 40.0 < 1e-3  # phi/Phi -> -z in the far left tail


@pytest.mark.parametrize("m1,m2,t", [(20, 30, 10)])
def test_precision_and_init_noise_study(m1, m2, t):
    """What 'GP variances within 1e-4 rel' can mean ('Precision'):
    the reference's own float32/float64 split stays within ~1e-5 of the float64 ground truth,
    while its unseeded 1e-3*randn initial mean moves the variances by percents."""
    X, y, Xt = _problem(1, m1, m2, t)
    mu64, var64, _ = so.svgp_fit_predict_autograd(X, y, Xt, 50, "f64")
    mu32, var32, _ = so.svgp_fit_predict_autograd(X, y, Xt, 50, "mixed")
    assert np.max(np.abs(var32 - var64) / var64) < 1e-4
    rng = np.random.default_rng(0)
    _, var_r, _ = so.svgp_fit_predict_autograd(X, y, Xt, 50, "f64", init_mean=1e-3 * rng.standard_normal(m1 + m2))
    assert np.max(np.abs(var_r - var64) / var64) > 1e-3


def test_the_one_ill_conditioned_s3dis_fit_drifts_between_the_oracles_own_implementations():
    """Evidence for the tolerance of tests/test_pipeline_gpu.py::test_s3dis_shaped_scene_matches_oracle (BASELINE
    configs[3]).  Of that scene's 66 GP fits exactly one is not reproducible to float32 rounding by any two float64
    implementations: here the oracle's own two (autograd / hand-derived backward: the same formulas in a different
    summation order) end > 1e-5 apart in sigma^2 on it after 50 Adam steps and < 1e-7 apart on the control fit, the
    most drifting of the other 65 (tests/golden/make_s3dis_fits.py measured all 66).  Early steps agree to 1e-9 on
    both: the gap is amplified rounding noise, not a formula difference.  The same fit moves by as much when a single
    input coordinate is changed by one ulp (on the GPU box's host a second fit, M = 58, behaves the
    same way at 3e-6: tools/loose_fits.py)."""
    import os

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fits_s3dis.npz"))
    out = {}
    for tag in ("ill", "ctl"):
        f, m1, m2 = z[tag + "_feats"], int(z[tag + "_m1"]), int(z[tag + "_m2"])
        X, Xt = f[:m1 + m2].astype(np.float64), f[m1 + m2:].astype(np.float64)
        y = np.r_[-np.ones(m1), np.ones(m2)]
        (_, va, _), sa = so.svgp_fit_predict_autograd(X, y, Xt, 50, "f64", return_trace=True)
        (_, vm, _), sm = so.svgp_fit_predict_manual(X, y, Xt, 50, return_trace=True)
        out[tag] = float(np.max(np.abs(va - vm) / va))
        np.testing.assert_allclose(sm["loss"][:5], sa["loss"][:5], rtol=1e-9)
        # the fit's own conditioning (round 5): ONE input coordinate moved by ONE ulp, same implementation
        Xp = X.copy()
        Xp[0, 0] = np.nextafter(Xp[0, 0], np.inf)
        _, vp, _ = so.svgp_fit_predict_autograd(Xp, y, Xt, 50, "f64")
        out[tag + "_ulp"] = float(np.max(np.abs(vp - va) / va))
    assert out["ctl"] < 1e-7 < 1e-5 < out["ill"], out
    # measured 3.9e-5 on the ill-conditioned fit -- and 5.8e-6 on the control (M = 58): the control's two implementations
    # happen to agree to 3e-8 on this host and part by 2.7e-6 on the GPU box's; it is the scene's second fit of this kind
    assert out["ill_ulp"] > 1e-6 and out["ctl_ulp"] > 1e-7, out
    assert abs(np.log10(out["ill"] / float(z["ill_drift"][0]))) < 1.5  # same order as when the file was written
