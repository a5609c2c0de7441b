"""oracle/consumer_oracle.py held to outputs of the REFERENCE's own consumer-side code:
``tests/golden/consumer_*.npz`` were written by tests/golden/make_golden_consumer.py, which imports
ISBNet/isbnet/model/model_utils.py::custom_scatter_mean and executes criterion.py:287-288 / :435-463 in the build
container.  CPU only: this pins the oracle; tests/test_consumer_gpu.py holds the HIP ops to the same fixtures."""
import os

import numpy as np
import pytest
import torch

from oracle import consumer_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_scatter_mean_restated_equals_reference_function():
    z = np.load(os.path.join(G, "consumer_pool.npz"))
    idx = torch.from_numpy(z["idx"])
    got = O.scatter_mean3(torch.from_numpy(z["prob"]), torch.from_numpy(z["mu"]), torch.from_numpy(z["var"]), idx)
    for a, k in zip(got, ("ref_prob", "ref_mu", "ref_var")):
        assert a.dtype == torch.float32 and len(a) == len(z[k])  # output length = largest id present + 1
        np.testing.assert_array_equal(a.numpy(), z[k])           # same float32 sums in the same order: same bits
    assert float(z["ref_prob"][7]) == 0.0                        # the empty superpoint: count clamped at 1
    # the reference casts the pooled means back to the input type (model_utils.py:608-611)
    ph = torch.from_numpy(z["prob"]).half()
    got_h = O.scatter_mean3(ph.float(), ph.float(), ph.float(), idx)[0].half()
    np.testing.assert_array_equal(got_h.float().numpy(), z["ref_prob_half"])


def test_weighted_bce_restated_equals_reference_lines():
    z = np.load(os.path.join(G, "consumer_losses.npz"))
    for k in range(int(z["bce_cases"])):
        x = torch.from_numpy(z["bce%d_logits" % k]).requires_grad_(True)
        y, w = torch.from_numpy(z["bce%d_target" % k]), torch.from_numpy(z["bce%d_w" % k])
        loss = O.weighted_bce(x, y, w)
        (g,) = torch.autograd.grad(loss, x)
        np.testing.assert_allclose(float(loss.detach()), float(z["bce%d_loss" % k]), rtol=2e-6)
        np.testing.assert_allclose(g.numpy(), z["bce%d_grad" % k], rtol=2e-5, atol=1e-9)
        # and in float64 the restatement is the same function to 1e-7 (the reference evaluates in float32)
        l64 = O.weighted_bce(x.detach().double(), y.double(), w.double())
        np.testing.assert_allclose(float(l64), float(z["bce%d_loss" % k]), rtol=2e-6)


def test_kl_to_gp_restated_equals_reference_lines_every_branch():
    z = np.load(os.path.join(G, "consumer_losses.npz"))
    w = float(z["kl_weight"])
    for k in range(int(z["kl_cases"])):
        mu_p = torch.from_numpy(z["kl%d_mu_p" % k]).requires_grad_(True)
        lv_p = torch.from_numpy(z["kl%d_lv_p" % k]).requires_grad_(True)
        mu_l, var_l = torch.from_numpy(z["kl%d_mu_l" % k]), torch.from_numpy(z["kl%d_var_l" % k])
        loss = O.kl_to_gp(mu_p, lv_p, mu_l, var_l, weight=w)
        np.testing.assert_allclose(float(loss.detach()), float(z["kl%d_loss" % k]), rtol=5e-6, atol=1e-30)
        if float(z["kl%d_loss" % k]) != 0.0:
            gm, gl = torch.autograd.grad(loss, (mu_p, lv_p))
            np.testing.assert_allclose(gm.numpy(), z["kl%d_gmu" % k], rtol=2e-5, atol=1e-10)
            np.testing.assert_allclose(gl.numpy(), z["kl%d_glv" % k], rtol=2e-5, atol=1e-10)
        else:
            assert float(loss) == 0.0  # nothing labelled: the reference's zero tensor
