"""HIP consumer-side label ops against the torch restatement of the reference lines:
values and gradients within float32 round-off (the kernels reduce in float64)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 2e-6


def test_pool_labels_to_superpoints_matches_scatter_mean():
    import torch
    from gapro_amd.consumer_ops import pool_labels_to_superpoints
    from oracle.consumer_oracle import scatter_mean3

    g = torch.Generator().manual_seed(0)
    n, s = 120000, 1500
    idx = torch.randint(0, s, (n,), generator=g)
    idx[idx == 7] = 8  # an empty superpoint: mean 0 (count clamped at 1)
    prob = torch.rand(n, generator=g)
    mu = torch.where(torch.rand(n, generator=g) < 0.3, torch.full((n,), -100.0), torch.randn(n, generator=g))
    var = torch.where(mu == -100, torch.full((n,), -100.0), torch.rand(n, generator=g))
    ref = scatter_mean3(prob, mu, var, idx, s)
    got = pool_labels_to_superpoints(prob.cuda(), mu.cuda(), var.cuda(), idx.cuda(), s)
    for a, b in zip(got, ref):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=1e-5, atol=1e-6)
    assert float(got[0][7]) == 0.0 and got[0].dtype == torch.float32 and len(got[1]) == s


@pytest.mark.parametrize("G,P", [(1, 1000), (7, 33333), (40, 15000)])
def test_prob_weighted_bce_value_and_gradient(G, P):
    import torch
    from gapro_amd.consumer_ops import prob_weighted_bce_with_logits
    from oracle.consumer_oracle import weighted_bce

    g = torch.Generator().manual_seed(G)
    x = (4 * torch.randn(G, P, generator=g)).requires_grad_(True)
    y = (torch.rand(G, P, generator=g) < 0.3).float()
    w = 0.5 + 0.5 * torch.rand(P, generator=g)
    ref = weighted_bce(x.double(), y.double(), w.double())
    (gref,) = torch.autograd.grad(ref * 3.0, x)
    xd = x.detach().cuda().requires_grad_(True)
    got = prob_weighted_bce_with_logits(xd, y.cuda(), w.cuda())
    (3.0 * got).backward()
    np.testing.assert_allclose(float(got.detach()), float(ref), rtol=RTOL)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), gref.numpy(), rtol=2e-6, atol=1e-30)


def test_kl_to_gp_value_and_gradient_all_branches():
    import torch
    from gapro_amd.consumer_ops import kl_to_gp_loss
    from oracle.consumer_oracle import kl_to_gp

    g = torch.Generator().manual_seed(3)
    n = 50000
    mu_l = torch.randn(n, generator=g)
    var_l = torch.rand(n, generator=g) * 0.5
    r = torch.rand(n, generator=g)
    var_l[r < 0.2] = 5e-5          # the var <= epsilon branch
    mu_l[(r >= 0.2) & (r < 0.5)] = -100.0  # ignored
    var_l[(r >= 0.2) & (r < 0.5)] = -100.0
    mu_p = torch.randn(n, generator=g).requires_grad_(True)
    lv_p = (0.5 * torch.randn(n, generator=g)).requires_grad_(True)
    ref = kl_to_gp(mu_p.double(), lv_p.double(), mu_l.double(), var_l.double(), weight=0.7)
    gm_ref, gl_ref = torch.autograd.grad(ref, (mu_p, lv_p))
    a = mu_p.detach().cuda().requires_grad_(True)
    b = lv_p.detach().cuda().requires_grad_(True)
    got = kl_to_gp_loss(a, b, mu_l.cuda(), var_l.cuda(), weight=0.7)
    got.backward()
    np.testing.assert_allclose(float(got.detach()), float(ref), rtol=1e-5)
    np.testing.assert_allclose(a.grad.cpu().numpy(), gm_ref.numpy(), rtol=2e-6, atol=1e-30)
    np.testing.assert_allclose(b.grad.cpu().numpy(), gl_ref.numpy(), rtol=2e-6, atol=1e-12)
    # nothing labelled: zero loss, zero gradients
    none = torch.full((n,), -100.0)
    a2 = mu_p.detach().cuda().requires_grad_(True)
    z = kl_to_gp_loss(a2, b.detach(), none.cuda(), none.cuda())
    z.backward()
    assert float(z) == 0.0 and float(a2.grad.abs().max()) == 0.0


def test_generated_labels_feed_the_consumer_ops(golden):
    """End to end across the file boundary: labels generated for a golden scene, pooled the way the trainers pool
    them; pooled prob of a superpoint equals the superpoint's prob, mu / var at superpoint length."""
    import torch
    from gapro_amd import gen_pseudo_label_gaussian_process
    from gapro_amd.consumer_ops import pool_labels_to_superpoints

    kw = golden.api_inputs()
    sem, ins, prob, mu, var = gen_pseudo_label_gaussian_process(**kw, device="cuda:0", broadcast_mu_var=True)
    _, inv = np.unique(kw["spp"], return_inverse=True)
    inv_t = torch.from_numpy(inv).cuda()
    p_spp, mu_spp, var_spp = pool_labels_to_superpoints(prob.cuda(), mu.cuda(), var.cuda(), inv_t)
    np.testing.assert_allclose(p_spp[inv_t].cpu().numpy(), prob.cpu().numpy(), rtol=1e-6)
    np.testing.assert_allclose(mu_spp[inv_t].cpu().numpy(), mu.cpu().numpy(), rtol=1e-6)
    np.testing.assert_allclose(var_spp[inv_t].cpu().numpy(), var.cpu().numpy(), rtol=1e-6)


# ---- the same ops against outputs of the REFERENCE's own functions (tests/golden/make_golden_consumer.py) ----------
def _golden(name):
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name))


def test_pool_matches_the_reference_custom_scatter_mean():
    import torch
    from gapro_amd.consumer_ops import pool_labels_to_superpoints

    z = _golden("consumer_pool.npz")
    got = pool_labels_to_superpoints(torch.from_numpy(z["prob"]).cuda(), torch.from_numpy(z["mu"]).cuda(),
                                     torch.from_numpy(z["var"]).cuda(), torch.from_numpy(z["idx"]).cuda())
    for a, k in zip(got, ("ref_prob", "ref_mu", "ref_var")):
        assert len(a) == len(z[k])
        # the kernel sums in float64 and rounds once; torch_scatter's CPU path sums float32 sequentially
        np.testing.assert_allclose(a.cpu().numpy(), z[k], rtol=2e-6, atol=2e-6)


def test_bce_matches_the_reference_lines():
    import torch
    from gapro_amd.consumer_ops import prob_weighted_bce_with_logits

    z = _golden("consumer_losses.npz")
    for k in range(int(z["bce_cases"])):
        x = torch.from_numpy(z["bce%d_logits" % k]).cuda().requires_grad_(True)
        got = prob_weighted_bce_with_logits(x, torch.from_numpy(z["bce%d_target" % k]).cuda(),
                                            torch.from_numpy(z["bce%d_w" % k]).cuda())
        got.backward()
        np.testing.assert_allclose(float(got.detach()), float(z["bce%d_loss" % k]), rtol=3e-6)
        np.testing.assert_allclose(x.grad.cpu().numpy(), z["bce%d_grad" % k], rtol=3e-5, atol=1e-9)


def test_kl_matches_the_reference_lines_every_branch():
    import torch
    from gapro_amd.consumer_ops import kl_to_gp_loss

    z = _golden("consumer_losses.npz")
    w = float(z["kl_weight"])
    for k in range(int(z["kl_cases"])):
        a = torch.from_numpy(z["kl%d_mu_p" % k]).cuda().requires_grad_(True)
        b = torch.from_numpy(z["kl%d_lv_p" % k]).cuda().requires_grad_(True)
        got = kl_to_gp_loss(a, b, torch.from_numpy(z["kl%d_mu_l" % k]).cuda(),
                            torch.from_numpy(z["kl%d_var_l" % k]).cuda(), weight=w)
        got.backward()
        np.testing.assert_allclose(float(got.detach()), float(z["kl%d_loss" % k]), rtol=1e-5, atol=1e-30)
        np.testing.assert_allclose(a.grad.cpu().numpy(), z["kl%d_gmu" % k], rtol=3e-5, atol=1e-9)
        np.testing.assert_allclose(b.grad.cpu().numpy(), z["kl%d_glv" % k], rtol=3e-5, atol=1e-9)
