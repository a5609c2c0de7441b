#!/usr/bin/env python3
"""Generate tests/golden/consumer_*.npz by running the REFERENCE's consumer-side code in this container
(the first step on the other side of the .pth file boundary).

Run from the repo root:  python tests/golden/make_golden_consumer.py
Needs /root/reference (read-only).  What is executed from the reference:

* ``custom_scatter_mean`` -- imported as-is from ISBNet/isbnet/model/model_utils.py:600-613, used the way
  isbnet.py:387-389 uses it (prob / mu / var pooled to superpoints).  ``torch_scatter`` is not installed here; it is
  replaced IN THIS HARNESS ONLY by the same pure-torch shim tests/golden/make_golden.py uses (sequential float32
  sums, count clamped at 1 -- the CPU semantics of torch_scatter.scatter_mean).  ``isbnet.ops`` (a CUDA extension
  model_utils imports but these lines never call) is an empty stub.
* the probability-weighted BCE, ISBNet/isbnet/model/criterion.py:287-288, and the KL-to-GP loss, :435-463.  They sit
  in the middle of ``Criterion.single_layer_loss`` / ``Criterion.forward`` (which need a full network output), so the
  generator reads exactly those source lines from the reference file AT GENERATION TIME, dedents them and executes
  them on small tensors with the names they use bound to the inputs below.  Nothing of the reference's text is
  written to the fixture or to this repo: the fixture holds inputs, the values those lines produced and the
  gradients torch autograd gave for them.

The fixtures are plain data and travel to the GPU box; the reference does not.
"""
import importlib
import os
import sys
import textwrap
import types

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/ISBNet"
OUT = os.path.dirname(os.path.abspath(__file__))


def install_stubs():
    ts = types.ModuleType("torch_scatter")

    def scatter_mean(src, index, dim=0, out=None, dim_size=None):
        assert dim == 0 and out is None
        n = int(index.max()) + 1 if dim_size is None else dim_size
        res = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype).index_add_(0, index, src)  # sequential on CPU
        cnt = torch.zeros(n, dtype=src.dtype).index_add_(0, index, torch.ones(src.shape[0], dtype=src.dtype))
        cnt.clamp_(1)
        return res / (cnt[:, None] if res.dim() > 1 else cnt)

    ts.scatter_mean = scatter_mean
    sys.modules["torch_scatter"] = ts
    # package shells so that ``from ..ops import ballquery_batchflat`` resolves without running isbnet/__init__.py
    # (which pulls in spconv and the compiled ops)
    pkg = types.ModuleType("isbnet")
    pkg.__path__ = [os.path.join(REF, "isbnet")]
    ops = types.ModuleType("isbnet.ops")
    ops.ballquery_batchflat = None
    model = types.ModuleType("isbnet.model")
    model.__path__ = [os.path.join(REF, "isbnet", "model")]
    sys.modules.update({"isbnet": pkg, "isbnet.ops": ops, "isbnet.model": model})


def ref_lines(path, lo, hi):
    """Source lines lo..hi (1-based, inclusive) of a reference file, dedented."""
    with open(path) as f:
        src = f.readlines()[lo - 1:hi]
    return textwrap.dedent("".join(src))


def main():
    install_stubs()
    mu_mod = importlib.import_module("isbnet.model.model_utils")
    crit_path = os.path.join(REF, "isbnet", "model", "criterion.py")
    bce_src = ref_lines(crit_path, 287, 288)
    kl_src = ref_lines(crit_path, 435, 463)
    assert "binary_cross_entropy_with_logits" in bce_src and "prob_labels_b" in bce_src, "criterion.py moved"
    assert "mask_kl_varzero" in kl_src and "loss_kl_var" in kl_src, "criterion.py moved"

    # ---- custom_scatter_mean on prob / mu / var, as isbnet.py:387-389 ---------------------------------------------
    g = torch.Generator().manual_seed(20260301)
    n, s = 12000, 300
    idx = torch.randint(0, s, (n,), generator=g)
    idx[idx == 7] = 8          # an empty superpoint
    idx[idx == s - 1] = s - 2  # and the last one empty: dim_size comes from the largest id present
    prob = torch.rand(n, generator=g)
    mu = torch.where(torch.rand(n, generator=g) < 0.3, torch.full((n,), -100.0), torch.randn(n, generator=g))
    var = torch.where(mu == -100, torch.full((n,), -100.0), torch.rand(n, generator=g))
    pooled = [mu_mod.custom_scatter_mean(t, idx, dim=0, pool=True) for t in (prob, mu, var)]
    half = mu_mod.custom_scatter_mean(prob.half(), idx, dim=0, pool=True)  # output cast back to the input type
    assert half.dtype == torch.float16 and mu_mod.custom_scatter_mean(prob, idx, pool=False) is prob
    np.savez_compressed(os.path.join(OUT, "consumer_pool.npz"), idx=idx.numpy(), prob=prob.numpy(), mu=mu.numpy(),
                        var=var.numpy(), ref_prob=pooled[0].numpy(), ref_mu=pooled[1].numpy(),
                        ref_var=pooled[2].numpy(), ref_prob_half=half.float().numpy())

    # ---- probability-weighted BCE (criterion.py:287-288) ---------------------------------------------------------------
    rec = {}
    for k, (G, P) in enumerate([(1, 700), (7, 1111), (40, 300)]):
        g = torch.Generator().manual_seed(100 + k)
        logits = (4 * torch.randn(G, P, generator=g)).requires_grad_(True)
        target = (torch.rand(G, P, generator=g) < 0.3).float()
        w = (0.5 + 0.5 * torch.rand(P, generator=g))
        w[::17] = 1.0  # determined points carry probability one
        ns = dict(F=F, torch=torch, mask_logit_pred=logits, inst_label=target, prob_labels_b=w, num_gt_batch=G)
        exec(bce_src, ns)
        loss = ns["bce_loss"]
        (grad,) = torch.autograd.grad(loss, logits)
        rec.update({"bce%d_logits" % k: logits.detach().numpy(), "bce%d_target" % k: target.numpy(),
                    "bce%d_w" % k: w.numpy(), "bce%d_loss" % k: loss.detach().numpy(),
                    "bce%d_grad" % k: grad.numpy()})
    rec["bce_cases"] = np.int64(3)

    # ---- KL-to-GP loss (criterion.py:435-463) ------------------------------------------------------------------------------
    class _Self:
        loss_weight = {"kl_loss": 0.7}

    cases = 0
    for k, kind in enumerate(["mixed", "only_tiny", "only_var", "none"]):
        g = torch.Generator().manual_seed(200 + k)
        n = 3000
        mu_l = torch.randn(n, generator=g)
        var_l = torch.rand(n, generator=g) * 0.5 + 2e-4
        r = torch.rand(n, generator=g)
        if kind in ("mixed", "only_tiny"):
            var_l[r < (0.2 if kind == "mixed" else 2.0)] = 5e-5   # the var <= epsilon branch
        if kind == "mixed":
            var_l[(r >= 0.2) & (r < 0.25)] = 1e-4                 # exactly epsilon: '<=' side
        ign = (r >= 0.3) & (r < 0.55) if kind != "none" else torch.ones(n, dtype=torch.bool)
        mu_l[ign] = -100.0
        var_l[ign] = -100.0
        mu_p = torch.randn(n, generator=g).requires_grad_(True)
        lv_p = (0.5 * torch.randn(n, generator=g)).requires_grad_(True)
        loss_dict = {}
        ns = dict(torch=torch, self=_Self(), loss_dict=loss_dict, instance_labels=mu_l,
                  model_outputs={"dc_mu_labels": mu_l, "dc_var_labels": var_l, "mu_pred": mu_p, "logvar_pred": lv_p})
        exec(kl_src.replace("return loss_dict", ""), ns)  # the block ends the method: drop the bare return
        loss = loss_dict["kl_loss"]
        if kind == "none":
            gm, gl = torch.zeros(n), torch.zeros(n)
        else:
            gm, gl = torch.autograd.grad(loss, (mu_p, lv_p))
        rec.update({"kl%d_mu_l" % k: mu_l.numpy(), "kl%d_var_l" % k: var_l.numpy(),
                    "kl%d_mu_p" % k: mu_p.detach().numpy(), "kl%d_lv_p" % k: lv_p.detach().numpy(),
                    "kl%d_loss" % k: loss.detach().numpy(), "kl%d_gmu" % k: gm.numpy(), "kl%d_glv" % k: gl.numpy()})
        cases += 1
    rec["kl_cases"] = np.int64(cases)
    rec["kl_weight"] = np.float64(_Self.loss_weight["kl_loss"])
    np.savez_compressed(os.path.join(OUT, "consumer_losses.npz"), **rec)
    print("wrote consumer_pool.npz, consumer_losses.npz")


if __name__ == "__main__":
    main()
