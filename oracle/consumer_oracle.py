"""CPU (torch) restatement of the consumer-side label ops.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The expressions are the reference's own lines (no third-party code involved), evaluated with torch on the CPU:
* ``scatter_mean3``     -- ISBNet/isbnet/model/model_utils.py:600-613 (torch_scatter.scatter_mean in float32; the
  count is clamped at 1), used as isbnet.py:387-389 does.
* ``weighted_bce``      -- ISBNet/isbnet/model/criterion.py:287-288.
* ``kl_to_gp``          -- ISBNet/isbnet/model/criterion.py:435-463.
Pinned (round 3): tests/golden/make_golden_consumer.py imports the reference's ``custom_scatter_mean`` and executes
the two blocks of criterion.py in the build container (stubs only for the uninstalled torch_scatter / isbnet.ops);
tests/test_consumer_golden.py holds these restatements to its fixtures, values and autograd gradients.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def scatter_mean3(prob, mu, var, idx, n_out=None):
    idx = idx.long()
    n_out = int(idx.max()) + 1 if n_out is None else n_out
    cnt = torch.bincount(idx, minlength=n_out).clamp(min=1).to(torch.float32)
    outs = []
    for t in (prob, mu, var):
        s = torch.zeros(n_out, dtype=torch.float32).index_add_(0, idx, t.to(torch.float32))
        outs.append(s / cnt)
    return tuple(outs)


def weighted_bce(logits, targets, point_weights):
    """criterion.py:287-288: per-element BCE-with-logits, weighted per point (column), normalised by the weight
    sum and by the number of rows (+1e-6)."""
    rows = logits.shape[0]
    per_elem = torch.clamp(logits, min=0) - logits * targets + torch.log1p(torch.exp(-logits.abs()))
    return (per_elem * point_weights[None, :]).sum() / point_weights.sum() / (rows + 1e-6)


def kl_to_gp(mu_pred, logvar_pred, mu_labels, var_labels, weight=1.0, epsilon=1e-4):
    """criterion.py:435-463: labelled entries (neither label is -100) split by the GP variance; tiny variances
    pull the predicted variance to 1 and the mean to the label, the others use the Gaussian KL expression; each
    group is averaged over its own size (+1e-4) and scaled by ``weight``; an empty group contributes nothing."""
    labelled = (mu_labels != -100) & (var_labels != -100)
    tiny = labelled & (var_labels <= epsilon)
    rest = labelled & (var_labels > epsilon)
    total = torch.zeros((), dtype=mu_pred.dtype)
    if bool(tiny.any()):
        dmu = mu_pred[tiny] - mu_labels[tiny]
        term = (logvar_pred[tiny].exp() - 1.0).square() + dmu.square()
        total = total + weight * term.sum() / (tiny.sum() + 1e-4)
    if bool(rest.any()):
        v = var_labels[rest]
        lv = logvar_pred[rest]
        dmu = mu_pred[rest] - mu_labels[rest]
        term = (lv - v.log()) + (dmu.square() + v.square()) * torch.exp(-2.0 * lv) - 0.5
        total = total + weight * term.sum() / (rest.sum() + 1e-4)
    return total
