"""CPU (torch) restatement of the consumer-side label ops.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The expressions are the reference's own lines (no third-party code involved), evaluated with torch on the CPU:
* ``scatter_mean3``     -- ISBNet/isbnet/model/model_utils.py:600-613 (torch_scatter.scatter_mean in float32; the
  count is clamped at 1), used as isbnet.py:387-389 does.
* ``weighted_bce``      -- ISBNet/isbnet/model/criterion.py:287-288.
* ``kl_to_gp``          -- ISBNet/isbnet/model/criterion.py:435-463.
Parity unpinned against a reference run (torch_scatter / spconv are not installed here); the formulas are short
enough to be checked by eye against the cited lines, and gradients are checked against torch autograd of these.
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


def weighted_bce(mask_logit_pred, inst_label, prob_labels_b):
    num_gt_batch = mask_logit_pred.shape[0]
    bce_loss = F.binary_cross_entropy_with_logits(mask_logit_pred, inst_label, reduction="none")
    return ((bce_loss * prob_labels_b).sum() / prob_labels_b.sum()).sum() / (num_gt_batch + 1e-6)


def kl_to_gp(mu_pred, logvar_pred, mu_labels, var_labels, weight=1.0, epsilon=1e-4):
    loss = torch.zeros((), dtype=mu_pred.dtype)
    mask_kl_varzero = (mu_labels != -100) & (var_labels != -100) & (var_labels <= epsilon)
    mask_kl_var = (mu_labels != -100) & (var_labels != -100) & (var_labels > epsilon)
    if mask_kl_varzero.sum() > 0:
        l0 = (torch.exp(logvar_pred[mask_kl_varzero]) - 1) ** 2 + (mu_pred[mask_kl_varzero] - mu_labels[mask_kl_varzero]) ** 2
        loss = loss + l0.sum() / (mask_kl_varzero.sum() + 1e-4) * weight
    if mask_kl_var.sum() > 0:
        l1 = ((logvar_pred[mask_kl_var] - torch.log(var_labels[mask_kl_var]))
              + ((mu_pred[mask_kl_var] - mu_labels[mask_kl_var]) ** 2 + var_labels[mask_kl_var] ** 2)
              * (torch.exp(-2 * logvar_pred[mask_kl_var])) - 0.5)
        loss = loss + l1.sum() / (mask_kl_var.sum() + 1e-4) * weight
    return loss
