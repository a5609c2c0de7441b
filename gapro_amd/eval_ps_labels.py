"""Pseudo-label quality metric: mirror of reference gapro/eval_ps_labels.py:35-42,100-147.

SURVEY.md section 8(f) row 1 ("next"): this is not part of the generator's hot path; it is wired to
``gen_ps --eval_pslabel``.  The reference builds two one-hot matrices [I, N] and multiplies them; here
the same intersection counts come from one bincount over (gt, pseudo) id pairs (integer arithmetic,
identical values), on whatever device the labels live on.
"""
from __future__ import annotations

import torch


def _first_point_class(instance_label, semantic_label, n_inst):
    """Class of the first point of every instance id, -1 for empty ids (eval_ps_labels.py:101-108)."""
    cls = torch.full((n_inst,), -1.0, device=instance_label.device)
    valid = (instance_label >= 0) & (instance_label < n_inst)
    idx = torch.nonzero(valid).view(-1)
    if len(idx):
        ids = instance_label[idx]
        first = torch.full((n_inst,), instance_label.numel(), dtype=torch.long, device=instance_label.device)
        first.scatter_reduce_(0, ids, idx, reduce="amin")
        has = first < instance_label.numel()
        cls[has] = semantic_label[first[has]].float()
    return cls


def get_miou_scene(semantic_label, instance_label, ps_semantic_label, ps_instance_label):
    """Per GT instance: max IoU over pseudo instances of the same class (eval_ps_labels.py:100-147).

    IoU = inter / (|gt| + |ps| - inter + 1e-4), float32 as in ``cal_iou`` (:35-42)."""
    n_inst = int(instance_label.max()) + 1
    n_ps = int(ps_instance_label.max()) + 1
    if n_inst <= 0:
        return torch.zeros(0, device=instance_label.device)
    gt_cls = _first_point_class(instance_label, semantic_label, n_inst)
    ps_cls = _first_point_class(ps_instance_label, ps_semantic_label, max(n_ps, 0)) if n_ps > 0 else \
        torch.zeros(0, device=instance_label.device)
    if n_ps <= 0:
        return torch.zeros(int((gt_cls >= 0).sum()), device=instance_label.device)
    g = torch.where(instance_label < 0, torch.zeros_like(instance_label), instance_label + 1).long()
    p = torch.where(ps_instance_label < 0, torch.zeros_like(ps_instance_label), ps_instance_label + 1).long()
    pair = torch.bincount(g * (n_ps + 1) + p, minlength=(n_inst + 1) * (n_ps + 1)).view(n_inst + 1, n_ps + 1)
    inter = pair[1:, 1:].float()
    gt_n = pair[1:, :].sum(1, keepdim=True).float()
    ps_n = pair[:, 1:].sum(0, keepdim=True).float()
    ious = inter / (gt_n + ps_n - inter + 1e-4)
    ious = ious * (gt_cls[:, None] == ps_cls[None, :]).float()
    max_ious, _ = torch.max(ious, dim=1)
    return max_ious[gt_cls >= 0]
