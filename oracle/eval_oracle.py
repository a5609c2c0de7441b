"""CPU restatement of the reference's label-side helpers.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

* ``get_miou_scene``      -- reference gapro/eval_ps_labels.py:35-42,100-147.  The reference builds two one-hot
  matrices [I, N] and multiplies them; the same intersection counts come from one bincount over (gt, pseudo) id
  pairs (integer arithmetic, identical values), the IoU arithmetic is float32 in the reference's order.
* ``get_scene_sem_conf``  -- reference gapro/eval_ps_labels.py:150-172.
* ``get_instance_info``   -- reference gapro/gen_ps_utils.py:195-239, literal per-instance loop.

Pinned: get_miou_scene against the IoUs the imported reference produced on the golden scenes
(tests/golden/*.npz ``ref_ious``, tests/test_host_golden.py); get_instance_info against the reference's own
boxes in the same fixtures through gapro_amd.gen_ps_utils.getInstanceInfo (tests/test_host_golden.py).
"""
from __future__ import annotations

import numpy as np
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


def get_scene_sem_conf(semantic_label, ps_semantic_label, num_classes=19):
    """eval_ps_labels.py:150-172: confusion counts conf[gt, pseudo] over the points whose GT class is not -100; a
    pseudo label of -100 is replaced by a class that is certainly wrong (gt + 1, or gt - 1 for gt >= 18)."""
    keep = semantic_label != -100
    gt = semantic_label[keep].long()
    ps = ps_semantic_label[keep].long()
    wrong = torch.where(gt < 18, gt + 1, gt - 1)
    ps = torch.where(ps == -100, wrong, ps)
    flat = gt * num_classes + ps
    return torch.bincount(flat, minlength=num_classes * num_classes).view(num_classes, num_classes)


def get_instance_info(xyz, instance_label, semantic_label, scannet=True):
    """gen_ps_utils.py:195-239: for every non-empty instance id in ascending order the box [min | max], the class
    of the instance's first point (ScanNet: minus 2 unless -100), the volume prod(max(extent, 0)), and per point
    the float32 offsets to its instance's corners (-100 without instance).
    Returns (instance_num, cls, box[B,6], volume[B], corners f32[N,6]) or None."""
    pts = np.asarray(xyz, dtype=np.float64)
    inst = np.asarray(instance_label)
    sem = np.asarray(semantic_label)
    n_ids = int(inst.max()) + 1
    corners = np.full((len(pts), 6), -100.0, dtype=np.float32)
    rows = []
    for ident in range(n_ids):
        members = np.flatnonzero(inst == ident)
        if members.size == 0:
            continue
        p = pts[members]
        lo, hi = p.min(axis=0), p.max(axis=0)
        corners[members] = np.concatenate([lo - p, hi - p], axis=1)
        extent = np.clip(hi - lo, 0.0, None)
        rows.append((sem[members[0]], np.concatenate([lo, hi]), np.prod(extent)))
    if not rows:
        return None
    cls = np.array([r[0] for r in rows], dtype=np.float64)
    if scannet:
        cls[cls != -100] -= 2
    return n_ids, cls, np.stack([r[1] for r in rows]), np.array([r[2] for r in rows]), corners
