"""Same-name mirror of reference gapro/gen_ps_utils.py for the live functions of the path.

``gen_pseudo_label_gaussian_process`` keeps the reference signature (gen_ps_utils.py:293-307) and
return contract (:482): (sem i32[N], inst i32[N], prob f32[N], mu f32[S], var f32[S]).
"""
from __future__ import annotations

import numpy as np
import torch

from .pipeline import Pipeline, make_job

_PIPELINES = {}


def _pipeline(device, training_iter, **kw) -> Pipeline:
    key = (str(device), int(training_iter), tuple(sorted(kw.items())))
    p = _PIPELINES.get(key)
    if p is None:
        p = _PIPELINES[key] = Pipeline(device=device, training_iter=training_iter, **kw)
    return p


def _pick_device(x, device):
    if device is not None:
        return torch.device(device)
    if isinstance(x, torch.Tensor) and x.is_cuda:
        return x.device
    return torch.device("cuda:0")


def gen_pseudo_label_gaussian_process(
    coords_float,
    mask_feats,
    spp,
    instance_cls,
    instance_box,
    instance_box_volume,
    wall_box,
    wall_box_volume,
    instance_classes=18,
    dataset_name="scannetv2",
    ground_h=0.1,
    training_iter=50,
    thresh_spp_occu=0.8,
    *,
    broadcast_mu_var=False,
    device=None,
    init_mean_std=0.0,
    seed=0,
    eval_stale_chol=False,
):
    """Reference gen_ps_utils.py:293-482 on the GPU.  ``dataset_name`` is accepted and unused, as in
    the reference.  Extra keyword-only options (defaults reproduce the reference contract):

    broadcast_mu_var  return mu/var broadcast to point length (what the ISBNet/SPFormer loaders index,
                      SURVEY Q2) instead of the reference generator's superpoint length.
    init_mean_std     std of the random initial variational mean (gpytorch uses 1e-3 with an unseeded
                      RNG; 0 = deterministic zeros), ``seed`` seeds it.
    eval_stale_chol   predict with the Cholesky factor of the last training step.
    """
    dev = _pick_device(coords_float, device)
    was_cpu = not (isinstance(coords_float, torch.Tensor) and coords_float.is_cuda)
    job = make_job(coords_float, mask_feats, spp, instance_cls, instance_box, instance_box_volume, wall_box,
                   wall_box_volume, instance_classes, ground_h, thresh_spp_occu, device=dev)
    pipe = _pipeline(dev, training_iter, init_mean_std=init_mean_std, seed=seed, eval_stale_chol=eval_stale_chol)
    sem, ins, prob, mu, var = pipe.run([job])[0]
    if broadcast_mu_var:
        idx = job.spp_inv.long()
        mu, var = mu[idx], var[idx]
    outs = (sem, ins, prob, mu, var)
    if was_cpu:
        outs = tuple(o.cpu() for o in outs)
    return outs


def gen_pseudo_label_gaussian_process_batch(scenes, training_iter=50, device=None, **pipe_kw):
    """Several scenes through one pipeline pass (all their GP fits share one launch).

    ``scenes`` is a list of dicts holding the positional arguments of
    ``gen_pseudo_label_gaussian_process`` by name.  Returns a list of 5-tuples of device tensors.
    """
    dev = torch.device(device if device is not None else "cuda:0")
    jobs = [make_job(s["coords_float"], s["mask_feats"], s["spp"], s["instance_cls"], s["instance_box"],
                     s["instance_box_volume"], s["wall_box"], s["wall_box_volume"],
                     s.get("instance_classes", 18), s.get("ground_h", 0.1), s.get("thresh_spp_occu", 0.8),
                     device=dev) for s in scenes]
    return _pipeline(dev, training_iter, **pipe_kw).run(jobs)


def getInstanceInfo(xyz, instance_label, semantic_label, dataset_name="scannetv2"):
    """Reference gen_ps_utils.py:195-239 (host NumPy, as in the reference).

    Boxes are indexed by rank among the non-empty instance ids; class = semantic label of the
    instance's first point (ScanNet: shifted by -2 unless -100); volume = prod(clip(max-min, 0)).
    Returns None when there is no instance.  One sort instead of an O(N*I) np.where loop.
    """
    xyz = np.asarray(xyz)
    instance_label = np.asarray(instance_label)
    semantic_label = np.asarray(semantic_label)
    instance_num = int(instance_label.max()) + 1
    corners_label = np.ones((xyz.shape[0], 3 * 2), dtype=np.float32) * -100.0
    valid = np.nonzero((instance_label >= 0) & (instance_label < instance_num))[0]
    if len(valid) == 0:
        return None
    lab = instance_label[valid].astype(np.int64)
    order = np.argsort(lab, kind="stable")
    sidx = valid[order]
    slab = lab[order]
    starts = np.nonzero(np.r_[True, slab[1:] != slab[:-1]])[0]
    pts = xyz[sidx]
    mins = np.minimum.reduceat(pts, starts, axis=0)
    maxs = np.maximum.reduceat(pts, starts, axis=0)
    seg = np.repeat(np.arange(len(starts)), np.diff(np.r_[starts, len(slab)]))
    corners_label[sidx, :3] = mins[seg] - pts
    corners_label[sidx, 3:] = maxs[seg] - pts
    instance_cls = np.array(semantic_label[sidx[starts]])  # first point of each instance (stable sort)
    instance_box = np.concatenate([mins, maxs], axis=1)
    instance_box_volume = np.prod(np.clip(maxs - mins, a_min=0.0, a_max=None), axis=1)
    if dataset_name == "scannetv2":
        instance_cls[instance_cls != -100] -= 2
    return instance_num, instance_cls, instance_box, instance_box_volume, corners_label


def batch_giou_cross(boxes1, boxes2):
    """Reference gen_ps_utils.py:33-61 (torch, any device): returns (iou, giou)."""
    boxes1 = boxes1[:, None, :]
    boxes2 = boxes2[None, :, :]
    intersection = torch.prod(
        torch.clamp(torch.min(boxes1[..., 3:], boxes2[..., 3:]) - torch.max(boxes1[..., :3], boxes2[..., :3]),
                    min=0.0), -1)
    v1 = torch.prod(torch.clamp(boxes1[..., 3:] - boxes1[..., :3], min=0.0), -1)
    v2 = torch.prod(torch.clamp(boxes2[..., 3:] - boxes2[..., :3], min=0.0), -1)
    union = v1 + v2 - intersection
    iou = intersection / (union + 1e-6)
    bound = torch.prod(
        torch.clamp(torch.max(boxes1[..., 3:], boxes2[..., 3:]) - torch.min(boxes1[..., :3], boxes2[..., :3]),
                    min=0.0), -1)
    giou = iou - (bound - union) / (bound + 1e-6)
    return iou, giou


def is_box1_in_box2(box1, box2, offset=0.05):
    """Reference gen_ps_utils.py:75-76."""
    return torch.all((box1[:3] + offset) >= box2[:3]) & torch.all((box1[3:] - offset) <= box2[3:])
