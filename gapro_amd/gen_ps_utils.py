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


def getInstanceInfo_native(xyz, instance_label, semantic_label, dataset_name="scannetv2"):
    """getInstanceInfo (gen_ps_utils.py:195-239) in one native host pass (gapro_scene_instance_boxes), without the
    corner labels: (instance_num, instance_cls f64[B], instance_box f64[B,6], instance_box_volume f64[B], None), or
    None without instances -- the same values as getInstanceInfo.  What the gen_ps driver's loader threads use."""
    import ctypes as C

    from . import _lib

    lib = _lib.load()
    xyz = np.ascontiguousarray(np.asarray(xyz, dtype=np.float64))
    inst = np.ascontiguousarray(np.asarray(instance_label, dtype=np.float64)).reshape(-1)
    sem = np.ascontiguousarray(np.asarray(semantic_label, dtype=np.float64)).reshape(-1)
    n = int(xyz.shape[0])
    cap = 256
    nb, inum = C.c_int32(0), C.c_int32(0)
    while True:
        box = np.empty((cap, 6), dtype=np.float64)
        cls = np.empty(cap, dtype=np.float64)
        vol = np.empty(cap, dtype=np.float64)
        rc = lib.gapro_scene_instance_boxes(xyz.ctypes.data, inst.ctypes.data, sem.ctypes.data, n,
                                            1 if dataset_name == "scannetv2" else 0, cap, box.ctypes.data,
                                            cls.ctypes.data, vol.ctypes.data, C.byref(nb), C.byref(inum))
        if rc == 0:
            break
        if inum.value > cap:
            cap = int(inum.value)
            continue
        raise _lib.GaproError(rc, "gapro_scene_instance_boxes")
    if nb.value == 0:
        return None
    k = int(nb.value)
    return int(inum.value), cls[:k].copy(), box[:k].copy(), vol[:k].copy(), None


def getInstanceInfo_device(xyz, instance_label, semantic_label, dataset_name="scannetv2", device=None,
                           return_corners=False):
    """getInstanceInfo (gen_ps_utils.py:195-239) on the GPU: one pass over the points behind
    ``gapro_instance_info`` (gapro_amd/csrc/labels.hip) instead of an O(N * I) ``np.where`` loop.

    Inputs: arrays or tensors (moved to the device as float64, the dtype the ScanNet .pth files hold).  Returns
    the reference's tuple ``(instance_num, instance_cls f64[B], instance_box f64[B,6], instance_box_volume
    f64[B], corners_label f32[N,6] or None)`` as host NumPy arrays (they are tiny), or None without instances.
    SURVEY.md section 8(f) row 2."""
    import ctypes as C

    from ._lib import Context, InstanceHeader

    if device is None:
        device = xyz.device if isinstance(xyz, torch.Tensor) and xyz.is_cuda else torch.device("cuda", 0)
    device = torch.device(device)
    if not torch.cuda.is_available():
        raise RuntimeError("getInstanceInfo_device needs a HIP device; getInstanceInfo is the host form")

    def dev64(a):
        a = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        return a.to(device=device, dtype=torch.float64).contiguous()

    coords, inst, sem = dev64(xyz), dev64(instance_label).view(-1), dev64(semantic_label).view(-1)
    n = int(coords.shape[0])
    if n == 0:
        return None
    ctx = Context.get(device.index or 0)
    lib = ctx.lib
    with torch.cuda.device(device):
        stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        cap = 1024
        while True:
            ws_bytes = int(lib.gapro_instance_info_workspace_bytes(cap))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
            box = torch.empty((cap, 6), dtype=torch.float64, device=device)
            cls = torch.empty(cap, dtype=torch.float64, device=device)
            vol = torch.empty(cap, dtype=torch.float64, device=device)
            corners = torch.empty((n, 6), dtype=torch.float32, device=device) if return_corners else None
            d_hdr = torch.empty(C.sizeof(InstanceHeader), dtype=torch.uint8, device=device)
            h_hdr = torch.empty(C.sizeof(InstanceHeader), dtype=torch.uint8, pin_memory=True)
            ctx.check(lib.gapro_instance_info(
                ctx.handle, stream, n, coords.data_ptr(), inst.data_ptr(), sem.data_ptr(), cap,
                1 if dataset_name == "scannetv2" else 0, ws.data_ptr(), ws_bytes, box.data_ptr(), cls.data_ptr(),
                vol.data_ptr(), corners.data_ptr() if corners is not None else None, d_hdr.data_ptr(), h_hdr.data_ptr()))
            torch.cuda.current_stream(device).synchronize()
            hdr = InstanceHeader.from_buffer_copy(h_hdr.numpy().tobytes())
            if hdr.status == 0:
                break
            cap = max(2 * cap, int(inst.max()) + 1)  # an id beyond the table: size it from the data and retry
    nb = int(hdr.n_boxes)
    if nb == 0:
        return None
    return (int(hdr.instance_num), cls[:nb].cpu().numpy(), box[:nb].cpu().numpy(), vol[:nb].cpu().numpy(),
            corners.cpu().numpy() if corners is not None else None)


def _heuristic_labels(coords_float, spp, instance_cls, instance_box, instance_box_volume, instance_classes,
                      dataset_name, rule):
    import ctypes as C

    from ._lib import Context, SceneHeader

    if not torch.cuda.is_available():
        raise RuntimeError("the heuristic labelers run on a HIP device; there is no CPU fallback")

    def dev(a, dtype):
        a = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        return a.to(device=device, dtype=dtype).contiguous()

    device = coords_float.device if isinstance(coords_float, torch.Tensor) and coords_float.is_cuda \
        else torch.device("cuda", torch.cuda.current_device())
    coords = dev(coords_float, torch.float64)
    n = int(coords.shape[0])
    box, vol, cls = dev(instance_box, torch.float32), dev(instance_box_volume, torch.float32), dev(instance_cls, torch.int64)
    B = int(box.shape[0])
    ctx = Context.get(device.index or 0)
    lib = ctx.lib
    align = dataset_name == "scannetv2"
    with torch.cuda.device(device):
        stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        spp_inv, n_spps = None, 0
        if align:  # torch.unique(spp, return_inverse=True) (:537 / :275): the partition's dense-rank kernels
            spp_d = dev(spp, torch.int64)
            cap = max(4 * n, 1 << 20)
            nbytes = int(lib.gapro_partition_prepare_workspace_bytes(n, cap))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            spp_inv = torch.empty(n, dtype=torch.int32, device=device)
            dummy = torch.zeros(n, dtype=torch.float32, device=device)
            hdr = SceneHeader()
            ctx.check(lib.gapro_partition_prepare(ctx.handle, stream, n, 1, coords.data_ptr(), dummy.data_ptr(),
                                                  spp_d.data_ptr(), cap, ws.data_ptr(), nbytes, spp_inv.data_ptr(),
                                                  C.byref(hdr)))
            n_spps = int(hdr.n_spps)
        ws_bytes = int(lib.gapro_label_heuristic_workspace_bytes(n, max(n_spps, 1), B))
        lws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
        sem = torch.empty(n, dtype=torch.int32, device=device)
        ins = torch.empty(n, dtype=torch.int32, device=device)
        ctx.check(lib.gapro_label_heuristic(ctx.handle, stream, n, coords.data_ptr(),
                                            spp_inv.data_ptr() if spp_inv is not None else None, n_spps, B,
                                            box.data_ptr(), vol.data_ptr(), cls.data_ptr(), rule, 1 if align else 0,
                                            int(instance_classes), lws.data_ptr(), ws_bytes, sem.data_ptr(),
                                            ins.data_ptr()))
        torch.cuda.current_stream(device).synchronize()
    return sem, ins


def gen_pseudo_label(coords_float, spp, instance_cls, instance_box, instance_box_volume, instance_classes=18,
                     dataset_name="scannetv2", heuristic_rule="volume"):
    """Reference gen_ps_utils.py:485-569 (the box-heuristic labeler: a point inside several boxes goes to the
    smallest box / the nearest box centre / nowhere), on the GPU behind ``gapro_label_heuristic``.  Returns
    ``(ps_semantic_label i32[N], ps_instance_label i32[N])`` device tensors.  SURVEY.md 8(f) row 4."""
    rules = {"volume": 0, "dist": 1, "none": 2}
    if heuristic_rule not in rules:
        raise Exception  # as the reference does (:534)
    return _heuristic_labels(coords_float, spp, instance_cls, instance_box, instance_box_volume, instance_classes,
                             dataset_name, rules[heuristic_rule])


def gen_pseudo_label_box2mask(coords_float, spp, instance_cls, instance_box, instance_box_volume, instance_classes=18,
                              dataset_name="scannetv2"):
    """Reference gen_ps_utils.py:242-290 (Box2Mask-style labeler: smallest box, plain superpoint majority vote)."""
    return _heuristic_labels(coords_float, spp, instance_cls, instance_box, instance_box_volume, instance_classes,
                             dataset_name, 3)


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
