"""Pseudo-label quality metrics on the GPU: mirror of reference gapro/eval_ps_labels.py:35-42,100-172.

SURVEY.md section 8(f) row 1 ("next"): not part of the generator's hot path; wired to ``gen_ps --eval_pslabel``.
Same call signatures as the reference (which runs them on ``.cuda()`` tensors); the work is done by the HIP
kernels of gapro_amd/csrc/labels.hip behind ``gapro_eval_miou`` / ``gapro_eval_sem_confusion``: one histogram
pass over the points instead of two [I, N] one-hot matrices and their product.  There is no CPU path: inputs
are moved to the device, and without a HIP device the call raises.
"""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import Context, EvalHeader


def _dev_long(t, device):
    t = t if isinstance(t, torch.Tensor) else torch.as_tensor(t)
    return t.to(device=device, dtype=torch.int64).contiguous()


def _device_of(*tensors):
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return t.device
    if not torch.cuda.is_available():
        raise RuntimeError("gapro_amd.eval_ps_labels needs a HIP device; there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def get_miou_scene(semantic_label, instance_label, ps_semantic_label, ps_instance_label):
    """Per GT instance: max IoU over pseudo instances of the same class (eval_ps_labels.py:100-147).

    IoU = inter / (|gt| + |ps| - inter + 1e-4), float32 as in ``cal_iou`` (:35-42).  Returns a float32 device
    tensor with one entry per non-empty GT instance id, in id order."""
    dev = _device_of(semantic_label, instance_label, ps_semantic_label, ps_instance_label)
    sem, ins = _dev_long(semantic_label, dev), _dev_long(instance_label, dev)
    ps_sem, ps_ins = _dev_long(ps_semantic_label, dev), _dev_long(ps_instance_label, dev)
    n = int(ins.numel())
    if n == 0:
        return torch.zeros(0, device=dev)
    ctx = Context.get(dev.index or 0)
    lib = ctx.lib
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        cap_gt, cap_ps = 256, 256
        while True:
            ws_bytes = int(lib.gapro_eval_workspace_bytes(cap_gt, cap_ps))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            max_iou = torch.empty(cap_gt, dtype=torch.float32, device=dev)
            gt_cls = torch.empty(cap_gt, dtype=torch.float32, device=dev)
            d_hdr = torch.empty(C.sizeof(EvalHeader), dtype=torch.uint8, device=dev)
            h_hdr = torch.empty(C.sizeof(EvalHeader), dtype=torch.uint8, pin_memory=True)
            ctx.check(lib.gapro_eval_miou(ctx.handle, stream, n, sem.data_ptr(), ins.data_ptr(), ps_sem.data_ptr(),
                                          ps_ins.data_ptr(), cap_gt, cap_ps, ws.data_ptr(), ws_bytes, max_iou.data_ptr(),
                                          gt_cls.data_ptr(), d_hdr.data_ptr(), h_hdr.data_ptr()))
            torch.cuda.current_stream(dev).synchronize()
            hdr = EvalHeader.from_buffer_copy(h_hdr.numpy().tobytes())
            if hdr.status == 0:
                break
            # an id beyond the table: size the tables from the data and retry
            cap_gt = max(cap_gt, int(ins.max()) + 1)
            cap_ps = max(cap_ps, int(ps_ins.max()) + 1)
    n_gt = int(hdr.n_gt)
    if n_gt <= 0:
        return torch.zeros(0, device=dev)
    return max_iou[:n_gt][gt_cls[:n_gt] >= 0]


def get_scene_sem_conf(semantic_label, ps_semantic_label, num_classes=19):
    """Semantic confusion matrix i64[C, C] (eval_ps_labels.py:150-172); the inputs are not modified."""
    dev = _device_of(semantic_label, ps_semantic_label)
    sem, ps_sem = _dev_long(semantic_label, dev), _dev_long(ps_semantic_label, dev)
    conf = torch.zeros((num_classes, num_classes), dtype=torch.int64, device=dev)
    n = int(sem.numel())
    if n == 0:
        return conf
    ctx = Context.get(dev.index or 0)
    with torch.cuda.device(dev):
        ctx.check(ctx.lib.gapro_eval_sem_confusion(ctx.handle, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), n,
                                                   sem.data_ptr(), ps_sem.data_ptr(), int(num_classes), conf.data_ptr()))
    return conf
