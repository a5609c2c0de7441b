"""Consumer-side label ops on the GPU: the first thing ISBNet / SPFormer do with
the pseudo labels this package generates.  Same semantics as the reference lines cited per function; the work is
done by gapro_amd/csrc/consumer.hip behind the C ABI; the losses are ``torch.autograd.Function``s whose backward
uses the gradients the forward launch already produced.  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import Context


def _ctx_stream(t):
    if not t.is_cuda:
        raise RuntimeError("gapro_amd.consumer_ops works on HIP device tensors; there is no CPU fallback")
    ctx = Context.get(t.device.index or 0)
    return ctx, C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def pool_labels_to_superpoints(prob_labels, mu_labels, var_labels, spps, n_out=None):
    """``custom_scatter_mean`` (ISBNet/isbnet/model/model_utils.py:600-613) of the three label channels in one
    pass (isbnet.py:387-389): float32 means per superpoint index, output length ``max(spps) + 1`` unless given."""
    ctx, stream = _ctx_stream(prob_labels)
    dev = prob_labels.device
    idx = spps.to(device=dev, dtype=torch.int64).contiguous()
    chans = [t.to(device=dev, dtype=torch.float32).contiguous() for t in (prob_labels, mu_labels, var_labels)]
    n = int(idx.numel())
    if n_out is None:
        n_out = int(idx.max()) + 1 if n else 0
    outs = [torch.zeros(n_out, dtype=torch.float32, device=dev) for _ in range(3)]
    if n == 0 or n_out == 0:
        return tuple(outs)
    sums = torch.empty(3 * n_out, dtype=torch.float64, device=dev)
    counts = torch.empty(n_out, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        ctx.check(ctx.lib.gapro_label_pool_mean(ctx.handle, stream, n, n_out, idx.data_ptr(), chans[0].data_ptr(),
                                                chans[1].data_ptr(), chans[2].data_ptr(), sums.data_ptr(),
                                                counts.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(),
                                                outs[2].data_ptr()))
    return tuple(outs)


class _WeightedBCE(torch.autograd.Function):
    @staticmethod
    def forward(fctx, logits, targets, weights):
        ctx, stream = _ctx_stream(logits)
        x = logits.to(torch.float32).contiguous()
        y = targets.to(device=x.device, dtype=torch.float32).contiguous()
        w = weights.to(device=x.device, dtype=torch.float32).contiguous()
        G, P = x.shape
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x) if logits.requires_grad else None
        acc = torch.empty(2, dtype=torch.float64, device=x.device)
        with torch.cuda.device(x.device):
            ctx.check(ctx.lib.gapro_weighted_bce_with_logits(ctx.handle, stream, G, P, x.data_ptr(), y.data_ptr(),
                                                             w.data_ptr(), 1.0, acc.data_ptr(), loss.data_ptr(),
                                                             grad.data_ptr() if grad is not None else None))
        fctx.grad = grad
        fctx.in_dtype = logits.dtype
        return loss

    @staticmethod
    def backward(fctx, grad_out):
        g = fctx.grad * grad_out if fctx.grad is not None else None
        return (g.to(fctx.in_dtype) if g is not None else None), None, None


def prob_weighted_bce_with_logits(mask_logit_pred, inst_label, prob_labels):
    """``(bce * prob_labels).sum() / prob_labels.sum() / (num_gt + 1e-6)`` with ``bce =
    F.binary_cross_entropy_with_logits(mask_logit_pred, inst_label, reduction="none")`` over [num_gt, n_points]
    (ISBNet/isbnet/model/criterion.py:287-288)."""
    return _WeightedBCE.apply(mask_logit_pred, inst_label, prob_labels)


class _KLToGP(torch.autograd.Function):
    @staticmethod
    def forward(fctx, mu_pred, logvar_pred, mu_labels, var_labels, weight, epsilon):
        ctx, stream = _ctx_stream(mu_pred)
        dev = mu_pred.device
        mp = mu_pred.to(torch.float32).contiguous().view(-1)
        lp = logvar_pred.to(torch.float32).contiguous().view(-1)
        ml = mu_labels.to(device=dev, dtype=torch.float32).contiguous().view(-1)
        vl = var_labels.to(device=dev, dtype=torch.float32).contiguous().view(-1)
        n = int(mp.numel())
        loss = torch.zeros((), dtype=torch.float32, device=dev)
        need = mu_pred.requires_grad or logvar_pred.requires_grad
        gm = torch.empty_like(mp) if need else None
        gl = torch.empty_like(lp) if need else None
        if n:
            acc = torch.empty(4, dtype=torch.float64, device=dev)
            with torch.cuda.device(dev):
                ctx.check(ctx.lib.gapro_kl_gp_loss(ctx.handle, stream, n, ml.data_ptr(), vl.data_ptr(), mp.data_ptr(),
                                                   lp.data_ptr(), float(epsilon), float(weight), 1.0, acc.data_ptr(),
                                                   loss.data_ptr(), gm.data_ptr() if need else None,
                                                   gl.data_ptr() if need else None))
        fctx.grads = (gm, gl)
        fctx.shapes = (mu_pred.shape, logvar_pred.shape, mu_pred.dtype, logvar_pred.dtype)
        return loss

    @staticmethod
    def backward(fctx, grad_out):
        gm, gl = fctx.grads
        s_mu, s_lv, d_mu, d_lv = fctx.shapes
        if gm is None:
            return None, None, None, None, None, None
        return ((gm * grad_out).view(s_mu).to(d_mu), (gl * grad_out).view(s_lv).to(d_lv), None, None, None, None)


def kl_to_gp_loss(mu_pred, logvar_pred, mu_labels, var_labels, weight=1.0, epsilon=1e-4):
    """The KL-to-GP auxiliary loss of ISBNet/isbnet/model/criterion.py:435-463 (``weight`` = loss_weight["kl_loss"])."""
    return _KLToGP.apply(mu_pred, logvar_pred, mu_labels, var_labels, weight, epsilon)
