"""NumPy restatement of reference gapro/gen_ps_utils.py:293-482.  TEST INFRASTRUCTURE ONLY.

Every function cites the reference lines it follows (paths relative to /root/reference).
Pinned against the imported reference by tests/golden/make_golden.py.

One deliberate, documented difference: the per-superpoint *feature* mean.  The reference
sums float32 with ``torch_scatter.scatter(reduce="mean")`` (gen_ps_utils.py:357), whose
CUDA path uses float atomics in arrival order, i.e. the reference itself is not
reproducible in the last bits.  The product pools with an exact,
order-independent fixed-point sum; ``pooled_feature_mean`` below restates that integer
arithmetic so the product's pooled features can be checked bit for bit.  It agrees with
a float64 mean rounded to float32 except in rare double-rounding ties and with the
reference's float32 running sum to float32 rounding.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

MAX_NUM = 1000000  # gen_ps_utils.py:308
BOX_MARGIN = 0.005  # gen_ps_utils.py:350
IOU_OVERLAP = 0.0001  # gen_ps_utils.py:393
IOU_SKIP = 0.6  # gen_ps_utils.py:425
CONTAIN_OFFSET = 0.1  # gen_ps_utils.py:411,418


def _np(x, dtype=None):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    x = np.asarray(x)
    return x if dtype is None else x.astype(dtype)


# ------------------------------------------------------------------------------------------
# boxes
# ------------------------------------------------------------------------------------------
def assemble_boxes(coords, instance_cls, instance_box, instance_box_volume, wall_box, wall_box_volume,
                   instance_classes=18, ground_h=0.1):
    """gen_ps_utils.py:317-345.  Returns boxes f64[B,6], boxes_cls i64[B], boxes_volume f64[B].

    Instance / wall corners arrive float32 and are promoted to float64 by ``torch.cat`` with
    the float64 floor box: float32-rounded values held in float64.
    """
    coords = _np(coords, np.float64)
    min_range = coords.min(0)  # :317
    max_range = coords.max(0)  # :318
    floor_box = np.array([[min_range[0], min_range[1], min_range[2], max_range[0], max_range[1],
                           min_range[2] + ground_h]], dtype=np.float64)  # :319-325
    floor_vol = np.prod(np.maximum(floor_box[:, 3:] - floor_box[:, :3], 0.001), axis=1)  # :326
    ibox = _np(instance_box, np.float32).astype(np.float64).reshape(-1, 6)
    ivol = _np(instance_box_volume, np.float32).astype(np.float64).reshape(-1)
    icls = _np(instance_cls, np.int64).reshape(-1)
    if len(wall_box) > 0:  # :328
        wbox = _np(wall_box, np.float32).astype(np.float64).reshape(-1, 6)
        wvol = _np(wall_box_volume, np.float32).astype(np.float64).reshape(-1)
        boxes = np.concatenate([ibox, wbox, floor_box], 0)
        cls = np.concatenate([icls, np.full(len(wbox) + 1, instance_classes, dtype=np.int64)])
        vol = np.concatenate([ivol, wvol, floor_vol])
    else:  # :339-345
        boxes = np.concatenate([ibox, floor_box], 0)
        cls = np.concatenate([icls, np.full(1, instance_classes, dtype=np.int64)])
        vol = np.concatenate([ivol, floor_vol])
    return boxes, cls, vol


def batch_iou_cross(boxes1, boxes2):
    """IoU half of ``batch_giou_cross`` (gen_ps_utils.py:33-49); the GIoU half is discarded at :385."""
    b1 = boxes1[:, None, :]
    b2 = boxes2[None, :, :]
    inter = np.prod(np.maximum(np.minimum(b1[..., 3:], b2[..., 3:]) - np.maximum(b1[..., :3], b2[..., :3]), 0.0), -1)
    v1 = np.prod(np.maximum(b1[..., 3:] - b1[..., :3], 0.0), -1)
    v2 = np.prod(np.maximum(b2[..., 3:] - b2[..., :3], 0.0), -1)
    union = v1 + v2 - inter
    return inter / (union + 1e-6)


def is_box1_in_box2(box1, box2, offset=CONTAIN_OFFSET):
    """gen_ps_utils.py:75-76."""
    return bool(np.all((box1[:3] + offset) >= box2[:3]) and np.all((box1[3:] - offset) <= box2[3:]))


# ------------------------------------------------------------------------------------------
# partition: membership + superpoint pooling
# ------------------------------------------------------------------------------------------
def fixed_point_shift(fmax: float, n_points: int) -> int:
    """Exponent k of the exact pooled-feature sum: q = rint(x * 2**k) summed in int64.

    Chosen so that |q| * n_points < 2**61.  Shared spec with the HIP kernel
    (gapro_amd/csrc/partition.hip, ``stats_finalize``).
    """
    if not (fmax > 0.0) or not math.isfinite(fmax):
        return 0
    _, e = math.frexp(float(fmax))  # fmax = f * 2**e, f in [0.5, 1)
    lg = max(0, int(n_points - 1).bit_length())  # ceil(log2(n_points))
    return int(max(-1000, min(1000, 61 - e - lg)))


def pooled_feature_mean(feats_f32, spp_inv, n_spps):
    """Exact fixed-point superpoint mean of float32 features -> float32[S, D]."""
    feats_f32 = np.ascontiguousarray(feats_f32, dtype=np.float32)
    n, d = feats_f32.shape
    fmax = float(np.max(np.abs(feats_f32))) if feats_f32.size else 0.0
    k = fixed_point_shift(fmax, n)
    q = np.rint(np.ldexp(feats_f32.astype(np.float64), k)).astype(np.int64)
    sums = np.zeros((n_spps, d), dtype=np.int64)
    np.add.at(sums, spp_inv, q)
    cnt = np.bincount(spp_inv, minlength=n_spps).astype(np.float64)
    mean = np.ldexp(sums.astype(np.float64), -k) / np.maximum(cnt, 1.0)[:, None]
    return mean.astype(np.float32)


@dataclass
class Partition:
    n_spps: int
    spp_inv: np.ndarray  # i64[N] dense rank of each point's superpoint id (torch.unique inverse)
    unique_spps: np.ndarray
    boxes: np.ndarray  # f64[B,6]
    boxes_cls: np.ndarray  # i64[B]
    boxes_volume: np.ndarray  # f64[B]
    occ_count: np.ndarray  # i64[S,B] number of the superpoint's points inside each box
    point_count: np.ndarray  # i64[S]
    occ_spp: np.ndarray  # bool[S,B]
    n_bbs_per_spp: np.ndarray  # i64[S]
    feats_spp: np.ndarray  # f32[S,D]
    coords_spp: np.ndarray  # f64[S,3]


def partition(coords, feats, spp, boxes, boxes_cls, boxes_volume, thresh_spp_occu=0.8) -> Partition:
    """gen_ps_utils.py:312-315,347-363."""
    coords = _np(coords, np.float64)
    feats = _np(feats).astype(np.float32)  # :315 mask_feats.float()
    spp = _np(spp, np.int64)
    unique_spps, spp_inv = np.unique(spp, return_inverse=True)  # :312
    spp_inv = spp_inv.reshape(-1).astype(np.int64)
    S = len(unique_spps)
    B = len(boxes)
    lo = boxes[None, :, :3] - BOX_MARGIN  # :350 (float64 arithmetic on float32-rounded corners)
    hi = boxes[None, :, 3:] + BOX_MARGIN
    occ = np.all(coords[:, None, :] >= lo, -1) & np.all(coords[:, None, :] <= hi, -1)  # :79-80, :349
    occ_count = np.zeros((S, B), dtype=np.int64)
    np.add.at(occ_count, spp_inv, occ.astype(np.int64))
    point_count = np.bincount(spp_inv, minlength=S).astype(np.int64)
    # :359-362  scatter-mean of occ.float() is (integer count as f32) / (point count as f32), one
    # IEEE float32 division; the Python float threshold is compared as float32.
    occ_mean = occ_count.astype(np.float32) / np.maximum(point_count, 1).astype(np.float32)[:, None]
    occ_spp = occ_mean >= np.float32(thresh_spp_occu)
    n_bbs = occ_spp.sum(1).astype(np.int64)  # :363
    coords_sum = np.zeros((S, 3), dtype=np.float64)
    np.add.at(coords_sum, spp_inv, coords)
    coords_spp = coords_sum / np.maximum(point_count, 1)[:, None]  # :354-356 (unused downstream)
    feats_spp = pooled_feature_mean(feats, spp_inv, S)  # :357 (see module docstring)
    return Partition(S, spp_inv, unique_spps, boxes, boxes_cls, boxes_volume, occ_count, point_count, occ_spp,
                     n_bbs, feats_spp, coords_spp)


# ------------------------------------------------------------------------------------------
# static pair schedule
# ------------------------------------------------------------------------------------------
@dataclass
class Event:
    kind: str  # "contain" or "fit"
    b1: int
    b2: int
    winner: int  # contain: the box that takes the intersection; fit: -1
    intersect_inds: np.ndarray  # i64
    b1_inds: Optional[np.ndarray] = None
    b2_inds: Optional[np.ndarray] = None


def enumerate_schedule(boxes, occ_spp, n_bbs) -> List[Event]:
    """Control flow of gen_ps_utils.py:385-437 with the GP call replaced by an event.

    Which pairs are fitted, and their index sets, depend only on (boxes, occ_spp): GP outputs
    never feed back into the loop, so the schedule is static.
    """
    B = len(boxes)
    iou = batch_iou_cross(boxes, boxes)  # :385
    np.fill_diagonal(iou, 0.0)  # :386
    visited = np.zeros(B, dtype=bool)  # :388
    single = n_bbs == 1
    # inst_per_point restricted to single-box superpoints never changes inside the loop
    single_box = np.where(single, np.argmax(occ_spp, axis=1), -1)
    events: List[Event] = []
    for b1 in range(B):  # :390
        cand = np.nonzero((iou[b1] > IOU_OVERLAP) & (~visited))[0]  # :393-394 (snapshot)
        if len(cand) == 0:  # :397-399
            visited[b1] = True
            continue
        for b2 in cand:  # :401
            b2 = int(b2)
            inter = np.nonzero(occ_spp[:, b1] & occ_spp[:, b2])[0].astype(np.int64)  # :403-405
            if len(inter) == 0:  # :408-409
                continue
            if is_box1_in_box2(boxes[b1], boxes[b2]):  # :411-416
                events.append(Event("contain", b1, b2, b1, inter))
                visited[b1] = True
                break
            if is_box1_in_box2(boxes[b2], boxes[b1]):  # :418-423
                events.append(Event("contain", b1, b2, b2, inter))
                visited[b2] = True
                continue
            if iou[b1, b2] >= IOU_SKIP:  # :425-426
                continue
            t1 = np.nonzero(single_box == b1)[0].astype(np.int64)  # :428
            t2 = np.nonzero(single_box == b2)[0].astype(np.int64)  # :429
            if len(t1) == 0 or len(t2) == 0:  # :431-432
                continue
            events.append(Event("fit", b1, b2, -1, inter, t1, t2))
        visited[b1] = True  # :448
    return events


# ------------------------------------------------------------------------------------------
# merge, fallback, labels
# ------------------------------------------------------------------------------------------
FitFn = Callable[[np.ndarray, np.ndarray, np.ndarray, np.ndarray], Tuple[np.ndarray, ...]]


def merge_and_label(part: Partition, events: Sequence[Event], fit_results: Sequence[Tuple[np.ndarray, ...]],
                    n_fg_instances: int, instance_classes=18):
    """gen_ps_utils.py:365-383 (init), :411-446 (state updates), :450-482 (fallback, labels).

    ``fit_results[i]`` belongs to the i-th "fit" event:
    (pred_probs, pred_probs_new, pred_labels, pred_mu, pred_variance), each length |I|.
    """
    S = part.n_spps
    inst = np.full(S, -100, dtype=np.int32)  # :365
    determined = np.zeros(S, dtype=np.int64)  # :366
    prob = np.zeros(S, dtype=np.float32)  # :367
    mu = np.zeros(S, dtype=np.float32) - np.float32(100.0)  # :368
    var = np.zeros(S, dtype=np.float32) - np.float32(100.0)  # :369
    n_bbs = part.n_bbs_per_spp
    one = n_bbs == 1
    inst[one] = np.argmax(part.occ_spp[one], axis=1).astype(np.int32)  # :373-375
    prob[one] = 1
    determined[one] = MAX_NUM
    zero = n_bbs == 0
    inst[zero] = -1  # :381-383
    prob[zero] = 1
    determined[zero] = MAX_NUM

    fi = 0
    for ev in events:
        I = ev.intersect_inds
        if ev.kind == "contain":  # :412-414 / :419-421
            inst[I] = ev.winner
            determined[I] = MAX_NUM
            prob[I] = 1
            continue
        _, p_new, labels, p_mu, p_var = fit_results[fi]
        fi += 1
        p_new = np.asarray(p_new, dtype=np.float32)
        labels = np.asarray(labels).astype(bool)
        ow = prob[I] < p_new  # :438 strict, float32
        sel = I[ow]
        inst[sel[labels[ow]]] = ev.b2  # :440
        inst[sel[~labels[ow]]] = ev.b1  # :441
        prob[sel] = p_new[ow]  # :442
        mu[sel] = np.asarray(p_mu, dtype=np.float32)[ow]  # :443
        var[sel] = np.asarray(p_var, dtype=np.float32)[ow]  # :444
        determined[sel] = len(I)  # :446

    left = (n_bbs > 1) & (determined == 0)  # :450-452
    if left.any():
        rows = np.nonzero(left)[0]
        vol = np.where(part.occ_spp[rows], part.boxes_volume[None, :], np.inf)
        inst[rows] = np.argmin(vol, axis=1).astype(np.int32)  # :453-461 scatter_min, first minimum
        prob[rows] = 1.0  # :463

    sem_spp = np.full(S, -100, dtype=np.int32)  # :466
    inst_spp = np.full(S, -100, dtype=np.int32)  # :467
    pos = inst >= 0
    sem_spp[pos] = part.boxes_cls[inst[pos].astype(np.int64)].astype(np.int32)  # :469
    sem_spp[inst == -1] = instance_classes  # :470
    inst_spp[pos] = inst[pos]  # :472
    drop = inst_spp >= n_fg_instances
    inst_spp[drop] = -100  # :474
    sem_spp[inst_spp >= n_fg_instances] = instance_classes  # :475 (no-op after :474)
    sem = sem_spp[part.spp_inv].astype(np.int32)  # :478
    ins = inst_spp[part.spp_inv].astype(np.int32)  # :479
    prb = prob[part.spp_inv].astype(np.float32)  # :480
    spp_state = dict(inst=inst, prob=prob, mu=mu, var=var, determined=determined, sem_spp=sem_spp,
                     inst_spp=inst_spp)
    return (sem, ins, prb, mu, var), spp_state  # mu/var stay superpoint-length


def gen_pseudo_label_gaussian_process(coords_float, mask_feats, spp, instance_cls, instance_box,
                                      instance_box_volume, wall_box, wall_box_volume, instance_classes=18,
                                      dataset_name="scannetv2", ground_h=0.1, training_iter=50,
                                      thresh_spp_occu=0.8, fit_fn: Optional[FitFn] = None,
                                      return_debug=False):
    """Same signature as gen_ps_utils.py:293-307 (+ ``fit_fn`` to plug the GP)."""
    boxes, cls, vol = assemble_boxes(coords_float, instance_cls, instance_box, instance_box_volume, wall_box,
                                     wall_box_volume, instance_classes, ground_h)
    part = partition(coords_float, mask_feats, spp, boxes, cls, vol, thresh_spp_occu)
    events = enumerate_schedule(boxes, part.occ_spp, part.n_bbs_per_spp)
    if fit_fn is None:
        from .svgp_oracle import fit_gp_spp_oracle

        def fit_fn(feats_spp, b1, b2, it):
            return fit_gp_spp_oracle(feats_spp, b1, b2, it, training_iter=training_iter)

    results = [fit_fn(part.feats_spp, e.b1_inds, e.b2_inds, e.intersect_inds) for e in events if e.kind == "fit"]
    out, state = merge_and_label(part, events, results, n_fg_instances=len(_np(instance_box).reshape(-1, 6)),
                                 instance_classes=instance_classes)
    if return_debug:
        return out, dict(part=part, events=events, results=results, state=state)
    return out
