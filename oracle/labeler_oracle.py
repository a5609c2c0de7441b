"""NumPy restatement of the reference's heuristic labelers.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

* ``gen_pseudo_label``           -- reference gapro/gen_ps_utils.py:485-569 (rules "volume", "dist", "none")
* ``gen_pseudo_label_box2mask``  -- reference gapro/gen_ps_utils.py:242-290
* ``spp_align_label``            -- reference gapro/gen_ps_utils.py:99-123

Pinned: tests/test_labelers.py compares every output with tests/golden/labelers_<scene>.npz, which
tests/golden/make_golden_labelers.py produced by running the real reference functions on the golden scenes.
"""
from __future__ import annotations

import numpy as np


def _occupancy(coords, box):
    """is_within_bb_torch (:79-80) as called at :502-504 / :248-250: the margin is applied to the float32 box in
    float32, the comparison with the float64 coordinates is in float64."""
    box = np.asarray(box, dtype=np.float32)
    lo = (box[:, :3] - np.float32(0.005)).astype(np.float64)
    hi = (box[:, 3:] + np.float32(0.005)).astype(np.float64)
    c = np.asarray(coords, dtype=np.float64)[:, None, :]
    return np.all(c >= lo[None], axis=-1) & np.all(c <= hi[None], axis=-1)


def _first_argmin(values, occ):
    """scatter_min over the occupied boxes of each point: strict '<', first minimum wins."""
    v = np.where(occ, values[None, :], np.inf)
    return np.argmin(v, axis=1)


def spp_align_label(spp_ids, label, n_classes, occ_spp=None):
    n_spp = int(spp_ids.max()) + 1
    count = np.zeros((n_classes, n_spp), dtype=np.int64)
    np.add.at(count, (label, spp_ids), 1)
    if occ_spp is not None:
        count[1:, :] = count[1:, :] * occ_spp.astype(np.int64)
    return np.argmax(count, axis=0)[spp_ids]


def _finish(inst, instance_cls, instance_classes):
    n = len(inst)
    sem = np.full(n, -100, dtype=np.int32)
    out = np.full(n, -100, dtype=np.int32)
    pos = inst >= 0
    sem[pos] = np.asarray(instance_cls)[inst[pos]].astype(np.int32)
    sem[inst == -1] = instance_classes
    out[pos] = inst[pos]
    return sem, out


def gen_pseudo_label(coords, spp, instance_cls, instance_box, instance_box_volume, instance_classes=18,
                     dataset_name="scannetv2", heuristic_rule="volume"):
    coords = np.asarray(coords, dtype=np.float64)
    box = np.asarray(instance_box, dtype=np.float32)
    vol = np.asarray(instance_box_volume, dtype=np.float32)
    occ = _occupancy(coords, box)
    nbb = occ.sum(1)
    inst = np.full(len(coords), -100, dtype=np.int64)
    inst[nbb == 1] = np.argmax(occ[nbb == 1], axis=1)
    inst[nbb == 0] = -1
    multi = nbb > 1
    if heuristic_rule == "volume":
        inst[multi] = _first_argmin(vol.astype(np.float64), occ[multi])
    elif heuristic_rule == "dist":
        center = ((box[:, :3] + box[:, 3:]) / np.float32(2.0)).astype(np.float64)
        # reference quirk (:525): point_inds index the SUBSET of multi-box points, but are used to index the full
        # coordinate array -- the k-th multi-box point is measured from the coordinates of scene point k
        d = ((coords[:int(multi.sum())][:, None, :] - center[None]) ** 2).sum(-1)
        inst[multi] = np.argmin(np.where(occ[multi], d, np.inf), axis=1)
    elif heuristic_rule == "none":
        inst[multi] = -2
    else:
        raise ValueError(heuristic_rule)
    if dataset_name == "scannetv2":
        _, ids = np.unique(np.asarray(spp), return_inverse=True)
        n_spp = int(ids.max()) + 1
        cnt = np.bincount(ids, minlength=n_spp).astype(np.float32)
        occ_sum = np.zeros((occ.shape[1], n_spp), dtype=np.float32)
        np.add.at(occ_sum, (np.nonzero(occ)[1], ids[np.nonzero(occ)[0]]), np.float32(1.0))
        occ_spp = (occ_sum / np.maximum(cnt, 1)[None, :]) >= np.float32(0.7)
        lab = np.where(inst >= 0, inst + 1, 0)
        lab = spp_align_label(ids, lab, occ.shape[1] + 1, occ_spp)
        inst = np.where(lab > 0, lab - 1, -1)
    sem, out = _finish(inst, instance_cls, instance_classes)
    sem[inst == -2] = -100
    return sem, out


def gen_pseudo_label_box2mask(coords, spp, instance_cls, instance_box, instance_box_volume, instance_classes=18,
                              dataset_name="scannetv2"):
    coords = np.asarray(coords, dtype=np.float64)
    box = np.asarray(instance_box, dtype=np.float32)
    vol = np.asarray(instance_box_volume, dtype=np.float32)
    occ = _occupancy(coords, box)
    nbb = occ.sum(1)
    inst = np.full(len(coords), -100, dtype=np.int64)
    multi = nbb > 1
    inst[multi] = _first_argmin(vol.astype(np.float64), occ[multi])
    inst[nbb == 1] = np.argmax(occ[nbb == 1], axis=1)
    inst[nbb == 0] = -1
    if dataset_name == "scannetv2":
        _, ids = np.unique(np.asarray(spp), return_inverse=True)
        lab = np.where(inst >= 0, inst + 1, 0)
        lab = spp_align_label(ids, lab, occ.shape[1] + 1)
        inst = np.where(lab > 0, lab - 1, -1)
    return _finish(inst, instance_cls, instance_classes)
