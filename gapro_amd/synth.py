"""Seeded, ScanNet-shaped synthetic scenes.

No ScanNet/S3DIS data can be shipped (licence-gated, no network), so every test,
fixture and benchmark in this repo runs on scenes made here.  A scene mimics what
the reference driver loads per scan (reference gapro/gen_ps.py:43-69):

* ``xyz``   f64[N,3]  raw (un-aligned, mean-centred) vertex coordinates
* ``rgb``   f64[N,3]  colours in [-1, 1]
* ``sem``   f64[N]    semantic label (0 wall, 1 floor, 2..19 objects)
* ``inst``  f64[N]    instance label (-100 for wall/floor)
* ``spp``   i64[N]    superpoint id per vertex (non-contiguous, shuffled)
* ``axis_align`` f64[4,4]  the scan's axisAlignment matrix
* ``quads`` dict or None   ScanNet-Planes style wall quads (raw mesh frame)

Geometry: a box room (floor + 4 walls) and K box-shaped objects standing on the
floor, surface-sampled.  A fraction of the objects is placed so that their
axis-aligned boxes overlap a neighbour (IoU 0.01-0.5), lie inside a neighbour
(containment rule, reference gen_ps_utils.py:411-423) or nearly coincide with
it (IoU >= 0.6 skip rule, gen_ps_utils.py:425), so every branch of the pair
scheduler is exercised.  Superpoints are surface patches: ~50 points on
objects, ~400 on the large planar structures (graph-based segmentation merges
planar regions into big superpoints).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional

import numpy as np


@dataclass
class Scene:
    scan_name: str
    xyz: np.ndarray
    rgb: np.ndarray
    sem: np.ndarray
    inst: np.ndarray
    spp: np.ndarray
    axis_align: np.ndarray
    quads: Optional[dict] = None
    meta: dict = field(default_factory=dict)

    @property
    def n_points(self) -> int:
        return int(self.xyz.shape[0])

    def aligned_xyz(self) -> np.ndarray:
        """xyz after the axis alignment of reference gen_ps.py:65-69 (float64)."""
        pts = np.ones((self.xyz.shape[0], 4))
        pts[:, 0:3] = self.xyz[:, 0:3]
        pts = np.dot(pts, self.axis_align.transpose())
        return pts[:, :3]

    def default_feats(self) -> np.ndarray:
        """The non-deepfeat features of reference gen_ps.py:55 (un-aligned xyz + rgb)."""
        return np.concatenate([self.xyz, self.rgb], axis=-1)


def _sample_rect(rng, n, origin, eu, ev, lu, lv):
    """n uniform points on the rectangle origin + a*eu + b*ev, a in [0,lu], b in [0,lv]."""
    a = rng.random(n) * lu
    b = rng.random(n) * lv
    pts = origin[None, :] + a[:, None] * eu[None, :] + b[:, None] * ev[None, :]
    return pts, a, b


def _patch_ids(a, b, lu, lv, pts_per_patch):
    """Grid the (a,b) parametrisation so that a cell holds ~pts_per_patch points."""
    n = len(a)
    if n == 0:
        return np.zeros(0, dtype=np.int64), 0
    n_cells = max(1, int(round(n / pts_per_patch)))
    aspect = max(lu, 1e-6) / max(lv, 1e-6)
    nu = max(1, int(round(math.sqrt(n_cells * aspect))))
    nv = max(1, int(round(n_cells / nu)))
    iu = np.minimum((a / max(lu, 1e-9) * nu).astype(np.int64), nu - 1)
    iv = np.minimum((b / max(lv, 1e-9) * nv).astype(np.int64), nv - 1)
    return iu * nv + iv, nu * nv


def make_scene(
    seed: int = 0,
    n_points: int = 150_000,
    n_objects: Optional[int] = None,
    with_walls_json: Optional[bool] = None,
    obj_patch: int = 50,
    plane_patch: int = 400,
    feat_noise: float = 0.05,
    scan_name: Optional[str] = None,
    p_kinds=(0.30, 0.05, 0.02),
    p_wall: float = 0.0,
    mesh_order: bool = False,
) -> Scene:
    """mesh_order: vertices of a face in Z-order (consecutive vertices are spatial neighbours and mostly share a
    superpoint, as in a reconstructed mesh) instead of the order they were sampled in (random within the face); the set
    of points, their superpoints and every output of the generator are the same, only the vertex order differs."""
    rng = np.random.default_rng(seed)
    ex, ey, ez = rng.uniform(4.0, 8.0), rng.uniform(3.0, 7.0), rng.uniform(2.4, 3.0)
    K = int(rng.integers(10, 41)) if n_objects is None else int(n_objects)
    if with_walls_json is None:
        with_walls_json = bool(rng.random() < 0.7)

    # ---- object boxes (aligned frame, room spans [0,ex]x[0,ey]x[0,ez]) ----------------
    centers = np.zeros((K, 2))
    sizes = np.zeros((K, 3))
    z0 = np.zeros(K)
    yaw = np.zeros(K)
    kind = np.zeros(K, dtype=np.int64)  # 0 free, 1 overlapping, 2 contained, 3 near-duplicate
    for k in range(K):
        sz = np.array([rng.uniform(0.35, 1.4), rng.uniform(0.35, 1.4), rng.uniform(0.3, 1.2)])
        r = rng.random()
        p_ov, p_in, p_dup = p_kinds
        if k > 0 and r < p_ov:
            j = int(rng.integers(0, k))
            # shift by a fraction of the neighbour's extent: AABB IoU roughly 0.01-0.5
            frac = rng.uniform(0.35, 0.9, size=2) * rng.choice([-1.0, 1.0], size=2)
            c = centers[j] + frac * 0.5 * (sizes[j, :2] + sz[:2])
            kind[k] = 1
            zb = 0.0
        elif k > 0 and r < p_ov + p_in:
            j = int(rng.integers(0, k))
            sz = sizes[j] * rng.uniform(0.25, 0.45, size=3)
            c = centers[j] + rng.uniform(-0.1, 0.1, size=2) * sizes[j, :2]
            zb = z0[j] + sizes[j, 2] * rng.uniform(0.2, 0.4)  # sits inside the neighbour's box
            kind[k] = 2
        elif k > 0 and r < p_ov + p_in + p_dup and sizes[:k, :2].max() >= 0.8:
            # near-duplicate: same size, shifted 0.12-0.16 m along its longer horizontal axis, so
            # that IoU >= 0.6 but neither box contains the other within the 0.1 m slack
            j = int(np.argmax(sizes[:k, :2].max(1)))
            ax = int(np.argmax(sizes[j, :2]))
            sz = sizes[j].copy()
            c = centers[j].copy()
            c[ax] += rng.uniform(0.12, 0.16) * rng.choice([-1.0, 1.0])
            zb = z0[j]
            kind[k] = 3
        else:
            c = np.array([rng.uniform(0.6, ex - 0.6), rng.uniform(0.6, ey - 0.6)])
            zb = 0.0
        sz[:2] = np.minimum(sz[:2], [ex - 0.3, ey - 0.3])
        c = np.clip(c, 0.5 * sz[:2] + 0.1, np.array([ex, ey]) - 0.5 * sz[:2] - 0.1)
        centers[k], sizes[k], z0[k] = c, sz, zb
        yaw[k] = rng.uniform(-0.25, 0.25) if (kind[k] in (0, 1) and rng.random() < 0.5) else 0.0
        if p_wall > 0.0 and kind[k] in (0, 1) and rng.random() < p_wall:
            # furniture pushed against a wall: the face towards the nearest wall lies in the wall's plane (within a
            # few millimetres), so the object's box and the wall box of reference scannet_planes.py:101-159 overlap
            # and the pair reaches the GP (gen_ps_utils.py:328-345, 401-437).  Drawn only when asked for, so the
            # random stream of every other scene is unchanged.
            ax = int(rng.integers(0, 2))
            lim = (ex, ey)[ax]
            off = rng.uniform(-0.003, 0.003)
            centers[k, ax] = (0.5 * sizes[k, ax] + off) if centers[k, ax] < 0.5 * lim else (lim - 0.5 * sizes[k, ax] - off)
            yaw[k] = 0.0

    # ---- surface areas -> point budget ------------------------------------------------
    obj_area = 2 * (sizes[:, 0] + sizes[:, 1]) * sizes[:, 2] + sizes[:, 0] * sizes[:, 1]
    n_floor = int(0.18 * n_points)
    n_wall = int(0.30 * n_points)
    n_obj = n_points - n_floor - n_wall
    per_obj = np.maximum(60, np.floor(n_obj * obj_area / obj_area.sum()).astype(np.int64))
    per_obj[-1] = max(60, n_obj - int(per_obj[:-1].sum()))

    all_xyz, all_rgb, all_sem, all_inst, all_spp = [], [], [], [], []
    spp_base = 0

    all_face = []

    def add(pts, rgb0, sem, inst, pid, npatch):
        nonlocal spp_base
        all_xyz.append(pts)
        all_rgb.append(np.clip(rgb0[None, :] + rng.normal(0, feat_noise, size=(len(pts), 3)), -1, 1))
        all_sem.append(np.full(len(pts), float(sem)))
        all_inst.append(np.full(len(pts), float(inst)))
        all_spp.append(pid + spp_base)
        all_face.append(np.full(len(pts), len(all_face), np.int64))
        spp_base += npatch

    # floor
    X, Y, Zv = np.eye(3)
    pts, a, b = _sample_rect(rng, n_floor, np.zeros(3), X, Y, ex, ey)
    pts[:, 2] += rng.normal(0, 0.004, size=n_floor)
    pid, npatch = _patch_ids(a, b, ex, ey, plane_patch)
    add(pts, rng.uniform(-0.6, 0.6, 3), 1, -100, pid, npatch)
    # walls
    wall_specs = [
        (np.array([0.0, 0.0, 0.0]), X, Zv, ex, ez),
        (np.array([0.0, ey, 0.0]), X, Zv, ex, ez),
        (np.array([0.0, 0.0, 0.0]), Y, Zv, ey, ez),
        (np.array([ex, 0.0, 0.0]), Y, Zv, ey, ez),
    ]
    wall_len = np.array([w[3] for w in wall_specs])
    per_wall = np.floor(n_wall * wall_len / wall_len.sum()).astype(np.int64)
    per_wall[-1] = n_wall - int(per_wall[:-1].sum())
    for (o, eu, ev, lu, lv), nw in zip(wall_specs, per_wall):
        pts, a, b = _sample_rect(rng, int(nw), o, eu, ev, lu, lv)
        pts += rng.normal(0, 0.004, size=pts.shape)
        pid, npatch = _patch_ids(a, b, lu, lv, plane_patch)
        add(pts, rng.uniform(-0.6, 0.6, 3), 0, -100, pid, npatch)
    # objects: top + 4 sides, rotated by yaw about their centre
    for k in range(K):
        sx, sy, sz = sizes[k]
        cx, cy = centers[k]
        faces = [
            (np.array([-sx / 2, -sy / 2, sz]), X, Y, sx, sy),
            (np.array([-sx / 2, -sy / 2, 0.0]), X, Zv, sx, sz),
            (np.array([-sx / 2, sy / 2, 0.0]), X, Zv, sx, sz),
            (np.array([-sx / 2, -sy / 2, 0.0]), Y, Zv, sy, sz),
            (np.array([sx / 2, -sy / 2, 0.0]), Y, Zv, sy, sz),
        ]
        area = np.array([f[3] * f[4] for f in faces])
        cnt = np.floor(per_obj[k] * area / area.sum()).astype(np.int64)
        cnt[0] += per_obj[k] - int(cnt.sum())
        cth, sth = math.cos(yaw[k]), math.sin(yaw[k])
        R = np.array([[cth, -sth, 0], [sth, cth, 0], [0, 0, 1.0]])
        col = rng.uniform(-0.9, 0.9, 3)
        sem = int(rng.integers(2, 20))
        for (o, eu, ev, lu, lv), nf in zip(faces, cnt):
            if nf <= 0:
                continue
            pts, a, b = _sample_rect(rng, int(nf), o, eu, ev, lu, lv)
            pts = pts @ R.T + np.array([cx, cy, z0[k]])[None, :]
            pts += rng.normal(0, 0.003, size=pts.shape)
            pid, npatch = _patch_ids(a, b, lu, lv, obj_patch)
            add(pts, col, sem, k, pid, npatch)

    xyz_al = np.concatenate(all_xyz, 0)
    rgb = np.concatenate(all_rgb, 0)
    sem = np.concatenate(all_sem, 0)
    inst = np.concatenate(all_inst, 0)
    spp = np.concatenate(all_spp, 0)

    if mesh_order:  # Z-order within every face (10 bits per axis over the scene's extent), faces in generation order
        lo, hi = xyz_al.min(0), xyz_al.max(0)
        q = np.clip(((xyz_al - lo) / np.maximum(hi - lo, 1e-9) * 1023.0).astype(np.int64), 0, 1023)
        code = np.zeros(len(xyz_al), np.int64)
        for bit in range(10):
            for ax in range(3):
                code |= ((q[:, ax] >> bit) & 1) << (3 * bit + ax)
        o = np.lexsort((code, np.concatenate(all_face)))
        xyz_al, rgb, sem, inst, spp = xyz_al[o], rgb[o], sem[o], inst[o], spp[o]
    # shuffle vertex order a little (mesh order is locally coherent, not sorted by patch)
    N = xyz_al.shape[0]
    blk = 256
    nblk = (N + blk - 1) // blk
    order = (rng.permutation(nblk)[:, None] * blk + np.arange(blk)[None, :]).reshape(-1)
    order = order[order < N]
    xyz_al, rgb, sem, inst, spp = xyz_al[order], rgb[order], sem[order], inst[order], spp[order]

    # non-contiguous, shuffled superpoint ids (int64) in [0, 4*n_spp)
    n_spp_ids = int(spp.max()) + 1
    remap = rng.permutation(4 * n_spp_ids)[:n_spp_ids].astype(np.int64)
    spp = remap[spp]

    # raw frame = inverse axis alignment of the aligned frame (yaw + translation), mean-centred
    th = rng.uniform(-math.pi, math.pi)
    c, s = math.cos(th), math.sin(th)
    A = np.eye(4)
    A[:3, :3] = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    centre_al = np.array([ex / 2, ey / 2, 0.0])
    # choose translation so that raw = R^T (aligned - t) is mean-centred
    Rm = A[:3, :3]
    raw = (xyz_al - centre_al[None, :]) @ Rm  # = R^T (x - centre)
    shift = raw.mean(0)
    raw = raw - shift[None, :]
    # aligned = R raw + t  with  t = centre + R shift
    A[:3, 3] = centre_al + Rm @ shift

    quads = None
    if with_walls_json:
        # ScanNet-Planes: verts in the raw *mesh* frame with y/z swapped relative to ours:
        # reference scannet_planes.py:190-193 maps (x, y, z) -> (x, -z, y) before aligning.
        verts, qs = [], []
        Ainv = np.linalg.inv(A)
        for (o, eu, ev, lu, lv) in wall_specs:
            corners_al = np.stack([o, o + eu * lu, o + eu * lu + ev * lv, o + ev * lv], 0)
            # real plane fits are noisy; exactly axis-aligned quads make the reference's
            # least-squares normal (scannet_planes.py:33-47) singular
            corners_al = corners_al + rng.normal(0, 0.01, size=corners_al.shape)
            corners_raw = (np.c_[corners_al, np.ones(4)] @ Ainv.T)[:, :3]
            base = len(verts)
            for p in corners_raw:
                # inverse of (x, y, z) -> (x, -z, y):  file stores (x, z', -y') s.t. mapping gives p
                verts.append([float(p[0]), float(p[2]), float(-p[1])])
            qs.append([base, base + 1, base + 2, base + 3])
        # one horizontal quad (ceiling) that the vertical filter must drop, one triangle that the
        # 4-vertex filter must drop
        ceil_al = np.array([[0, 0, ez], [ex, 0, ez], [ex, ey, ez], [0, ey, ez]], dtype=np.float64)
        ceil_raw = (np.c_[ceil_al, np.ones(4)] @ Ainv.T)[:, :3]
        base = len(verts)
        for p in ceil_raw:
            verts.append([float(p[0]), float(p[2]), float(-p[1])])
        qs.append([base, base + 1, base + 2, base + 3])
        qs.append([0, 1, 2])
        quads = {"quads": qs, "verts": verts}

    name = scan_name or ("scene%04d_00" % (seed % 10000))
    return Scene(
        scan_name=name,
        xyz=np.ascontiguousarray(raw),
        rgb=np.ascontiguousarray(rgb),
        sem=sem,
        inst=inst,
        spp=spp,
        axis_align=A,
        quads=quads,
        meta=dict(extent=(ex, ey, ez), n_objects=K, kinds=kind.tolist(), seed=seed),
    )


def make_gp_problem(seed: int, m1: int, m2: int, t: int, d: int = 6, sep: float = 1.4, std: float = 1.0):
    """Config-5 style direct GP problem: two Gaussian blobs in R^d.

    ``std`` scales the whole problem.  With the kernel's initial lengthscale ln 2 = 0.69, blobs of unit
    std in d = 32 put every pair of points ~8 lengthscales apart (kernel values ~1e-29): the fit is then
    driven by rounding noise that Adam's normalisation amplifies, and no two float64 implementations
    agree (DESIGN.md "Precision").  Use std ~0.3 for d = 32 to get a well-conditioned problem.

    Returns feats_spp f32[m1+m2+t, d] and the three index vectors that
    ``fit_gp_spp`` takes (reference gaussian_process_utils.py:382).
    """
    rng = np.random.default_rng(seed)
    mu = np.zeros(d)
    mu[0] = sep
    a = rng.normal(0, 1, size=(m1, d))
    b = rng.normal(0, 1, size=(m2, d)) + mu[None, :]
    w = rng.random(t)[:, None]
    c = rng.normal(0, 1, size=(t, d)) + w * mu[None, :]
    feats = (std * np.concatenate([a, b, c], 0)).astype(np.float32)
    b1 = np.arange(0, m1, dtype=np.int64)
    b2 = np.arange(m1, m1 + m2, dtype=np.int64)
    it = np.arange(m1 + m2, m1 + m2 + t, dtype=np.int64)
    return feats, b1, b2, it


def write_scannet_layout(scene: Scene, data_root: str, split: str = "train", deepfeat_dir: Optional[str] = None):
    """Write one scene in the on-disk layout reference gapro/gen_ps.py reads:

      <root>/<split>/<scan>_inst_nostuff.pth   tuple(xyz f64, rgb f64, sem f64, inst f64)
      <root>/superpoints/<scan>.pth            int64[N]
      <root>/scans_transform/<scan>/<scan>.txt "axisAlignment = 16 floats"
      <root>/scannet_planes/<scan>.json        optional wall quads
    """
    import json
    import os

    import torch

    os.makedirs(os.path.join(data_root, split), exist_ok=True)
    os.makedirs(os.path.join(data_root, "superpoints"), exist_ok=True)
    os.makedirs(os.path.join(data_root, "scans_transform", scene.scan_name), exist_ok=True)
    torch.save((scene.xyz, scene.rgb, scene.sem, scene.inst),
               os.path.join(data_root, split, scene.scan_name + "_inst_nostuff.pth"))
    torch.save(scene.spp, os.path.join(data_root, "superpoints", scene.scan_name + ".pth"))
    with open(os.path.join(data_root, "scans_transform", scene.scan_name, scene.scan_name + ".txt"), "w") as f:
        f.write("colorHeight = 968\n")
        f.write("axisAlignment = " + " ".join(repr(float(v)) for v in scene.axis_align.reshape(-1)) + "\n")
        f.write("numDepthFrames = 1\n")
    if scene.quads is not None:
        os.makedirs(os.path.join(data_root, "scannet_planes"), exist_ok=True)
        with open(os.path.join(data_root, "scannet_planes", scene.scan_name + ".json"), "w") as f:
            json.dump(scene.quads, f)
    if deepfeat_dir is not None:
        os.makedirs(deepfeat_dir, exist_ok=True)
        rng = np.random.default_rng(scene.meta.get("seed", 0) + 99)
        proj = rng.standard_normal((6, 32)) / np.sqrt(6.0)
        torch.save((scene.default_feats() @ proj * 0.5).astype(np.float32),
                   os.path.join(deepfeat_dir, scene.scan_name + ".pth"))
