#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference code in this container.

Run from the repo root:  python tests/golden/make_golden.py
Needs /root/reference (read-only); the fixtures it writes are plain data (inputs and the
reference's outputs) and travel to the GPU box, the reference does not.

What is imported as-is from /root/reference/gapro:
  gen_ps_utils.gen_pseudo_label_gaussian_process, gen_ps_utils.getInstanceInfo,
  scannet_planes.get_wall_boxes, eval_ps_labels.get_miou_scene
What has to be stubbed *in this harness only* because it is not installed:
  torch_scatter -> ~30-line pure-torch shim (sequential CPU semantics: sum/mean, first-min arg)
  gpytorch      -> empty module tree (only needed so the import of gaussian_process_utils succeeds);
                   ``gen_ps_utils.fit_gp_spp`` is replaced by a deterministic stand-in whose
                   inputs and outputs are recorded, so the partition, the static pair schedule
                   and the merge/fallback/label logic are pinned independently of GP numerics.
GP numerics themselves cannot be pinned from the reference (no gpytorch) -- see oracle/__init__.py.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
REF = "/root/reference/gapro"

from gapro_amd.synth import make_scene  # noqa: E402

CAPTURE = {}


def install_stubs():
    ts = types.ModuleType("torch_scatter")

    def scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
        assert dim == 0
        n = int(index.max()) + 1 if dim_size is None else dim_size
        res = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype)
        res.index_add_(0, index[:, 0] if index.dim() > 1 else index, src)  # sequential on CPU
        if reduce == "mean":
            ones = torch.ones(src.shape[0], dtype=src.dtype)
            cnt = torch.zeros(n, dtype=src.dtype).index_add_(0, index[:, 0] if index.dim() > 1 else index, ones)
            cnt.clamp_(1)
            res = res / (cnt[:, None] if res.dim() > 1 else cnt)
        CAPTURE.setdefault("scatter", []).append(res.clone())
        return res

    def scatter_min(src, index, dim=0):
        n = int(index.max()) + 1 if len(index) else 0
        out = torch.full((n,), float("inf"), dtype=src.dtype)
        arg = torch.full((n,), len(src), dtype=torch.long)
        for e in range(len(src)):  # torch_scatter CPU kernel: strict '<' -> first minimum wins
            i = int(index[e])
            if src[e] < out[i]:
                out[i] = src[e]
                arg[i] = e
        return out, arg

    def scatter_add(src, index, dim=0):
        return scatter(src, index, dim=dim, reduce="sum")

    ts.scatter, ts.scatter_min, ts.scatter_add = scatter, scatter_min, scatter_add
    sys.modules["torch_scatter"] = ts

    gp = types.ModuleType("gpytorch")
    for sub in ["mlls", "mlls.variational_elbo", "models", "variational", "means", "kernels", "distributions",
                "likelihoods"]:
        mod = types.ModuleType("gpytorch." + sub)
        sys.modules["gpytorch." + sub] = mod
        parent = gp
        parts = sub.split(".")
        for p in parts[:-1]:
            parent = getattr(parent, p)
        setattr(parent, parts[-1], mod)
    sys.modules["gpytorch"] = gp
    gp.mlls.variational_elbo.VariationalELBO = object
    gp.models.AbstractVariationalGP = object
    gp.variational.CholeskyVariationalDistribution = object
    gp.variational.VariationalStrategy = object
    np.int = int  # scannet_planes.py:225 uses the alias NumPy 1.24 removed
    torch.Tensor.cuda = lambda self, *a, **k: self  # eval_ps_labels.py:102,111 hard-code .cuda()


def fake_fit(coords_float_spp, feats_spp, b1_inds, b2_inds, intersect_inds, training_iter=50):
    """Deterministic stand-in for fit_gp_spp (same return contract, gaussian_process_utils.py:445)."""
    t = len(intersect_inds)
    seed = (int(b1_inds.sum()) * 1000003 + int(b2_inds.sum()) * 10007 + int(intersect_inds.sum())) % (2**31)
    g = torch.Generator().manual_seed(seed)
    probs = torch.rand(t, generator=g)
    # a few exact ties / extremes to exercise the strict '<' overwrite rule
    if t > 2:
        probs[0] = 0.5
        probs[1] = 1.0
    labels = probs.ge(0.5)
    probs_new = torch.where(labels == 1, probs, 1 - probs)
    mu = torch.randn(t, generator=g)
    var = torch.rand(t, generator=g) + 0.01
    CAPTURE.setdefault("fits", []).append(dict(
        b1_inds=b1_inds.numpy().copy(), b2_inds=b2_inds.numpy().copy(), intersect_inds=intersect_inds.numpy().copy(),
        probs=probs.numpy().copy(), probs_new=probs_new.numpy().copy(), labels=labels.numpy().copy(),
        mu=mu.numpy().copy(), var=var.numpy().copy()))
    return probs, probs_new, labels, mu, var


SCENES = [
    # name, kwargs
    ("s0_walls", dict(seed=11, n_points=6000, n_objects=12, with_walls_json=True, obj_patch=25, plane_patch=80)),
    ("s1_nowalls", dict(seed=5, n_points=5000, n_objects=10, with_walls_json=False, obj_patch=25, plane_patch=80)),
    ("s2_dense", dict(seed=23, n_points=8000, n_objects=20, with_walls_json=True, obj_patch=20, plane_patch=60)),
    ("s4_dups", dict(seed=7, n_points=6000, n_objects=10, with_walls_json=False, obj_patch=20, plane_patch=80,
                     p_kinds=(0.2, 0.15, 0.3))),
    ("s3_bigspp", dict(seed=42, n_points=12000, n_objects=4, with_walls_json=False, obj_patch=1100,
                       plane_patch=1300)),
    # furniture flush against the walls and small floor patches: the reference's recorded schedule holds
    # wall-object and floor-object GP pairs (the reason it adds those boxes, gen_ps_utils.py:328-345, 401-437)
    ("s5_lean", dict(seed=32, n_points=9000, n_objects=10, with_walls_json=True, obj_patch=20, plane_patch=25,
                     p_wall=0.6)),
]


def main():
    install_stubs()
    sys.path.insert(0, REF)
    import eval_ps_labels as ref_eval
    import gen_ps_utils as ref_utils
    import scannet_planes as ref_planes

    ref_utils.fit_gp_spp = fake_fit
    ref_utils.tqdm = lambda x: x
    outdir = os.path.dirname(os.path.abspath(__file__))
    summary = {}
    for name, kw in SCENES:
        sc = make_scene(**kw)
        CAPTURE.clear()
        xyz = sc.aligned_xyz()
        info = ref_utils.getInstanceInfo(xyz, instance_label=sc.inst, semantic_label=sc.sem)
        instance_num, instance_cls, instance_box, instance_box_volume, corners_label = info
        # get_wall_boxes reads dataset/scannetv2/... relative to cwd (scannet_planes.py:163,178)
        with tempfile.TemporaryDirectory() as td:
            cwd = os.getcwd()
            os.chdir(td)
            try:
                os.makedirs("dataset/scannetv2/scans_transform/" + sc.scan_name)
                with open("dataset/scannetv2/scans_transform/%s/%s.txt" % (sc.scan_name, sc.scan_name), "w") as f:
                    f.write("axisAlignment = " + " ".join(repr(float(v)) for v in sc.axis_align.reshape(-1)) + "\n")
                if sc.quads is not None:
                    os.makedirs("dataset/scannetv2/scannet_planes")
                    with open("dataset/scannetv2/scannet_planes/%s.json" % sc.scan_name, "w") as f:
                        json.dump(sc.quads, f)
                wall_cls, wall_box, wall_volume = ref_planes.get_wall_boxes(sc.scan_name)
            finally:
                os.chdir(cwd)
        # exactly the casts of gen_ps.py:79-89 (minus .cuda())
        t_cls = torch.from_numpy(instance_cls).long()
        t_box = torch.from_numpy(instance_box).float()
        t_vol = torch.from_numpy(instance_box_volume).float()
        t_xyz = torch.from_numpy(xyz)
        t_spp = torch.from_numpy(sc.spp)
        t_feats = torch.from_numpy(sc.default_feats()).float()
        if len(wall_box) > 0:
            t_wbox = torch.from_numpy(wall_box).float()
            t_wvol = torch.from_numpy(wall_volume).float()
        else:
            t_wbox, t_wvol = wall_box, wall_volume
        outs = ref_utils.gen_pseudo_label_gaussian_process(
            t_xyz, t_feats, t_spp, t_cls, t_box, t_vol, t_wbox, t_wvol, instance_classes=18,
            dataset_name="scannetv2", ground_h=0.1, training_iter=50, thresh_spp_occu=0.999)
        sem, ins, prob, mu, var = [o.numpy() for o in outs]
        coords_spp, feats_spp, occ_mean = [c.numpy() for c in CAPTURE["scatter"][:3]]
        # quality metric of gen_ps.py:116-124
        gt_sem = torch.from_numpy(sc.sem).int()
        gt_ins = torch.from_numpy(sc.inst).int()
        gt_sem[gt_sem != -100] -= 2
        gt_sem[(gt_sem == -1) | (gt_sem == -2)] = 18
        ious = ref_eval.get_miou_scene(gt_sem.long(), gt_ins.long(), outs[0].long(), outs[1].long()).numpy()

        fits = CAPTURE.get("fits", [])
        rec = dict(
            xyz_raw=sc.xyz, rgb=sc.rgb, sem_gt=sc.sem, inst_gt=sc.inst, spp=sc.spp, axis_align=sc.axis_align,
            quads_json=np.array(json.dumps(sc.quads) if sc.quads is not None else ""),
            scan_name=np.array(sc.scan_name),
            xyz_aligned=xyz,
            gi_instance_num=np.int64(instance_num), gi_cls=instance_cls, gi_box=instance_box,
            gi_vol=instance_box_volume, gi_corners_sum=np.float64(corners_label.astype(np.float64).sum()),
            gi_corners_head=corners_label[:64],
            wall_cls=np.asarray(wall_cls), wall_box=np.asarray(wall_box, dtype=np.float64),
            wall_vol=np.asarray(wall_volume, dtype=np.float64),
            ref_coords_spp=coords_spp, ref_feats_spp=feats_spp, ref_occ_mean=occ_mean,
            out_sem=sem, out_inst=ins, out_prob=prob, out_mu=mu, out_var=var, ref_ious=ious,
            n_fits=np.int64(len(fits)),
        )
        for i, f in enumerate(fits):
            for k, v in f.items():
                rec["fit%03d_%s" % (i, k)] = v
        np.savez_compressed(os.path.join(outdir, name + ".npz"), **rec)
        ms = [len(f["b1_inds"]) + len(f["b2_inds"]) for f in fits]
        n_inst, n_box = len(instance_box), len(instance_box) + len(wall_box) + 1
        # which boxes a recorded fit pairs: its training superpoints lie in exactly one box each
        occ = occ_mean >= np.float32(0.999)
        pairs = [(int(np.argmax(occ[f["b1_inds"][0]])), int(np.argmax(occ[f["b2_inds"][0]]))) for f in fits]
        n_wall_fits = sum(1 for a, b in pairs if n_inst <= max(a, b) < n_box - 1)
        n_floor_fits = sum(1 for a, b in pairs if max(a, b) == n_box - 1)
        summary[name] = dict(N=int(len(sem)), S=int(len(mu)), B=int(n_box),
                             n_fits=len(fits), M_max=int(max(ms) if ms else 0), walls=int(len(wall_box)),
                             n_gp_labelled=int((mu != -100).sum()), wall_fits=n_wall_fits,
                             floor_fits=n_floor_fits)
        print(name, summary[name])
    with open(os.path.join(outdir, "SUMMARY.json"), "w") as f:
        json.dump(summary, f, indent=1)


if __name__ == "__main__":
    main()
