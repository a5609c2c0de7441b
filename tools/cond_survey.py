#!/usr/bin/env python3
"""Conditioning figures (gapro_svgp_fit_batch_ex) of the S3DIS-shaped test scene's 66 fits next to their deviation from
the float64 autograd oracle: where the threshold GAPRO_COND_ILL sits (run on the GPU box).
    python tools/cond_survey.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    torch.set_num_threads(8)
    from gapro_amd.gaussian_process_utils import fit_gp_spp_batch
    from gapro_amd.gen_ps_utils import getInstanceInfo
    from gapro_amd.synth import make_scene
    from oracle import gen_ps_oracle as O
    from oracle.svgp_oracle import fit_gp_spp_oracle

    sc = make_scene(seed=7, n_points=1_000_000, n_objects=40, with_walls_json=False, obj_patch=60, plane_patch=400)
    xyz = sc.aligned_xyz()
    _, cls, box, vol, _ = getInstanceInfo(xyz, sc.inst, sc.sem)
    kw = dict(coords_float=xyz, mask_feats=sc.default_feats().astype(np.float32), spp=sc.spp,
              instance_cls=cls.astype(np.int64), instance_box=box.astype(np.float32),
              instance_box_volume=vol.astype(np.float32), wall_box=[], wall_box_volume=[], instance_classes=13,
              ground_h=0.1, training_iter=50, thresh_spp_occu=0.999)
    ref, dbg = O.gen_pseudo_label_gaussian_process(
        **kw, fit_fn=lambda f, b1, b2, it: fit_gp_spp_oracle(f, b1, b2, it, 50, impl="autograd", dtype="f64"),
        return_debug=True)
    fits = [e for e in dbg["events"] if e.kind == "fit"]
    got, res = fit_gp_spp_batch(dbg["part"].feats_spp, [(e.b1_inds, e.b2_inds, e.intersect_inds) for e in fits],
                                training_iter=50, reproducibility_probe=True)
    rows = []
    for k, (g, r, e) in enumerate(zip(got, dbg["results"], fits)):
        dp = np.max(np.abs(g[0].astype(np.float64) - r[0]))
        dv = np.max(np.abs(g[4].astype(np.float64) - r[4]) / r[4])
        rows.append((float(res["cond"][k]), k, len(e.b1_inds) + len(e.b2_inds), dv, dp, res["repro_dv"][k], res["repro_dp"][k]))
    for c, k, m, dv, dp, rv, rp in sorted(rows, reverse=True):
        print("fit %2d  M = %4d  cond %.3e  vs oracle dv %.2e dp %.2e  | own probe (jitter x (1 + 1e-11)) dv %.2e dp %.2e"
              % (k, m, c, dv, dp, rv, rp))


if __name__ == "__main__":
    main()
