"""The ill-conditioned fits of the S3DIS-shaped test scene (tests/test_pipeline_gpu.py::test_s3dis_shaped_scene_matches_oracle):
for every fit the kernel does not reproduce to float32 rounding, the kernel's deviation from the float64 autograd oracle
next to the deviation between the oracle's own two implementations -- how much room the test's 30x rule leaves on this box.
    python tools/loose_fits.py [--lib libgapro_hip_fastmath.so]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default="")
    ap.add_argument("--oracle-threads", type=int, default=8, help="torch threads of the CPU oracle (tests/conftest.py: 8); "
                    "another count reorders its BLAS sums -- on the ill-conditioned fits that alone moves ITS results by "
                    "1e-5 .. 3e-4 (256 threads on the GPU box: fit 46's sigma^2 by 3e-4; and takes 7 minutes)")
    args = ap.parse_args()
    import torch
    torch.set_num_threads(args.oracle_threads)
    from gapro_amd import _lib
    if args.lib:
        _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), args.lib)
    from gapro_amd.gaussian_process_utils import fit_gp_spp_batch
    from gapro_amd.gen_ps_utils import getInstanceInfo
    from gapro_amd.synth import make_scene
    from oracle import gen_ps_oracle as O
    from oracle import svgp_oracle as so
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
    got = fit_gp_spp_batch(dbg["part"].feats_spp, [(e.b1_inds, e.b2_inds, e.intersect_inds) for e in fits], training_iter=50)
    worst = 0.0
    for k, (g, r, e) in enumerate(zip(got, dbg["results"], fits)):
        dp = np.max(np.abs(g[0].astype(np.float64) - r[0]))
        dv = np.max(np.abs(g[4].astype(np.float64) - r[4]) / r[4])
        if dp < 3e-7 and dv < 1e-5:
            worst = max(worst, dv)
            continue
        f = dbg["part"].feats_spp
        X = np.concatenate([f[e.b1_inds], f[e.b2_inds]]).astype(np.float64)
        y = np.r_[-np.ones(len(e.b1_inds)), np.ones(len(e.b2_inds))]
        Xt = f[e.intersect_inds].astype(np.float64)
        a = so.svgp_fit_predict_autograd(X, y, Xt, 50, "f64")
        m = so.svgp_fit_predict_manual(X, y, Xt, 50)
        odv, odp = np.max(np.abs(a[1] - m[1]) / a[1]), np.max(np.abs(a[2] - m[2]))
        # the fit's own conditioning: ONE input coordinate moved by ONE ulp, same implementation, same thread count
        Xp = X.copy()
        Xp[0, 0] = np.nextafter(Xp[0, 0], np.inf)
        u = so.svgp_fit_predict_autograd(Xp, y, Xt, 50, "f64")
        udv = np.max(np.abs(u[1] - a[1]) / a[1])
        print("fit %d (M = %d): kernel vs autograd oracle dv %.2e dp %.2e | oracle autograd vs manual dv %.2e dp %.2e | "
              "ratio %.1f (the test allows 30) | the autograd oracle with one input moved by one ulp: dv %.2e"
              % (k, len(X), dv, dp, odv, odp, dv / odv, udv))
    print("%d fits; every other fit within dv %.1e" % (len(fits), worst))


if __name__ == "__main__":
    main()
