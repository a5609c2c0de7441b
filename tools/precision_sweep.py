#!/usr/bin/env python3
"""BASELINE configs[4]: precision sweep of the GP fit (run on the GPU box).

    python tools/precision_sweep.py [--sizes 64,128,512,1024,2048] [--fits 32] [--out profiles/r02_precision_sweep.md]

Per size: `--fits` concurrent fits (4 distinct two-blob problems) through ONE kernel (the cluster
kernel, debug bit 4) in float64 and in the reference's float32 / float64 split ("mixed": v_mfma_f32 for L_S^T A, dA,
dL_S; float64 for the Cholesky factor, the L^-1 products and their backward), 50 Adam steps, compared with the float64
autograd oracle: max relative error of sigma^2, max error of mu (relative to max |mu|), of p, and label flips among the
test superpoints.  Two more precision classes are studied on the CPU with the oracle (they are not kernel modes):
float32 everywhere including the factorisation, and the split fed with bfloat16-rounded features.  Launch time and
algorithmic TFLOP/s of both kernel modes are printed beside the errors.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def errs(out, ref):
    mu, var, p = ref
    dv = float(np.max(np.abs(out[4].astype(np.float64) - var) / var))
    dm = float(np.max(np.abs(out[3].astype(np.float64) - mu)) / max(float(np.max(np.abs(mu))), 1e-30))
    dp = float(np.max(np.abs(out[0].astype(np.float64) - p)))
    flips = int(np.sum(out[2] != (p.astype(np.float32) >= np.float32(0.5))))
    return dv, dm, dp, flips


def bf16_round(x):
    """float32 array rounded to bfloat16 (round to nearest even) and widened again: what a bf16 input tensor holds"""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(x))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="64,128,512,1024,2048")
    ap.add_argument("--fits", type=int, default=32)
    ap.add_argument("--t", type=int, default=64)
    ap.add_argument("--oracle-max", type=int, default=1024, help="largest M the CPU oracle is run at")
    ap.add_argument("--cpu-study-max", type=int, default=512, help="largest M of the float32 / bfloat16 CPU study")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch

    from gapro_amd.gaussian_process_utils import fit_gp_spp_batch
    from gapro_amd.pipeline import fit_flops_each
    from gapro_amd.synth import make_gp_problem
    from oracle import svgp_oracle as so

    torch.set_num_threads(min(16, os.cpu_count() or 1))
    rows = []
    for m in [int(v) for v in args.sizes.split(",")]:
        m1, m2 = m // 2, m - m // 2
        parts, probs, base = [], [], 0
        for i in range(4):
            f, b1, b2, it = make_gp_problem(4000 + 10 * m + i, m1, m2, args.t, 6)
            parts.append(f)
            probs.append((b1 + base, b2 + base, it + base))
            base += len(f)
        feats = np.concatenate(parts)
        launch = [probs[i % 4] for i in range(args.fits)]
        res = {}
        for mode in ("f64", "mixed", "bf16in"):
            # round 6: "bf16in" = the device's MIXED kernel fed with bfloat16-rounded features (config 5's third leg as a
            # HIP run, not only as an oracle study)
            fin, prec = (bf16_round(feats), "mixed") if mode == "bf16in" else (feats, mode)
            fit_gp_spp_batch(fin, launch[:4], training_iter=2, precision=prec, cluster_all=True)  # warm
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fit_gp_spp_batch(fin, launch, training_iter=50, precision=prec, cluster_all=True)
            torch.cuda.synchronize()
            res[mode] = (out, time.perf_counter() - t0)
        flops = args.fits * (50 * (8.33 * m ** 3 + 12 * 6 * m * m) + m ** 3 / 3 + 2 * m * m * args.t + 12 * (m * m + m * args.t))
        row = {"M": m}
        ref = None
        if m <= args.oracle_max:
            worst = {"f64": [0, 0, 0, 0], "mixed": [0, 0, 0, 0], "bf16in": [0, 0, 0, 0]}
            for k in range(4):
                b1, b2, it = probs[k]
                X = np.concatenate([feats[b1], feats[b2]]).astype(np.float64)
                y = np.r_[-np.ones(m1), np.ones(m2)]
                Xt = feats[it].astype(np.float64)
                ref = so.svgp_fit_predict_autograd(X, y, Xt, 50, "f64")
                for mode in ("f64", "mixed", "bf16in"):
                    e = errs(res[mode][0][k], ref)
                    worst[mode] = [max(a, b) for a, b in zip(worst[mode], e)]
                if m <= args.cpu_study_max and k == 0:
                    for cm in ("mixed", "f32", "bf16in"):
                        try:
                            mu, var, p = so.svgp_fit_predict_autograd(X, y, Xt, 50, cm)
                            fake = (p.astype(np.float32), None, p.astype(np.float32) >= np.float32(0.5), mu, var)
                            row["cpu_" + cm] = errs(fake, ref)
                        except Exception as ex:  # noqa: BLE001 - e.g. float32 Cholesky not positive definite
                            row["cpu_" + cm] = "failed: %s" % str(ex)[:40]
            row["gpu_f64"], row["gpu_mixed"], row["gpu_bf16in"] = worst["f64"], worst["mixed"], worst["bf16in"]
        else:  # no oracle at this size: the split against the float64 kernel (which matches the oracle elsewhere)
            worst, worst_b = [0, 0, 0, 0], [0, 0, 0, 0]
            for k in range(4):
                o64 = res["f64"][0][k]
                r64 = (o64[3].astype(np.float64), o64[4].astype(np.float64), o64[0].astype(np.float64))
                worst = [max(a, b) for a, b in zip(worst, errs(res["mixed"][0][k], r64))]
                worst_b = [max(a, b) for a, b in zip(worst_b, errs(res["bf16in"][0][k], r64))]
            row["gpu_mixed_vs_gpu_f64"], row["gpu_bf16in_vs_gpu_f64"] = worst, worst_b
        for mode in ("f64", "mixed"):
            row["ms_" + mode] = 1e3 * res[mode][1]
            row["tf_" + mode] = flops / res[mode][1] / 1e12
        # every copy of a problem must agree bit for bit within a launch
        for mode in ("f64", "mixed"):
            o = res[mode][0]
            row["bitwise_" + mode] = all(np.array_equal(a, b) for i in range(4, args.fits) for a, b in zip(o[i], o[i % 4]))
        rows.append(row)
        print(row, flush=True)
    lines = ["# Precision sweep of the GP fit (BASELINE configs[4]) -- MI355X, round 6 (round 2's table + the bfloat16-input "
             "leg as a device run)", "",
             "%d concurrent fits per size (4 distinct two-blob problems x %d), T = %d test superpoints, D = 6, 50 Adam steps, "
             "all through the cluster kernel (`gapro_fit_options.reserved` bit 4).  Errors are maxima over the 4 problems "
             "against the float64 autograd oracle (`oracle/svgp_oracle.py`, PARITY UNPINNED against gpytorch): "
             "relative error of sigma^2, error of mu relative to max|mu|, absolute error of p, label flips." % (args.fits, args.fits // 4, args.t), "",
             "| M | kernel float64: sigma^2 / mu / p / flips | kernel mixed (reference split) | kernel mixed on bfloat16-rounded features | launch ms f64 -> mixed | TFLOP/s f64 -> mixed | copies bitwise equal |",
             "|---|---|---|---|---|---|---|"]
    fmt = lambda e: "%.1e / %.1e / %.1e / %d" % tuple(e)  # noqa: E731
    for r in rows:
        if "gpu_f64" in r:
            a, b, c = fmt(r["gpu_f64"]), fmt(r["gpu_mixed"]), fmt(r["gpu_bf16in"])
        else:
            a, b = "(no oracle at this size)", "vs float64 kernel: " + fmt(r["gpu_mixed_vs_gpu_f64"])
            c = "vs float64 kernel: " + fmt(r["gpu_bf16in_vs_gpu_f64"])
        lines.append("| %d | %s | %s | %s | %.0f -> %.0f | %.1f -> %.1f | %s / %s |" % (
            r["M"], a, b, c, r["ms_f64"], r["ms_mixed"], r["tf_f64"], r["tf_mixed"], r["bitwise_f64"], r["bitwise_mixed"]))
    lines += ["", "CPU study with the oracle (problem 0 of each size; not kernel modes): the split restated in torch, float32 "
                  "everywhere including the Cholesky factorisation, and the split on bfloat16-rounded features.", "",
              "| M | oracle mixed | oracle all-float32 | oracle bfloat16 inputs |", "|---|---|---|---|"]
    for r in rows:
        if "cpu_mixed" in r:
            cell = lambda v: v if isinstance(v, str) else fmt(v)  # noqa: E731
            lines.append("| %d | %s | %s | %s |" % (r["M"], cell(r["cpu_mixed"]), cell(r["cpu_f32"]), cell(r["cpu_bf16in"])))
    text = "\n".join(lines) + "\n"
    print(text)
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(text)


if __name__ == "__main__":
    main()
