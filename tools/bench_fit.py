#!/usr/bin/env python3
"""Fit-kernel micro-benchmark: batches of equal-size GP fits (run on the GPU box).

python tools/bench_fit.py [--sizes 32,64,96,128,160,256] [--fits 512] [--t 32] [--d 6] [--iters 50]
Prints per size: launch ms, fits/s, achieved TFLOP/s (SURVEY 8d formula) and fraction of the FP64 MFMA peak.
"""
import argparse
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch  # noqa: E402

from gapro_amd._lib import FitDesc  # noqa: E402
from gapro_amd.pipeline import Pipeline, fit_flops  # noqa: E402
from gapro_amd.synth import make_gp_problem  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="32,64,96,128,160,256")
    ap.add_argument("--fits", type=int, default=512)
    ap.add_argument("--t", type=int, default=32)
    ap.add_argument("--d", type=int, default=6)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--profile", action="store_true", help="use libgapro_hip_prof.so and print phase shares")
    ap.add_argument("--force-staged", action="store_true")
    args = ap.parse_args()
    if args.profile:
        import os
        from gapro_amd import _lib
        _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libgapro_hip_prof.so")
    pipe = Pipeline(device=0, training_iter=args.iters, force_staged=args.force_staged)
    for m in [int(s) for s in args.sizes.split(",")]:
        m1 = m // 2
        m2 = m - m1
        n_distinct = 8
        feats_l, probs = [], []
        base = 0
        for i in range(n_distinct):
            f, b1, b2, it = make_gp_problem(i, m1, m2, args.t, args.d)
            feats_l.append(f)
            probs.append((b1 + base, b2 + base, it + base))
            base += len(f)
        feats = torch.from_numpy(np.concatenate(feats_l)).cuda()
        n = args.fits
        descs = (FitDesc * n)()
        idx = []
        io = oo = 0
        for i in range(n):
            b1, b2, it = probs[i % n_distinct]
            d = descs[i]
            d.m1, d.m2, d.t = len(b1), len(b2), len(it)
            d.idx_offset, d.out_offset = io, oo
            idx += [b1, b2, it]
            io += len(b1) + len(b2) + len(it)
            oo += len(it)
        h_idx = np.concatenate(idx).astype(np.int32)
        pipe.profile_fit = True
        times = []
        for r in range(args.reps + 1):
            pipe.fit_events = []
            res = pipe.fit_descs(feats, descs, n, h_idx, oo, keep_debug=args.profile)
            torch.cuda.synchronize()
            e0, e1, fl = pipe.fit_events[0]
            if r > 0:
                times.append(e0.elapsed_time(e1))
        ms = float(np.median(times))
        tf = fl / (ms * 1e-3) / 1e12
        print("M=%4d T=%3d D=%2d fits=%4d iters=%d : %9.2f ms/launch  %9.1f fits/s  %7.3f TFLOP/s  (%.2f%% of 78.6)  "
              "per-fit-step %.1f us (at %d concurrent)" % (m, args.t, args.d, n, args.iters, ms, n / (ms * 1e-3), tf,
                                                          100 * tf / 78.6, 1e3 * ms / max(args.iters, 1) /
                                                          max(1, (n + 255) // 256), min(n, 256)), flush=True)
        if args.profile:
            import ctypes as C
            from gapro_amd import _lib
            lay = (C.c_int64 * 8)()
            _lib.load().gapro_fit_workspace_layout(m, args.t, args.d, C.cast(lay, C.c_void_p))
            scal, total = int(lay[6]), int(lay[7])
            ws = res["workspace"].cpu().numpy()
            prof = np.stack([ws[i * total + scal + 24: i * total + scal + 44] for i in range(min(n, 64))]).mean(0)
            names = ["chol:update", "chol:rest", "inv", "kx", "A+BMT+meanvar", "chol:diag", "quad/kl", "Gm+GA",
                     "GLS+adam", "GKX", "GL", "Pm", "T1", "G", "kgrads+adamZ", "-", "adam", "predict", "chol:panel",
                     "misc"]
            if not args.force_staged and m <= 128:
                names = ["chol:update", "chol:rest", "inv", "s:fill", "s:A", "chol:diag", "post-strips", "s:B",
                         "GLS+adam", "s:meanvar", "s:lik", "s:scale+GLSacc", "s:GA", "tail", "kgrads+adamZ",
                         "s:GKX", "adam", "predict", "s:GLacc+store", "misc"]
            tot = prof.sum()
            print("    phases (us per fit, share): " + "  ".join(
                "%s %.0f (%.0f%%)" % (nm, v / 100.0, 100 * v / tot) for nm, v in zip(names, prof) if v > 0), flush=True)


if __name__ == "__main__":
    main()
