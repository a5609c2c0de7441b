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


def build_mix(mix, args):
    groups = [(int(a.split(":")[0]), int(a.split(":")[1])) for a in mix.split(",")]
    feats_l, sel = [], []
    base = 0
    for m, cnt in groups:
        for i in range(4):
            f, b1, b2, it = make_gp_problem(i, m // 2, m - m // 2, args.t, args.d)
            feats_l.append(f)
            sel.append((m, b1 + base, b2 + base, it + base))
            base += len(f)
    feats = torch.from_numpy(np.concatenate(feats_l)).cuda()
    n = sum(c for _, c in groups)
    descs = (FitDesc * n)()
    idx = []
    io = oo = k = 0
    for gi, (m, cnt) in enumerate(groups):
        for i in range(cnt):
            _, b1, b2, it = sel[4 * gi + i % 4]
            d = descs[k]
            d.m1, d.m2, d.t = len(b1), len(b2), len(it)
            d.idx_offset, d.out_offset = io, oo
            idx += [b1, b2, it]
            io += len(b1) + len(b2) + len(it)
            oo += len(it)
            k += 1
    h_idx = np.concatenate(idx).astype(np.int32)
    return feats, descs, n, h_idx, oo


def run_two_streams(pipe, args):
    """--mix 'A;B': batch A on one torch stream, batch B on another, launched back to back from the host."""
    import time
    mixes = args.mix.split(";")
    batches = [build_mix(m, args) for m in mixes]
    streams = [torch.cuda.Stream() for _ in batches]
    for mode in ("serial", "two streams"):
        for r in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pend = []
            for i, (b, st) in enumerate(zip(batches, streams)):
                with torch.cuda.stream(st if mode != "serial" else streams[0]):
                    pend.append(pipe.fit_launch(*b, slot="s%d" % i))
            for p in pend:
                pipe.fit_collect(p)
            torch.cuda.synchronize()
            dt = 1e3 * (time.perf_counter() - t0)
        print("%s: %-12s %.2f ms" % (args.mix, mode, dt))


def run_mix(pipe, args):
    if ";" in args.mix:
        return run_two_streams(pipe, args)
    feats, descs, n, h_idx, oo = build_mix(args.mix, args)
    pipe.profile_fit = True
    times = []
    for r in range(args.reps + 1):
        pipe.fit_events = []
        with torch.cuda.stream(torch.cuda.Stream() if args.own_stream else torch.cuda.current_stream()):
            pipe.fit_descs(feats, descs, n, h_idx, oo)
        torch.cuda.synchronize()
        ev = pipe.fit_events[0]
        fl = ev.flops
        if r > 0:
            times.append(ev.read()[2])
        else:
            ev.read()
    ms = float(np.median(times))
    print("mix %s fork=%s : %9.2f ms/launch  %9.1f fits/s  %7.3f TFLOP/s" % (args.mix, not args.no_fork, ms,
                                                                           n / (ms * 1e-3), fl / (ms * 1e-3) / 1e12))


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
    ap.add_argument("--lib", default="", help="alternative library file name inside gapro_amd/ (A/B runs)")
    ap.add_argument("--mix", default="", help="one launch of mixed sizes, e.g. 160:128,96:1024,64:1024 (M:count)")
    ap.add_argument("--no-fork", action="store_true", help="both fit kernels on one stream")
    ap.add_argument("--own-stream", action="store_true", help="launch from a non-default torch stream")
    ap.add_argument("--no-small", action="store_true", help="M_p <= 64 on the 512-thread strip kernel (A/B)")
    ap.add_argument("--no-cluster", action="store_true", help="large fits stay on one workgroup (A/B)")
    ap.add_argument("--cluster-all", action="store_true", help="the cluster kernel for every fit it can take (A/B)")
    ap.add_argument("--wide-tiles", action="store_true", help="64 x 64 wave tiles in the cluster kernel (A/B)")
    ap.add_argument("--full-barriers", action="store_true", help="cluster barriers always with the L2 write-back (A/B)")
    ap.add_argument("--flags", type=int, default=0, help="further gapro_fit_options.reserved debug bits (A/B)")
    args = ap.parse_args()
    if args.profile:
        import os
        from gapro_amd import _lib
        _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libgapro_hip_prof.so")
    if args.lib:
        import os
        from gapro_amd import _lib
        _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), args.lib)
    pipe = Pipeline(device=0, training_iter=args.iters, force_staged=args.force_staged)
    if args.no_fork:
        pipe.opt.reserved |= 2
    if args.no_small:
        pipe.opt.reserved |= 4
    if args.no_cluster:
        pipe.opt.reserved |= 8
    if args.cluster_all:
        pipe.opt.reserved |= 16
    if args.wide_tiles:
        pipe.opt.reserved |= 32
    if args.full_barriers:
        pipe.opt.reserved |= 256
    pipe.opt.reserved |= args.flags
    if args.mix:
        run_mix(pipe, args)
        return
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
            ev = pipe.fit_events[0]
            fl = ev.flops
            if r > 0:
                times.append(ev.read()[2])
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
            prof = np.stack([ws[i * total + scal + 24: i * total + scal + 52] for i in range(min(n, 64))]).mean(0)
            names = ["chol:update", "chol:rest", "inv", "kx", "A+BMT+meanvar", "chol:diag", "quad/kl", "Gm+GA",
                     "GLS+adam", "Pm(-GA A^T)", "GKX", "-", "W", "S", "kgrads+adamZ", "-", "adam", "predict", "chol:panel",
                     "misc"]
            route = _lib.load().gapro_fit_route(m, args.d)
            if not args.force_staged and route in (0, 3):
                names = ["chol:update", "chol:rest", "inv", "s:fill", "s:A", "chol:diag", "post-strips", "s:B",
                         "GLS+adam", "s:meanvar", "s:lik", "s:scale+GLSacc", "s:GA", "tail", "kgrads+adamZ",
                         "s:GKX+PmAcc", "adam", "predict", "s:kgrad-zx (D = 6) / GKXTstore", "misc", "kg:loop", "kg:sums", "(diag:factor", "(diag:inverse",
                         "(diag:stores", "x25", "x26", "x27"]
            if route == 5 and not (args.flags & (1 << 20)):
                names = ["hypers", "chol+inv", "kx", "A+colsums", "lik", "At+GLS+Gm", "GB+GA+PmT", "zx grads", "sums+loss",
                         "adamLS", "W+S+Wzz", "adamZ+m+scal", "predict"] + ["-"] * 15
            from gapro_amd import _lib as _l
            if (route == 4 or args.cluster_all) and not args.no_cluster:
                names = ["kzz", "chol:diag(leader)", "chol:panel", "chol:trailing", "inverse", "kx", "fwd:colpart",
                         "-", "chol:flag", "quad+kl", "Gm+GA", "GLS+adamLS+GKX+Pm", "-", "-", "W", "S", "kgrads(fused)", "-",
                         "adam", "predict", "fwd:A", "fwd:B", "-", "-", "-", "-", "-", "-"]
            prof = prof[:25]  # slots 25 .. 27: start / end / CU of the workgroup (tools/fit_timeline.py)
            tot = prof[:22].sum()
            print("    phases (us per fit, share): " + "  ".join(
                "%s %.0f (%.0f%%)" % (nm, v / 100.0, 100 * v / tot) for nm, v in zip(names, prof) if v > 0), flush=True)


if __name__ == "__main__":
    main()
