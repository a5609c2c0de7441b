#!/usr/bin/env python3
"""The staged kernel's product engines side by side (gapro_debug_product_bench): C = P^T Q per workgroup on its own
matrices, TFLOP/s chip-wide.  python tools/product_bench.py [--mp 256,384] [--wgs 256,64]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gapro_amd._lib import Context  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mp", default="256,384")
ap.add_argument("--wgs", default="256,64")
ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--lib", default="")
ap.add_argument("--check", action="store_true")
ap.add_argument("--engines", default="0,1,2")
ap.add_argument("--shapes", default="0,1,2,3,4,5")
args = ap.parse_args()
if args.lib:
    from gapro_amd import _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), args.lib)
ctx = Context.get(0)
names = {0: "per-wave 32x32", 1: "per-wave 64x64", 2: "workgroup-tiled"}
shapes = {0: "full", 1: "triangular P", 2: "lower output", 3: "k>=max(i,j)", 4: "k>=j", 5: "lower, k>=i"}
for mp in [int(x) for x in args.mp.split(",")]:
    for n_wg in [int(x) for x in args.wgs.split(",")]:
        base = torch.empty(n_wg * 3 * mp * mp, dtype=torch.float64, device="cuda").uniform_(-1, 1)
        for shape in [int(x) for x in args.shapes.split(',')]:
            slab = base.clone()
            v = slab.view(n_wg, 3, mp, mp)
            if shape == 1:    # P[k][i] = 0 for k > i
                v[:, 0] = torch.triu(v[:, 0])
            if shape in (3, 5):  # P[k][i] = 0 for k < i
                v[:, 0] = torch.tril(v[:, 0])
            if shape in (3, 4):  # Q[k][j] = 0 for k < j
                v[:, 1] = torch.tril(v[:, 1])
            frac = {0: 1.0, 1: 0.5, 2: 0.5, 3: 1.0 / 3, 4: 0.5, 5: 1.0 / 6}[shape]
            row = []
            for eng in [int(x) for x in args.engines.split(',')]:
                ms = C.c_float()
                ctx.check(ctx.dbg.gapro_debug_product_bench(ctx.handle, None, eng, shape, mp, args.reps, n_wg,
                                                            C.c_void_p(slab.data_ptr()), C.byref(ms)))
                us = 1e3 * ms.value / args.reps
                tf = frac * 2.0 * mp ** 3 * n_wg * args.reps / (ms.value * 1e-3) / 1e12
                row.append("%s %7.1f us %5.1f TF" % (names[eng], us, tf))
            print("Mp %3d, %3d WGs, %-13s: %s" % (mp, n_wg, shapes[shape], " | ".join(row)), flush=True)
            if args.check:  # every engine must leave the same bits in C (the last one ran last: rerun each and compare)
                outs = {}
                for eng in [int(x) for x in args.engines.split(',')]:
                    slab.view(n_wg, 3, mp, mp)[:, 2].zero_()
                    ms = C.c_float()
                    ctx.check(ctx.dbg.gapro_debug_product_bench(ctx.handle, None, eng, shape, mp, 1, n_wg,
                                                                C.c_void_p(slab.data_ptr()), C.byref(ms)))
                    c = slab.view(n_wg, 3, mp, mp)[0, 2].clone()
                    if shape in (2, 5):
                        c = torch.tril(c)
                    outs[eng] = c
                ref = outs[min(outs)]
                for eng, c in outs.items():
                    bad = (c != ref)
                    if bad.any():
                        idx = bad.nonzero()
                        print("    engine %d differs from engine %d in %d elements, rows %d..%d cols %d..%d, max abs %.3e"
                              % (eng, min(outs), int(bad.sum()), int(idx[:, 0].min()), int(idx[:, 0].max()),
                                 int(idx[:, 1].min()), int(idx[:, 1].max()), float((c - ref).abs().max())))
                    else:
                        print("    engine %d == engine %d bit for bit" % (eng, min(outs)))
