#!/usr/bin/env python3
"""Dump the outputs of a fixed mixed batch of fits (sizes across all kernels) to a .npz, with the library named
by --lib; two dumps from two builds are compared with --compare.  Used to prove a kernel change bit-neutral."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--compare", nargs=2, default=None)
    args = ap.parse_args()
    if args.compare:
        a, b = np.load(args.compare[0]), np.load(args.compare[1])
        bad = [k for k in a.files if not np.array_equal(a[k], b[k], equal_nan=True)]
        print("arrays %d, differing %d %s" % (len(a.files), len(bad), bad[:8]))
        for k in bad[:12]:
            x, y = a[k].astype(np.float64), b[k].astype(np.float64)
            print("   %s: max abs diff %.3e, max rel %.3e" % (k, np.max(np.abs(x - y)), np.max(np.abs(x - y) / np.maximum(np.abs(x), 1e-30))))
        sys.exit(1 if bad else 0)
    from gapro_amd import _lib

    if args.lib:
        _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), args.lib)
    from gapro_amd.gaussian_process_utils import fit_gp_spp_batch
    from gapro_amd.synth import make_gp_problem

    feats_l, probs, base = [], [], 0
    for i, (m1, m2, t) in enumerate([(1, 2, 1), (7, 9, 5), (15, 17, 40), (20, 28, 33), (31, 33, 64), (40, 39, 7),
                                     (45, 50, 90), (60, 52, 31), (64, 64, 100), (70, 74, 12), (80, 96, 50),
                                     (130, 126, 20), (200, 190, 77), (72, 70, 9), (100, 60, 300),
                                     (88, 87, 40), (230, 250, 33), (301, 100, 450)]):
        f, b1, b2, it = make_gp_problem(300 + i, m1, m2, t, 6)
        feats_l.append(f)
        probs.append((b1 + base, b2 + base, it + base))
        base += len(f)
    out = fit_gp_spp_batch(np.concatenate(feats_l), probs, training_iter=50)
    f32, b1, b2, it = make_gp_problem(77, 50, 60, 30, 32, std=0.3)
    out += fit_gp_spp_batch(f32, [(b1, b2, it)], training_iter=50)
    np.savez(args.out, **{"f%d_%d" % (i, j): np.asarray(a) for i, o in enumerate(out) for j, a in enumerate(o)})
    print("wrote", args.out)


if __name__ == "__main__":
    main()
