#!/usr/bin/env python3
"""Per-phase timeline of one Adam step of the LDS-staged fit kernel with the matrix pipe's busy share per phase
("Next round" item 1: the deliverable the next target is chosen from).

    python tools/phase_table.py [--sizes 256,384] [--fits 512] [--lib libgapro_hip_prof.so]

Runs `tools/bench_fit.py --profile` (the diagnostic library with in-kernel phase stamps: GAPRO_BUILD_PROFILE=1
gapro_amd/csrc/build.sh; built with -DGAPRO_PROFILE_SPLIT the three products of the merged backward phase are stamped
one by one, which re-introduces their barriers) and prices every phase:

    executed MFMA FLOP of the phase per Adam step  -- counted tile by tile from the contraction ranges the kernel uses
                                                       (padded M_p, triangular ranges rounded to the wave tile)
    matrix-pipe busy = executed FLOP / (phase time x 0.307 TFLOP/s)    -- 78.6 TFLOP/s / 256 CUs; with two workgroups
                                                       per CU the CU's pipe is busy for BOTH, i.e. twice this share

Prints a markdown table per size.  Nothing here is part of the product.
"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CU_PEAK = 78.6e12 / 256


def tiles(mp, ts):
    return [(i, j) for i in range(0, mp, ts) for j in range(0, mp, ts)]


def product_flops(mp, ts, kr, lower=False):
    """2 x tile area x contraction length, summed over the (lower) tiles of an mp x mp output."""
    f = 0
    for i0, j0 in tiles(mp, ts):
        if lower and j0 > i0:
            continue
        lo, hi = kr(i0, j0)
        f += 2 * ts * ts * max(0, min(hi, mp) - lo)
    return f


def phase_flops(mp):
    ts = 64 if (mp >= 320 and mp % 32 == 0) else 32  # the wave tile of the staged kernel at this size (D = 6)
    nb = mp // 16
    chol = sum(2 * 16 * 16 * 16 * kb * (nb - kb) for kb in range(nb))            # left-looking updates
    inv = sum(2 * 16 * 16 * 16 * (ii + 1) * 1 for k in range(nb) for ii in range(1, nb - k))  # block column chains
    return {
        "chol": chol, "inv": inv,
        "A+BMT+meanvar": product_flops(mp, ts, lambda i, j: (0, i + ts)) + product_flops(mp, ts, lambda i, j: (i, mp)),
        "Gm+GA": product_flops(mp, ts, lambda i, j: (0, i + ts)),
        "GLS+adam": product_flops(mp, ts, lambda i, j: (0, mp), lower=True),
        "Pm(-GA A^T)": product_flops(mp, ts, lambda i, j: (0, mp), lower=True),
        "GKX": product_flops(mp, ts, lambda i, j: (j, mp)),
        "W": product_flops(mp, ts, lambda i, j: (j, i + ts), lower=True),
        "S": product_flops(mp, ts, lambda i, j: (max(i, j), mp)),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="256,384")
    ap.add_argument("--fits", type=int, default=512)
    ap.add_argument("--lib", default="libgapro_hip_prof.so")
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    for m in [int(x) for x in args.sizes.split(",")]:
        cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_fit.py"), "--sizes", str(m), "--fits", str(args.fits),
               "--reps", "1", "--profile", "--iters", str(args.iters), "--lib", args.lib]
        env = dict(os.environ)
        txt = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env).stdout
        head = [ln for ln in txt.splitlines() if ln.startswith("M=")]
        ph = [ln for ln in txt.splitlines() if "phases (us per fit" in ln]
        if not head or not ph:
            print(txt[-800:])
            continue
        items = re.findall(r"([^\s(][^()]*?) (\d+) \((\d+)%\)", ph[0].split("share):")[1])
        us = {k.strip(): float(v) / args.iters for k, v, _ in items}  # per Adam step (prediction and set-up included pro rata)
        mp = (m + 15) // 16 * 16
        fl = phase_flops(mp)
        chol_us = sum(us.get(k, 0.0) for k in ("chol:update", "chol:rest", "chol:diag", "chol:panel"))
        rows = [("Cholesky (update + diagonal blocks + panel)", chol_us, fl["chol"]), ("triangular inverse", us.get("inv", 0), fl["inv"])]
        for k in ("kx", "A+BMT+meanvar", "quad/kl", "Gm+GA", "GLS+adam", "Pm(-GA A^T)", "GKX", "W", "S", "kgrads+adamZ",
                  "adam", "predict", "misc"):
            if k in us:
                rows.append((k, us[k], fl.get(k, 0)))
        tot = sum(r[1] for r in rows)
        per_cu = 2 if mp <= 256 and args.fits > 256 else 1
        print("\n### M = %d (M_p = %d), %d concurrent fits (%d per CU), %s\n" % (m, mp, args.fits, per_cu, head[0].split(":")[1].strip()))
        print("| phase | us per Adam step | share | executed MFMA MFLOP | matrix pipe busy (this workgroup's share of one CU) |")
        print("|---|---|---|---|---|")
        for name, t, f in rows:
            busy = f / (t * 1e-6 * CU_PEAK) if t > 0 and f else None
            print("| %s | %.0f | %.1f %% | %s | %s |" % (name, t, 100 * t / tot, "%.1f" % (f / 1e6) if f else "-",
                                                       "%.0f %%" % (100 * busy) if busy is not None else "-"))
        ftot = sum(r[2] for r in rows)
        print("| **step** | **%.0f** | 100 %% | %.1f | **%.0f %%**%s |" % (
            tot, ftot / 1e6, 100 * ftot / (tot * 1e-6 * CU_PEAK), " (x %d workgroups per CU)" % per_cu if per_cu > 1 else ""))


if __name__ == "__main__":
    main()
