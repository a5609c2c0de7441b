#!/usr/bin/env python3
"""Timeline of one fit launch of the bench's stream workload (run on the GPU box, diagnostic build).

    GAPRO_BUILD_PROFILE=1 bash gapro_amd/csrc/build.sh
    python tools/fit_timeline.py [--fit-m tools/data/stream_fit_m.npy] [--t 32] [--bins 24]

Every workgroup of the diagnostic build leaves its start / end time (100 MHz wall clock) and the CU it ran on in the
fit's workspace.  The tool launches the M mix of one bench step (tools/data/stream_fit_m.npy: the M of every fit of a
step, dumped by bench.py with GAPRO_DUMP_FIT_M) and prints, per kernel route, when its workgroups ran and how many CUs
they held, i.e. where the launch's CU-seconds go.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gapro_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fit-m", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "train_split_fit_m.npy"),
                    help="M of every fit of one launch (bench.py GAPRO_DUMP_FIT_M); default: the last step of the train-split "
                         "workload (round 4); data/stream_fit_m.npy is round 2's 64-seed step")
    ap.add_argument("--t", type=int, default=32)
    ap.add_argument("--d", type=int, default=6)
    ap.add_argument("--bins", type=int, default=24)
    ap.add_argument("--min-m", type=int, default=0, help="only fits with M >= this")
    ap.add_argument("--max-m", type=int, default=1 << 30)
    ap.add_argument("--flags", type=int, default=0, help="gapro_fit_options.reserved debug bits")
    ap.add_argument("--lib", default="libgapro_hip_prof.so", help="library file inside gapro_amd/; with the product "
                    "library (libgapro_hip.so) only the launch times by HIP events are printed (A/B runs)")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--no-raise", action="store_true", help="timing experiments that break the arithmetic")
    ap.add_argument("--dump", default="", help="save the per-fit rows (M, route, start, end, HW_ID, G, start ms, end ms) as .npy")
    args = ap.parse_args()
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), args.lib)
    prof = "prof" in args.lib
    from gapro_amd._lib import FitDesc
    from gapro_amd.pipeline import Pipeline
    from gapro_amd.synth import make_gp_problem

    if ".npz:" in args.fit_m:  # bench.py's per-step dump (GAPRO_DUMP_FIT_M): file.npz:step3
        fn, key = args.fit_m.rsplit(":", 1)
        ms = np.load(fn)[key].astype(np.int64)
    else:
        ms = np.load(args.fit_m).astype(np.int64)
    ms = ms[(ms >= args.min_m) & (ms <= args.max_m)]
    rng = np.random.default_rng(0)
    rng.shuffle(ms)
    probs, feats_l, base = {}, [], 0
    for m in np.unique(ms):
        f, b1, b2, it = make_gp_problem(int(m) % 7, int(m) // 2, int(m) - int(m) // 2, args.t, args.d)
        feats_l.append(f)
        probs[int(m)] = (b1 + base, b2 + base, it + base)
        base += len(f)
    feats = torch.from_numpy(np.concatenate(feats_l)).cuda()
    n = len(ms)
    descs = (FitDesc * n)()
    idx, io, oo = [], 0, 0
    for i, m in enumerate(ms):
        b1, b2, it = probs[int(m)]
        d = descs[i]
        d.m1, d.m2, d.t = len(b1), len(b2), len(it)
        d.idx_offset, d.out_offset = io, oo
        idx += [b1, b2, it]
        io += len(b1) + len(b2) + len(it)
        oo += len(it)
    h_idx = np.concatenate(idx).astype(np.int32)
    pipe = Pipeline(device=0, training_iter=50)
    pipe.opt.reserved |= args.flags
    if args.no_raise:
        pipe.opt.psd_retries = 0
    pipe.profile_fit = True
    spans = []
    for rep in range(2 if prof else args.reps + 1):
        pipe.fit_events = []
        res = pipe.fit_descs(feats, descs, n, h_idx, oo, keep_debug=prof, raise_on_failure=not args.no_raise)
        torch.cuda.synchronize()
        ev = pipe.fit_events[0].read()
        spans.append(ev[2])
    print("launch by HIP events (staged end, strip end, span, small end, cluster end):", ev)
    if not prof:
        print("span over %d launches: median %.1f ms  min %.1f  (%s)" % (args.reps, float(np.median(spans[1:])), min(spans[1:]),
                                                                       " ".join("%.0f" % v for v in spans[1:])))
        return
    lib = _lib.load()
    ws = res["workspace"].cpu().numpy()
    dd = res["descs"] if "descs" in res else descs
    lay = (C.c_int64 * 8)()
    rows = []
    for i in range(n):
        m = int(ms[i])
        lib.gapro_fit_workspace_layout(m, args.t, args.d, C.cast(lay, C.c_void_p))
        o = int(dd[i].ws_offset) + int(lay[6]) + 24
        r = int(lib.gapro_fit_route(m, args.d))
        if args.flags & 8 and r == 4:
            r = 1
        g = 1
        if r == 4:  # the default policy of gapro_cluster_size (fit_layout.h)
            unit = float(os.environ.get("GAPRO_CLUSTER_UNIT", 384))
            work = float(lay[0]) ** 3 / unit ** 3
            if os.environ.get("GAPRO_CLUSTER_ROUND") == "ceil":
                g = int(min(32, max(1, np.ceil(work))))
            else:
                while g < 32 and g < work:
                    g *= 2
        rows.append((m, r, ws[o + 25], ws[o + 26], int(ws[o + 27]), max(g, 1)))
    a = np.array(rows, dtype=np.float64)
    t0 = a[:, 2].min()
    st, en = (a[:, 2] - t0) / 1e5, (a[:, 3] - t0) / 1e5  # ms
    span = en.max()
    names = {0: "strip", 1: "staged", 2: "generic", 3: "small", 4: "cluster", 5: "wave"}
    print("launch span by workgroup stamps: %.1f ms, %d fits, %d distinct CU ids" % (span, n, len(np.unique(a[:, 4]))))
    tot_cu_ms = 0.0
    for r in sorted(set(a[:, 1].astype(int))):
        k = a[:, 1] == r
        dur = (en[k] - st[k])
        cu_ms = float((dur * a[k, 5]).sum())
        tot_cu_ms += cu_ms
        mm = a[k, 0]
        fl = float((50 * (8.33 * mm**3 + 12.0 * args.d * mm * mm) + mm**3 / 3.0).sum())
        print("  %-8s fits %5d  first start %7.1f  last end %7.1f  workgroup-ms %9.0f (= %.1f ms x 256)  "
              "%.2f TFLOP  longest fit %.1f ms" % (names[r], k.sum(), st[k].min(), en[k].max(), cu_ms, cu_ms / 256, fl / 1e12,
                                                 dur.max()))
    print("  sum of workgroup-ms / 256 = %.1f ms (a CU hosts 2 workgroups of the small / staged<4> kernels)" % (tot_cu_ms / 256))
    # CU-time with every workgroup weighted by the share of a CU it holds: strip and the one-per-CU staged fits a whole
    # CU, the two-per-CU staged (M_p <= 256) and small fits half, a cluster fit its G members (the leader's interval: the
    # members are resident from the cluster's first barrier to its last), a wave-per-fit fit 1/8 (M_p = 16), 1/4 (32) or
    # 1/3 (48: three fit one CU's LDS).  Against 256 CUs x the span this is the occupancy the launch really has --
    # the per-CU busy share at the end of this report cannot see cluster members other than leaders.
    mp = np.array([float(lib.gapro_fit_padded_m(int(m), args.d)) for m in a[:, 0]])
    w = np.ones(len(a))
    rr = a[:, 1].astype(int)
    w[(rr == 1) & (mp <= 256)] = 0.5
    w[rr == 3] = 0.5
    w[rr == 4] = a[rr == 4, 5]
    w[(rr == 5) & (mp <= 16)] = 1.0 / 8
    w[(rr == 5) & (mp == 32)] = 1.0 / 4
    w[(rr == 5) & (mp == 48)] = 1.0 / 3
    cu_time = float(((en - st) * w).sum())
    print("  CU-time (workgroup-ms weighted by the share of a CU each holds): %.0f CU-ms = %.1f ms x 256; span %.1f ms -> "
          "mean CU occupancy %.2f" % (cu_time, cu_time / 256, span, cu_time / (256 * span)))
    for r in sorted(set(rr)):
        k = rr == r
        print("    %-8s %5.1f %% of the CU-time" % (names[r], 100 * float(((en - st) * w)[k].sum()) / cu_time))
    edges = np.linspace(0, span, args.bins + 1)
    print("  active workgroups (x G for cluster fits) at the bin centres:")
    print("  t[ms]   " + " ".join("%8s" % names[r] for r in sorted(names)))
    for b in range(args.bins):
        tc = 0.5 * (edges[b] + edges[b + 1])
        act = (st <= tc) & (en > tc)
        print("  %6.0f  " % tc + " ".join("%8d" % int(a[act & (a[:, 1] == r), 5].sum()) for r in sorted(names)))
    # workgroups per XCD (HW_REG_XCC_ID) at the bin centres: a kernel's workgroups go to the XCDs round-robin
    xcd = (a[:, 4].astype(int) >> 16) & 15
    print("  active workgroups per XCD (all kernels, leaders only for cluster fits):")
    for b in range(args.bins):
        tc = 0.5 * (edges[b] + edges[b + 1])
        act = (st <= tc) & (en > tc)
        print("  %6.0f  " % tc + " ".join("%4d" % int((act & (xcd == x)).sum()) for x in range(8)))
    # per-CU busy share (leaders only for cluster fits); CU = (XCC_ID, SE_ID, SH_ID, CU_ID) of HW_REG_HW_ID
    cu = a[:, 4].astype(int) & 0xFFF00
    print("  CUs with at least one workgroup at the bin centres (of %d seen in all): " % len(np.unique(cu)) + " ".join(
        "%d" % len(np.unique(cu[(st <= 0.5 * (edges[b] + edges[b + 1])) & (en > 0.5 * (edges[b] + edges[b + 1]))]))
        for b in range(args.bins)))
    busy = []
    for c in np.unique(cu):
        k = cu == c
        iv = sorted(zip(st[k], en[k]))
        tot, cur_s, cur_e = 0.0, None, None
        for s_, e_ in iv:
            if cur_e is None or s_ > cur_e:
                if cur_e is not None:
                    tot += cur_e - cur_s
                cur_s, cur_e = s_, e_
            else:
                cur_e = max(cur_e, e_)
        tot += cur_e - cur_s
        busy.append(tot / span)
    busy = np.array(busy)
    # open intervals on one CU at once: a CU holds at most two of these workgroups, so more means that workgroups were
    # suspended between their start and end stamps (the hardware scheduler's time slicing of queues of one priority)
    over = []
    for c in np.unique(cu):
        k = cu == c
        ev = sorted([(s_, 1) for s_ in st[k]] + [(e_, -1) for e_ in en[k]], key=lambda t: (t[0], t[1]))
        cur = mx = 0
        for _, d_ in ev:
            cur += d_
            mx = max(mx, cur)
        over.append(mx)
    over = np.array(over)
    print("  most workgroups open at once on one CU: max %d, CUs with more than 2: %d of %d" % (over.max(), int((over > 2).sum()), len(over)))
    if args.dump:
        np.save(args.dump, np.column_stack([a, st, en]))
    print("  CU busy share (union of its workgroups' intervals, cluster members other than leaders not seen): "
          "mean %.2f  min %.2f  p10 %.2f" % (busy.mean(), busy.min(), np.percentile(busy, 10)))


if __name__ == "__main__":
    main()
