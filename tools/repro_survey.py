#!/usr/bin/env python3
"""How many GP fits of the headline workload are numerically soft?  The first --scenes scenes of bench.py's train-split
stream (same seeds, same generator) through the pipeline's partition and schedule stages, then every fit twice: as the
product runs it, and with the jitter on K_ZZ's diagonal scaled by (1 + 1e-11) (Pipeline.reproducibility_probe).
Prints the histogram of the sigma^2 / p movements and the count beyond REPRO_SOFT.  Run on the GPU box.

    python tools/repro_survey.py [--scenes 128]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=128)
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    import concurrent.futures as cf

    import bench
    from gapro_amd.pipeline import REPRO_SOFT, Pipeline, make_job

    with cf.ProcessPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as ex:
        kws = list(ex.map(bench._scene_task, [(s, 150000, 6, "stream", None) for s in range(args.scenes)]))
    pipe = Pipeline(device=0, training_iter=50)
    dv_all, dp_all, m_all = [], [], []
    for b0 in range(0, len(kws), args.batch):
        jobs = [make_job(**{k: v for k, v in kw.items()}) for kw in kws[b0:b0 + args.batch]]
        state = pipe._partition(jobs)
        pipe._schedule_all(state)
        n = state["n_fits"]
        if not n:
            continue
        dv, dp = pipe.reproducibility_probe(state["feats_spp_all"], state["descs"], n, state["h_idx"], state["n_out"])
        dv_all.append(dv)
        dp_all.append(dp)
        m_all.append(np.array([state["descs"][k].m1 + state["descs"][k].m2 for k in range(n)]))
        for j in state["jobs"]:
            pipe.lib.gapro_schedule_free(j.schedule)
    dv, dp, m = np.concatenate(dv_all), np.concatenate(dp_all), np.concatenate(m_all)
    mv = np.maximum(dv, dp)
    print("%d fits of %d scenes (train-split stream, seeds 0 .. %d)" % (len(dv), args.scenes, args.scenes - 1))
    edges = [0, 1e-7, 3e-7, 1e-6, 3e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1.0]
    for lo, hi in zip(edges[:-1], edges[1:]):
        sel = (mv >= lo) & (mv < hi)
        print("  moved by [%.0e, %.0e): %6d fits%s" % (lo, hi, int(sel.sum()),
                                                      ("   M = " + ", ".join(str(int(x)) for x in sorted(m[sel])[:12])
                                                       + (" ..." if sel.sum() > 12 else "")) if hi > 1e-5 and sel.any() else ""))
    soft = mv > REPRO_SOFT
    print("soft (beyond REPRO_SOFT = %.0e): %d of %d fits = %.3f %%; largest movement dv %.2e dp %.2e"
          % (REPRO_SOFT, int(soft.sum()), len(mv), 100.0 * soft.mean(), dv.max(), dp.max()))


if __name__ == "__main__":
    main()
