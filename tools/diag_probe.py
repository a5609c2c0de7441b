#!/usr/bin/env python3
"""While a fit launch is running, how long does a tiny kernel on another stream take to complete?
   GAPRO_RESERVED_CUS=R python tools/diag_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_fit import build_mix  # noqa: E402
from gapro_amd.pipeline import Pipeline  # noqa: E402


class A:
    t, d = 32, 6


pipe = Pipeline(device=0, training_iter=50)
batch = build_mix("112:1024,80:1024", A)
p = pipe.fit_launch(*batch)
pipe.fit_collect(p)
torch.cuda.synchronize()
side = torch.cuda.Stream()
x = torch.zeros(1 << 20, device="cuda")
big = torch.zeros(150000 * 64, device="cuda")
for label, tensor in (("1M-element add", x), ("9.6M-element add", big)):
    t0 = time.perf_counter()
    p = pipe.fit_launch(*batch)
    time.sleep(0.01)  # the fit kernel is running now
    lat = []
    with torch.cuda.stream(side):
        for _ in range(5):
            t1 = time.perf_counter()
            tensor.add_(1.0)
            side.synchronize()
            lat.append(1e3 * (time.perf_counter() - t1))
    pipe.fit_collect(p)
    torch.cuda.synchronize()
    print("reserved %s: %s latencies while the fit runs (ms): %s ; fit launch total %.1f ms"
          % (os.environ.get("GAPRO_RESERVED_CUS", "default"), label, " ".join("%.2f" % v for v in lat),
             1e3 * (time.perf_counter() - t0)))
