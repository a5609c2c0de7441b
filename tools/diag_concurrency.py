#!/usr/bin/env python3
"""Do two streams run kernels concurrently on this box?  (torch.cuda._sleep = one-thread spin kernel)"""
import time
import torch

torch.cuda.init()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
cyc = int(2e8)
torch.cuda._sleep(1000)
torch.cuda.synchronize()
for label, streams in (("one stream x2", (s1, s1)), ("two streams", (s1, s2))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in streams:
        with torch.cuda.stream(s):
            torch.cuda._sleep(cyc)
    torch.cuda.synchronize()
    print("%-16s %.1f ms" % (label, 1e3 * (time.perf_counter() - t0)))
