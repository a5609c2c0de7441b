#!/usr/bin/env python3
"""Measured matrix-core peak of the device (FP64 and FP32 16x16x4 MFMA, all CUs, dependent-free accumulator
chains, no memory traffic): the figure the fit kernels' roofline is priced against.

    python tools/mfma_peak.py
"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(device=0, iters=20000):
    import torch

    from gapro_amd._lib import Context

    ctx = Context.get(device)
    sink = torch.zeros(1, dtype=torch.float64, device="cuda:%d" % device)
    out = {}
    for kind, name in ((0, "f64_16x16x4"), (1, "f32_16x16x4")):
        best = 0.0
        for wps in (1, 2, 4):
            tf = C.c_double()
            ctx.check(ctx.lib.gapro_debug_mfma_peak(ctx.handle, None, kind, iters, wps, C.c_void_p(sink.data_ptr()),
                                                    C.byref(tf)))
            out["%s_wps%d" % (name, wps)] = tf.value
            best = max(best, tf.value)
        out[name] = best
    return out


if __name__ == "__main__":
    print(json.dumps(measure()))
