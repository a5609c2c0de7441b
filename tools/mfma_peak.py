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
    sink = torch.ones(32, dtype=torch.float64, device="cuda:%d" % device)
    out = {}
    for kind, name in ((0, "f64_16x16x4"), (1, "f32_16x16x4"), (2, "f64_4x4x4_4b"), (3, "f64_4x4x4_4b_tile")):
        best = 0.0
        for wps in (1, 2, 4):
            tf = C.c_double()
            ctx.check(ctx.dbg.gapro_debug_mfma_peak(ctx.handle, None, kind, iters, wps, C.c_void_p(sink.data_ptr()),
                                                    C.byref(tf)))
            out["%s_wps%d" % (name, wps)] = tf.value
            best = max(best, tf.value)
        out[name] = best
    # Round 3: the chains above feed every instruction the SAME A / B registers, and in that form v_mfma_f64_16x16x4
    # tops out at ~48 TFLOP/s -- rounds 1 and 2 took that for the matrix cores' ceiling.  In a loop shaped like the fit
    # kernels' products (fragments re-read from LDS every k-step, 8 accumulator blocks per wave, 512-thread workgroups)
    # the same instruction sustains ~73: that is the ceiling the products are priced against in DESIGN.md.
    n_cu = torch.cuda.get_device_properties(device).multi_processor_count
    src = torch.ones(2 * n_cu * 65536 + 65536, dtype=torch.float64, device="cuda:%d" % device)
    for key, mode, per_cu in (("f64_16x16x4_lds_loop", 8, 2), ("f64_16x16x4_lds_loop_1wg", 8, 1),
                              ("f64_16x16x4_wg_tiled_loop", 16 | 7, 2), ("f64_4x4x4_4b_lds_loop", 0, 2)):
        tf = C.c_double()
        try:
            ctx.check(ctx.dbg.gapro_debug_wgloop(ctx.handle, None, 4000, mode, per_cu * n_cu, C.c_void_p(src.data_ptr()),
                                                 C.c_void_p(sink.data_ptr()), C.byref(tf)))
            out[key] = tf.value
        except Exception as e:  # noqa: BLE001 - diagnostic
            out[key] = repr(e)
    return out


def clocks(device=0, iters=20000):
    """FP64 MFMA rate and the s_memtime rate with 1/8 .. all of the wave slots of the benchmark filled (the dispatcher
    spreads the workgroups over all CUs: 1/4 of them = one wave per SIMD everywhere).  On the boxes measured s_memtime
    ticks at ~2.39 GHz whatever the load, so the 48 of 78.6 TFLOP/s is not a visible clock drop."""
    import torch

    from gapro_amd._lib import Context

    ctx = Context.get(device)
    sink = torch.zeros(4, dtype=torch.float64, device="cuda:%d" % device)
    n_cu = torch.cuda.get_device_properties(device).multi_processor_count
    out = {}
    for frac in (8, 4, 2, 1):
        tf, mhz = C.c_double(), C.c_double()
        ctx.check(ctx.dbg.gapro_debug_mfma_clock(ctx.handle, None, iters, 4, 4 * n_cu // frac,
                                                 C.c_void_p(sink.data_ptr()), C.byref(tf), C.byref(mhz)))
        out["workgroups_%d" % (4 * n_cu // frac)] = {"tflops": round(tf.value, 2), "s_memtime_mhz": round(mhz.value, 1)}
    return out


if __name__ == "__main__":
    print(json.dumps(measure()))
    if "--clock" in sys.argv:
        print(json.dumps(clocks()))
