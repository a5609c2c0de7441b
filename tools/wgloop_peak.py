#!/usr/bin/env python3
"""The chunk loop of gemm_wg piece by piece (gapro_debug_wgloop): TFLOP/s with one / two workgroups per CU."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gapro_amd._lib import Context  # noqa: E402

ctx = Context.get(0)
lib = ctx.lib
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
src = torch.ones(2 * n_cu * 65536 + 65536, dtype=torch.float64, device="cuda")
sink = torch.zeros(8, dtype=torch.float64, device="cuda")
names = {0: "lds reads + mfma", 1: "+ barrier", 3: "+ barrier + lds stores", 7: "+ barrier + stores + global loads",
         4: "+ global loads only", 2: "+ lds stores only"}
src.uniform_(-1.0, 1.0)
extra = [(8 | 128, "16x16x4", "lds reads + mfma, RANDOM operands"), (16 | 128, "16x16x4 sp", "lds reads + mfma, RANDOM operands"),
         (16 | 7 | 128, "16x16x4 sp", "all, RANDOM operands"), (0 | 128, "4x4x4_4b", "lds reads + mfma, RANDOM operands"),
(16 | 7 | 32, "16x16x4 sp", "all, stores independent of loads"), (16 | 7 | 64, "16x16x4 sp", "all, loads at the end"),
         (16 | 5, "16x16x4 sp", "barrier + global loads (no stores)"), (16 | 6, "16x16x4 sp", "stores + global loads (no barrier)")]
for per_cu in (1, 2):
    for m, f, nm in extra:
        tf = C.c_double()
        ctx.check(ctx.dbg.gapro_debug_wgloop(ctx.handle, None, 20000, m, per_cu * n_cu, C.c_void_p(src.data_ptr()),
                                         C.c_void_p(sink.data_ptr()), C.byref(tf)))
        print("%d WG/CU  %-9s %-36s %6.2f TFLOP/s" % (per_cu, f, nm, tf.value))
    for form in (8, 16):
        for mode in (0, 1, 2, 3, 4, 7):
            tf = C.c_double()
            ctx.check(ctx.dbg.gapro_debug_wgloop(ctx.handle, None, 20000, mode | form, per_cu * n_cu,
                                             C.c_void_p(src.data_ptr()), C.c_void_p(sink.data_ptr()), C.byref(tf)))
            print("%d WG/CU  %-9s %-36s %6.2f TFLOP/s" % (per_cu, {0: "4x4x4_4b", 8: "16x16x4", 16: "16x16x4 sp"}[form], names[mode], tf.value))
