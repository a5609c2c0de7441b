import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from gapro_amd._lib import Context
ctx = Context.get(0); lib = ctx.lib
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
src = torch.empty(2 * n_cu * 65536 + 65536, dtype=torch.float64, device="cuda").uniform_(-1, 1)
sink = torch.zeros(8, dtype=torch.float64, device="cuda")
for iters in (20000, 200000, 1000000, 3000000):
    for mode, nm in ((16 | 128, "sp lds+mfma random"), (16 | 7 | 128, "sp all random")):
        tf = C.c_double()
        ctx.check(ctx.dbg.gapro_debug_wgloop(ctx.handle, None, iters, mode, n_cu, C.c_void_p(src.data_ptr()), C.c_void_p(sink.data_ptr()), C.byref(tf)))
        print("iters %8d (%.2f s) %-22s %6.2f TFLOP/s" % (iters, 2048.0*16*8*iters*n_cu/(tf.value*1e12), nm, tf.value), flush=True)
