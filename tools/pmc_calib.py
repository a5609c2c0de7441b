#!/usr/bin/env python3
"""Launch the known-byte-count streaming kernels (gapro_debug_stream) so that a rocprofv3 --pmc pass over
this script calibrates FETCH_SIZE / WRITE_SIZE for one-double-per-lane accesses (the fit kernel's width).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o calib_fetch --output-format csv -- python3 tools/pmc_calib.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gapro_amd._lib import Context  # noqa: E402

n = 1 << 26  # 64 Mi doubles = 512 MiB per pass: past the 256 MiB Infinity Cache
ctx = Context(0)
src = torch.ones(n, dtype=torch.float64, device="cuda")
dst = torch.zeros(n, dtype=torch.float64, device="cuda")
for mode in (0, 1, 0, 1):
    ctx.check(ctx.dbg.gapro_debug_stream(ctx.handle, None, n, C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()),
                                         mode))
torch.cuda.synchronize()
print("bytes_read_per_launch", 8 * n, "bytes_written_per_launch(mode1)", 8 * n)
