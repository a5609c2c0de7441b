#!/usr/bin/env python3
"""Which workspace matrix of a fit differs first between two settings of GAPRO_FIT_FLAGS (debug)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gapro_amd import _lib, gen_ps_utils
from gapro_amd._lib import FitDesc
from gapro_amd.synth import make_gp_problem
m1, m2, t, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
flags = int(sys.argv[5])
f, b1, b2, it = make_gp_problem(314, m1, m2, t, 6)
names = "LS LST MLS VLS GLS L LT LI U KX A AT BM BMT GA GKX GKXT".split()
res = {}
for fl in (0, flags):
    pipe = gen_ps_utils._pipeline(torch.device("cuda:0"), iters)
    pipe.opt.reserved = fl
    descs = (FitDesc * 1)()
    d = descs[0]; d.m1, d.m2, d.t, d.idx_offset, d.out_offset = m1, m2, t, 0, 0
    idx = np.concatenate([b1, b2, it]).astype(np.int32)
    out = pipe.fit_descs(torch.from_numpy(f).cuda(), descs, 1, idx, t, keep_debug=True, raise_on_failure=False)
    lay = (C.c_int64 * 8)()
    _lib.load().gapro_fit_workspace_layout(m1 + m2, t, 6, C.cast(lay, C.c_void_p))
    Mp = int(lay[0])
    ws = out["workspace"].cpu().numpy()
    res[fl] = (ws, Mp, out["mu"].copy())
a, Mp, mu_a = res[0]; b, _, mu_b = res[flags]
print("Mp", Mp, "mu diff", np.abs(mu_a - mu_b).max())
for k, nm in enumerate(names):
    A = a[k * Mp * Mp:(k + 1) * Mp * Mp].reshape(Mp, Mp); B = b[k * Mp * Mp:(k + 1) * Mp * Mp].reshape(Mp, Mp)
    bad = A != B
    if bad.any():
        ii = np.argwhere(bad)
        print("%-5s differs in %6d elements: rows %d..%d cols %d..%d max abs %.3e" % (nm, bad.sum(), ii[:,0].min(), ii[:,0].max(), ii[:,1].min(), ii[:,1].max(), np.nanmax(np.abs(A - B))))
    else:
        print("%-5s equal" % nm)
