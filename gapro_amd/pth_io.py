"""NumPy-level wrapper of the native scene / label file reader and writer (csrc/pth_io.cc, gapro_pth_*).

`load_arrays` reads what `torch.load` returns for the reference's input files -- a NumPy array or a tuple of NumPy
arrays pickled by `torch.save` (gen_ps.py:45-46; written by prepare_data_inst.py:104 / prepare_superpoint.py:27) --
without unpickling: the C library walks the pickle and transcodes each payload straight into the destination array,
with the GIL released (ctypes).  It returns None for a file the native reader does not handle (tensor storages,
compressed members, object arrays, ...): the caller then uses torch.load.  `save_arrays` writes a file that torch.load
reads back as the same tuple of arrays (gen_ps.py:132).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib

_KINDS = {"f": "f", "i": "i", "u": "u", "b": "b"}


def native_enabled() -> bool:
    return os.environ.get("GAPRO_NATIVE_PTH", "1") != "0"


def load_arrays(path, out=None):
    """-> (arrays, is_sequence) or None.  arrays: list of NumPy arrays in file order.  `out(i, shape, dtype)` may
    return a preallocated C-contiguous array of exactly that shape / dtype for array i (e.g. a view of pinned staging
    memory); None or no callback = a fresh np.empty."""
    lib = _lib.load()
    h = C.c_void_p()
    rc = lib.gapro_pth_open(os.fsencode(path), C.byref(h))
    if rc == _lib.GAPRO_ERR_UNSUPPORTED:
        return None
    if rc != _lib.GAPRO_OK:
        raise OSError("gapro_pth_open(%s): %s" % (path, (lib.gapro_pth_last_error() or b"").decode()))
    try:
        n = lib.gapro_pth_count(h)
        info = _lib.PthArray()
        arrays = []
        for i in range(n):
            if lib.gapro_pth_info(h, i, C.byref(info)) != _lib.GAPRO_OK:
                raise OSError("gapro_pth_info failed")
            kind = chr(info.kind)
            dt = np.dtype(np.bool_) if kind == "b" else np.dtype("<%s%d" % (kind, info.itemsize))
            shape = tuple(int(info.shape[d]) for d in range(info.ndim))
            a = out(i, shape, dt) if out is not None else None
            if a is None:
                a = np.empty(shape, dtype=dt)
            elif a.shape != shape or a.dtype != dt or not a.flags.c_contiguous:
                raise ValueError("load_arrays: the destination for array %d does not match %r %s" % (i, shape, dt))
            rc = lib.gapro_pth_read(h, i, C.c_void_p(a.ctypes.data), int(info.nbytes))
            if rc != _lib.GAPRO_OK:
                raise OSError("gapro_pth_read(%s, %d): %s" % (path, i, (lib.gapro_pth_last_error() or b"").decode()))
            arrays.append(a)
        return arrays, bool(lib.gapro_pth_is_sequence(h))
    finally:
        lib.gapro_pth_close(h)


def load(path):
    """Drop-in for torch.load(path, weights_only=False) on the reference's NumPy-payload files; falls back to
    torch.load for anything else."""
    got = load_arrays(path) if native_enabled() else None
    if got is None:
        import torch

        return torch.load(path, weights_only=False)
    arrays, is_seq = got
    return tuple(arrays) if is_seq else arrays[0]


def save_arrays(path, arrays, as_tuple=True) -> bool:
    """Write `arrays` (NumPy, any of f/i/u/bool dtypes, up to 4-D) as a torch.load()-able file, atomically.  Returns
    False when the native writer does not take them (empty arrays, exotic dtypes): the caller then uses torch.save."""
    lib = _lib.load()
    arrs = [np.ascontiguousarray(a) for a in arrays]
    n = len(arrs)
    descs = (_lib.PthArray * n)()
    ptrs = (C.c_void_p * n)()
    for i, a in enumerate(arrs):
        k = a.dtype.kind
        if k not in _KINDS or a.ndim > 4 or a.size == 0 or a.dtype.byteorder == ">":
            return False
        descs[i].kind = ord(k)
        descs[i].itemsize = a.dtype.itemsize
        descs[i].ndim = a.ndim
        for d in range(4):
            descs[i].shape[d] = a.shape[d] if d < a.ndim else 1
        descs[i].nbytes = a.nbytes
        ptrs[i] = a.ctypes.data
    rc = lib.gapro_pth_write(os.fsencode(path), n, descs, ptrs, 1 if as_tuple else 0)
    if rc == _lib.GAPRO_ERR_UNSUPPORTED:
        return False
    if rc != _lib.GAPRO_OK:
        raise OSError("gapro_pth_write(%s): %s" % (path, (lib.gapro_pth_last_error() or b"").decode()))
    return True
