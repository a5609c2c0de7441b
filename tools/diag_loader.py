#!/usr/bin/env python3
"""Where the gen_ps driver's host time goes: loader-process throughput alone, and the host->device copy rate
out of a fresh shared-memory block (pageable) against a registered (pinned) one."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _drop(msg):
    from multiprocessing import shared_memory

    if msg["shm"]:
        s = shared_memory.SharedMemory(name=msg["shm"])
        s.close()
        s.unlink()
    return 1


def _read_and_drop(fn, root):
    from gapro_amd.gen_ps import _read_scene_shm

    t = time.time()
    msg = _read_scene_shm(fn, root)
    dt = time.time() - t
    _drop(msg)
    return dt


def main():
    import multiprocessing as mp

    import numpy as np

    from gapro_amd.gen_ps import _loader_init
    from gapro_amd.synth import make_scene, write_scannet_layout

    root = tempfile.mkdtemp(prefix="gapro_dl_")
    data = os.path.join(root, "d")
    for i in range(8):
        write_scannet_layout(make_scene(seed=i, n_points=150000, n_objects=25, with_walls_json=False,
                                        scan_name="scene%04d_00" % i), data)
    fns = [os.path.join(data, "train", "scene%04d_00_inst_nostuff.pth" % (i % 8)) for i in range(1024)]
    for n in (8, 16, 32, 64):
        pool = mp.get_context("spawn").Pool(n, initializer=_loader_init)
        pool.starmap(_read_and_drop, [(f, data) for f in fns[:2 * n]])  # warm
        t = time.time()
        dts = pool.starmap(_read_and_drop, [(f, data) for f in fns], chunksize=1)
        dt = time.time() - t
        print("loaders %2d: %.0f scenes/s, %.1f ms per scene inside a loader" % (n, len(fns) / dt, 1e3 * np.mean(dts)),
              flush=True)
        pool.close()
        pool.join()
    import torch
    from multiprocessing import shared_memory

    dev = torch.device("cuda", 0)
    nbytes = 12 << 20
    d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    rt = torch.cuda.cudart()
    for mode in ("pageable fresh block", "registered block"):
        ts = []
        for _ in range(20):
            shm = shared_memory.SharedMemory(create=True, size=nbytes)
            v = np.ndarray((nbytes,), np.uint8, buffer=shm.buf)
            v[:] = 1
            t = time.time()
            if mode.startswith("registered"):
                rc = rt.cudaHostRegister(v.ctypes.data, nbytes, 0)
                t1 = time.time()
                h = torch.from_numpy(v)
                d.copy_(h, non_blocking=True)
                torch.cuda.synchronize()
                t2 = time.time()
                rt.cudaHostUnregister(v.ctypes.data)
                ts.append((t1 - t, t2 - t1, time.time() - t2, h.is_pinned()))
                del h
            else:
                d.copy_(torch.from_numpy(v))
                torch.cuda.synchronize()
                ts.append((time.time() - t,))
            del v
            shm.close()
            shm.unlink()
        print(mode, ["%.2f ms" % (1e3 * np.mean([x[i] for x in ts[2:]])) for i in range(len(ts[0]) - (len(ts[0]) == 4))],
              ts[-1][-1] if len(ts[0]) == 4 else "")
    import shutil

    shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
