"""Where a gen_ps worker's first second goes: wall-clock stamps of the start-up steps, in the order the worker takes them
(tools/startup_probe.py [--preload]: with --preload the torch-bundled HIP runtime is loaded first and libgapro_hip.so
initialises HIP BEFORE torch is imported -- the order the feeder needs to read files during the import)."""
import os
import sys
import time

T0 = time.time()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def stamp(what):
    print("%7.3f s  %s" % (time.time() - T0, what), flush=True)


def main():
    preload = "--preload" in sys.argv
    stamp("interpreter up")
    import numpy  # noqa: F401
    stamp("numpy")
    from gapro_amd import _lib
    if preload:
        import ctypes
        import importlib.util
        tdir = importlib.util.find_spec("torch").submodule_search_locations[0]
        ctypes.CDLL(os.path.join(tdir, "lib", "libamdhip64.so"), mode=ctypes.RTLD_GLOBAL)
        stamp("torch's libamdhip64.so preloaded")
        lib = _lib.load()
        stamp("libgapro_hip.so loaded")
        ctx = _lib.Context.get(0)
        stamp("gapro_ctx_create (HIP initialised by the library)")
    import torch
    stamp("import torch")
    ok = torch.cuda.is_available()
    stamp("torch.cuda.is_available() = %s" % ok)
    x = torch.zeros(1 << 20, device="cuda:0")
    torch.cuda.synchronize()
    stamp("first torch allocation + kernel")
    from gapro_amd.pipeline import Pipeline
    pipe = Pipeline(device=0, training_iter=50)
    stamp("Pipeline()")
    t = time.time()
    w = torch.zeros(int(25e9) // 8, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    stamp("25 GB workspace allocated and cleared (%.3f s)" % (time.time() - t))
    assert ok


if __name__ == "__main__":
    main()
