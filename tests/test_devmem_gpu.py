"""The library's own device plumbing (csrc/devmem.hip, gapro_amd/devmem.py) and the torch-free gen_ps worker built on it
(round 6): the arena's stream-ordered reuse, copies / views / events, the pipeline on the native
backend bit for bit against the torch backend, and a worker process that finishes without ever importing torch."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_arena_reuses_a_block_for_its_own_stream_only():
    from gapro_amd.devmem import NativeBackend

    be = NativeBackend(0)
    s2 = be.new_stream()
    a = be.empty(3_000_000)
    pa = a.data_ptr()
    del a
    b = be.empty(2_900_000)  # same stream, fits the freed block (<= 2x): handed out again
    assert b.data_ptr() == pa
    with be.stream(s2):
        c = be.empty(2_900_000)  # another stream never gets it, free or not
    assert c.data_ptr() != pa
    del b
    with be.stream(s2):
        d = be.empty(2_900_000)
    assert d.data_ptr() not in (pa,)
    e = be.empty(100)  # far smaller than the free 3 MB block: not taken (<= 2x rule), a new small block
    assert e.data_ptr() != pa
    free, total = be.mem_get_info()
    assert 0 < free <= total


def test_copies_views_events():
    from gapro_amd.devmem import NativeBackend

    be = NativeBackend(0)
    rng = np.random.default_rng(0)
    src = rng.normal(size=(1000, 6)).astype(np.float32)
    d = be.from_numpy(src)
    assert d.shape == (1000, 6) and d.dtype == np.float32 and d.dim() == 2 and d.numel() == 6000
    np.testing.assert_array_equal(d.cpu(), src)
    raw = d.view(be.u8).view(-1)
    assert raw.shape == (24000,) and raw.data_ptr() == d.data_ptr()
    part = raw[4 * 6 * 10:4 * 6 * 20].view(be.f32).view(10, 6)
    np.testing.assert_array_equal(part.cpu(), src[10:20])
    pin = be.pinned(24000)
    e0, e1 = be.event(True), be.event(True)
    e0.record()
    pin[:24000].copy_(raw)  # device -> pinned host, asynchronous
    done = be.current_stream().record_event()
    e1.record()
    done.synchronize()
    np.testing.assert_array_equal(pin.numpy().view(np.float32).reshape(1000, 6), src)
    e1.synchronize()
    assert e0.elapsed_time(e1) >= 0.0 and done.query()
    z = be.zeros(77, be.f64)
    assert (z.cpu() == 0).all() and z.dtype == np.float64
    c = d.clone()
    be.current_stream().synchronize()
    np.testing.assert_array_equal(c.cpu(), src)
    h = pin.numpy()
    h[:8] = 255  # the numpy view writes through to the pinned block and keeps it alive
    del pin
    assert (h[:8] == 255).all()


def test_pipeline_on_the_native_backend_equals_the_torch_backend_bit_for_bit(golden):
    import torch

    from gapro_amd.pipeline import Pipeline, make_job

    kw = golden.api_inputs()
    args = [kw[k] for k in ("coords_float", "mask_feats", "spp", "instance_cls", "instance_box", "instance_box_volume",
                            "wall_box", "wall_box_volume")]
    opts = dict(instance_classes=18, ground_h=0.1, thresh_spp_occu=0.999)
    pt = Pipeline(device=0, training_iter=50)
    out_t = pt.run([make_job(*args, **opts)])[0]
    pn = Pipeline(device=0, training_iter=50, backend="native")
    assert pn.be.name == "native"
    out_n = pn.run([make_job(*args, **opts, backend=pn.be)])[0]
    for a, b in zip(out_t, out_n):
        np.testing.assert_array_equal(a.cpu().numpy(), b.cpu())
    # ... and through the software pipeline (three slots, several batches)
    jobs = [[make_job(*args, **opts, backend=pn.be) for _ in range(2)] for _ in range(4)]
    for outs in pn.run_stream(iter(jobs)):
        for o in outs:
            for a, b in zip(out_t, o):
                np.testing.assert_array_equal(a.cpu().numpy(), b.cpu())
    assert torch.cuda.is_available()


def test_a_worker_process_never_imports_torch_and_writes_the_same_files(tmp_path):
    """`gen_ps` on the native backend in a fresh process: torch is not in sys.modules when it ends; the label files
    are byte for byte those of the torch-plumbed worker (GAPRO_BACKEND=torch)."""
    from gapro_amd.synth import make_scene, write_scannet_layout

    root, scenes = str(tmp_path / "dataset" / "scannetv2"), []
    for i in range(3):
        sc = make_scene(seed=30 + i, n_points=4000, n_objects=8, with_walls_json=(i == 0), obj_patch=25,
                        plane_patch=80, scan_name="scene%04d_00" % (700 + i))
        write_scannet_layout(sc, root)
        scenes.append(sc)
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from gapro_amd import gen_ps\n"
            "rc = gen_ps.main(['--save_folder', sys.argv[1], '--data_root', %r, '--batch_scenes', '2'])\n"
            "print('TORCH_IMPORTED', 'torch' in sys.modules)\n"
            "sys.exit(rc)\n" % (ROOT, root))
    outs = {}
    for backend in ("native", "torch"):
        save = str(tmp_path / ("labels_" + backend))
        env = dict(os.environ, GAPRO_BACKEND=backend)
        r = subprocess.run([sys.executable, "-c", code, save], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        assert "3 scenes written, 0 skipped/failed" in r.stdout
        assert ("TORCH_IMPORTED False" if backend == "native" else "TORCH_IMPORTED True") in r.stdout, r.stdout
        assert ("the library's own arena" in r.stdout) == (backend == "native")
        outs[backend] = {f: open(os.path.join(save, f), "rb").read() for f in sorted(os.listdir(save))}
    assert sorted(outs["native"]) == sorted(s.scan_name + ".pth" for s in scenes)
    assert outs["native"] == outs["torch"]
