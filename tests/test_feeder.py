"""The native batch feeder (csrc/feeder.hip, gapro_amd/feeder.py) on the CPU: its host-only mode runs the same loader
and writer threads without a GPU, so what it produces can be held, byte for byte, to the Python mirror of
gen_ps.py:37-77 (read_scene + add_instance_info) and to torch.load of what it writes."""
import ctypes as C
import os

import numpy as np
import pytest
import torch


def _dataset(tmp_path, n=3, **kw):
    from gapro_amd.synth import make_scene, write_scannet_layout

    root = str(tmp_path / "dataset" / "scannetv2")
    scenes = []
    for i in range(n):
        sc = make_scene(seed=60 + i, n_points=3000 + 700 * i, n_objects=6, with_walls_json=(i == 0), obj_patch=25,
                        plane_patch=80, scan_name="scene%04d_00" % (800 + i), **kw)
        write_scannet_layout(sc, root, deepfeat_dir=str(tmp_path / "deep"))
        scenes.append(sc)
    return root, scenes


def _view(base, off, dtype, shape):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    return np.ctypeslib.as_array(C.cast(base + off, C.POINTER(C.c_uint8)), (n,)).view(dtype).reshape(shape)


@pytest.mark.parametrize("deep", [False, True])
def test_host_image_equals_the_python_mirror_of_the_reference_loader(tmp_path, deep):
    from gapro_amd.feeder import NativeFeeder
    from gapro_amd.gen_ps import add_instance_info, read_scene

    root, scenes = _dataset(tmp_path, 4)
    names = sorted(os.path.join(root, "train", s.scan_name + "_inst_nostuff.pth") for s in scenes)
    f = NativeFeeder(-1, 3, 64 << 20)
    try:
        f.submit(names, root, deep, str(tmp_path / "deep"))
        f.close()
        n, nbytes = f.poll(len(names), 256, 60000)
        assert n == len(names) and nbytes > 0
        bid, recs = f.upload(n, 0, 0)
        for r in recs:
            want = add_instance_info(read_scene(r.filename, root, deep, str(tmp_path / "deep")), "host")
            N, D = r.n_points, r.feat_dim
            assert r.status == 0 and D == (32 if deep else 6) and N == len(want["spp"])
            got = dict(coords_float=_view(r.host_image, r.off[0], np.float64, (N, 3)),
                       mask_feats=_view(r.host_image, r.off[1], np.float32, (N, D)),
                       spp=_view(r.host_image, r.off[2], np.int64, (N,)),
                       semantic_label=_view(r.host_image, r.off[3], np.float64, (N,)),
                       instance_label=_view(r.host_image, r.off[4], np.float64, (N,)))
            for k, v in got.items():  # the alignment included: np.dot(pts, A.T) bit for bit
                assert v.dtype == np.asarray(want[k]).dtype and np.array_equal(v, want[k]), k
            for k in ("instance_cls", "instance_box", "instance_box_volume"):
                v = getattr(r, k)
                assert v.dtype == want[k].dtype and np.array_equal(v, want[k]), k
            assert all(o % 256 == 0 for o in r.off)
        f.release_batch(bid)
    finally:
        f.destroy()


def test_scenes_come_out_in_order_within_a_small_staging_budget_and_bad_files_are_reported(tmp_path):
    """Eight scenes through two threads with room for about two of them in the staging pool: they are handed out in
    submission order, a tensor payload (not the NumPy tuple of prepare_data_inst.py:104) comes back UNSUPPORTED, a
    missing superpoint file as an I/O error, a scene without instances with n_instances == 0 -- each without
    disturbing its neighbours."""
    from gapro_amd import _lib
    from gapro_amd.feeder import NativeFeeder

    root, scenes = _dataset(tmp_path, 8)
    names = sorted(os.path.join(root, "train", s.scan_name + "_inst_nostuff.pth") for s in scenes)
    torch.save((torch.zeros(5, 3), torch.zeros(5, 3), torch.zeros(5), torch.zeros(5)), names[2])  # tensors
    os.remove(os.path.join(root, "superpoints", os.path.basename(names[4])[:12] + ".pth"))
    xyz, rgb, sem, inst = torch.load(names[6], weights_only=False)
    torch.save((xyz, rgb, sem, np.full_like(inst, -100.0)), names[6])
    f = NativeFeeder(-1, 2, 400 << 10)
    try:
        f.submit(names[:5], root)
        f.submit(names[5:], root)
        f.close()
        seen = []
        while True:
            n, _ = f.poll(1, 3, 60000)
            if n == 0:
                break
            bid, recs = f.upload(n, 0, 0)
            seen += [(r.filename, r.status, r.n_instances) for r in recs]
            f.release_batch(bid)
        assert [s[0] for s in seen] == names
        st = [s[1] for s in seen]
        assert st[2] == _lib.GAPRO_ERR_UNSUPPORTED and st[4] == _lib.GAPRO_ERR_IO
        assert [st[i] for i in (0, 1, 3, 5, 6, 7)] == [0] * 6
        assert seen[6][2] == 0 and all(seen[i][2] > 0 for i in (0, 1, 3, 5, 7))
    finally:
        f.destroy()


def test_export_writes_the_reference_5_tuple(tmp_path):
    from gapro_amd.feeder import NativeFeeder

    f = NativeFeeder(-1, 2, 16 << 20)
    try:
        keep, items = [], []
        for k, (n, s) in enumerate([(1000, 37), (5, 5), (70000, 1200)]):
            rng = np.random.default_rng(k)
            arrs = (rng.integers(-100, 19, n).astype(np.int32), rng.integers(-100, 50, n).astype(np.int32),
                    rng.random(n).astype(np.float32), rng.normal(size=s).astype(np.float32),
                    rng.random(s).astype(np.float32))
            keep.append(arrs)
            items.append((str(tmp_path / ("scene%04d_00.pth" % k)),) + tuple(a.ctypes.data for a in arrs) + (n, s))
        f.export(items)
        done, failed = f.export_wait(-1, 60000)
        assert (done, failed) == (3, 0)
        assert sorted(os.listdir(tmp_path)) == ["scene%04d_00.pth" % k for k in range(3)]  # no temporary left behind
        for k, arrs in enumerate(keep):
            tup = torch.load(str(tmp_path / ("scene%04d_00.pth" % k)), weights_only=False)  # scannetv2.py:46-48
            assert isinstance(tup, tuple) and len(tup) == 5
            for u, v in zip(tup, arrs):
                assert isinstance(u, np.ndarray) and u.dtype == v.dtype and np.array_equal(u, v)
        # a path that cannot be written is counted and named, the others are not affected
        f.export([(str(tmp_path / "no_such_dir" / "x.pth"),) + tuple(a.ctypes.data for a in keep[1]) + (5, 5)])
        done, failed = f.export_wait(-1, 60000)
        assert (done, failed) == (4, 1) and "no_such_dir" in f.export_errors(failed)[0]
    finally:
        f.destroy()


def test_dry_run_worker_goes_through_the_feeder(tmp_path, capsys):
    """`gen_ps --dry_run`: the host side of a worker with no GPU -- claim list, native loaders, native writer; the
    stand-in label files land beside the label folder, never in it."""
    from gapro_amd import gen_ps

    root, scenes = _dataset(tmp_path, 5)
    save = str(tmp_path / "labels")
    rc = gen_ps.main(["--save_folder", save, "--data_root", root, "--batch_scenes", "2", "--dry_run"])
    out = capsys.readouterr()
    assert rc == 0 and "5 scenes written, 0 skipped/failed" in out.out and "native feeder" in out.out
    assert os.listdir(save) == []
    files = sorted(os.listdir(save + ".DRY_RUN"))
    assert files == sorted(s.scan_name + ".pth" for s in scenes)
    for s in scenes:
        tup = torch.load(os.path.join(save + ".DRY_RUN", s.scan_name + ".pth"), weights_only=False)
        assert [len(a) for a in tup[:3]] == [s.n_points] * 3 and tup[0].dtype == np.int32


def test_poll_for_more_scenes_than_the_budget_holds_returns_what_is_loaded(tmp_path):
    """`poll(min_ready = batch, timeout = -1)` with a staging budget smaller than the batch waited for
    ever -- the loaders blocked on the budget, the poller on the count, and the space only comes back through the
    poller taking scenes.  Now the poll hands out what is loaded once no loader can make progress; every scene still
    arrives, in order."""
    import threading

    from gapro_amd.feeder import NativeFeeder

    root, scenes = _dataset(tmp_path, 8)
    names = sorted(os.path.join(root, "train", s.scan_name + "_inst_nostuff.pth") for s in scenes)
    f = NativeFeeder(-1, 3, 300 << 10)  # room for one or two of the eight scenes
    got, err = [], []

    def consume():
        try:
            while True:
                n, _ = f.poll(8, 8, -1)  # asks for the whole list at once, no time limit
                if n == 0:
                    break
                assert n < 8
                bid, recs = f.upload(n, 0, 0)
                got.extend(r.filename for r in recs)
                assert all(r.status == 0 for r in recs)
                f.release_batch(bid)
        except Exception as e:  # noqa: BLE001
            err.append(e)

    try:
        f.submit(names, root)
        f.close()
        t = threading.Thread(target=consume, daemon=True)
        t.start()
        t.join(60)
        assert not t.is_alive(), "gapro_feed_poll is stuck behind the staging budget"
        assert not err and got == names
    finally:
        f.destroy()


def test_dry_run_worker_finishes_with_a_budget_smaller_than_a_batch(tmp_path, monkeypatch, capsys):
    """The same through the driver (the case ADVICE r05 reproduced: GAPRO_FEED_BUDGET_MB=1, --batch_scenes 16)."""
    from gapro_amd import gen_ps

    root, scenes = _dataset(tmp_path, 12)
    monkeypatch.setenv("GAPRO_FEED_BUDGET_MB", "1")
    save = str(tmp_path / "labels")
    rc = gen_ps.main(["--save_folder", save, "--data_root", root, "--batch_scenes", "16", "--dry_run"])
    assert rc == 0 and "12 scenes written, 0 skipped/failed" in capsys.readouterr().out
    assert sorted(os.listdir(save + ".DRY_RUN")) == sorted(s.scan_name + ".pth" for s in scenes)


def test_export_wait_counts_a_contiguous_prefix(tmp_path):
    """n_done of gapro_feed_export_wait is what a caller frees device memory by: exports finished as a prefix of the
    submission order, not the number of finished writes (a small file queued behind a large one finishes first)."""
    from gapro_amd.feeder import NativeFeeder

    f = NativeFeeder(-1, 4, 256 << 20)
    try:
        rng = np.random.default_rng(0)
        big = tuple(rng.random(4_000_000).astype(t) for t in (np.int32, np.int32, np.float32)) + \
            (np.zeros(5, np.float32), np.zeros(5, np.float32))
        small = tuple(np.zeros(4, t) for t in (np.int32, np.int32, np.float32, np.float32, np.float32))
        items = [(str(tmp_path / "a.pth"),) + tuple(a.ctypes.data for a in big) + (4_000_000, 5)]
        items += [(str(tmp_path / ("s%d.pth" % k)),) + tuple(a.ctypes.data for a in small) + (4, 4) for k in range(6)]
        f.export(items)
        seen = []
        while True:
            done, failed = f.export_wait(0, 0)
            seen.append(done)
            if done == 7:
                break
            # whenever the count says k, files 0 .. k-1 are complete on disk: the big one first
            if done >= 1:
                assert os.path.exists(str(tmp_path / "a.pth"))
        assert failed == 0 and seen == sorted(seen)
        assert (f.export_wait(-1, 60000)) == (7, 0)
    finally:
        f.destroy()
