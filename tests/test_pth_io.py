"""The native scene / label file reader and writer (gapro_pth_*, csrc/pth_io.cc) against torch.load / torch.save: the
reference reads its inputs with torch.load (gen_ps.py:45-46) and writes its outputs with torch.save (gen_ps.py:132)."""
import io
import os
import pickle
import subprocess
import sys
import zipfile

import numpy as np
import pytest
import torch

from gapro_amd import _lib, pth_io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(a, b):
    assert type(a) is type(b) or (isinstance(a, np.ndarray) and isinstance(b, np.ndarray))
    assert a.dtype == b.dtype and a.shape == b.shape
    assert a.tobytes() == b.tobytes()  # byte equality: NaN payloads and -0.0 included


def _scene_tuple(rng, n):
    xyz = rng.normal(size=(n, 3)) * 3
    rgb = rng.integers(0, 256, size=(n, 3)) / 127.5 - 1
    sem = rng.integers(0, 20, n).astype(np.float64)
    ins = np.where(rng.random(n) < 0.2, -100.0, rng.integers(0, 30, n))
    return xyz, rgb, sem, ins


@pytest.mark.parametrize("n", [1, 7, 63, 64, 65, 1000, 40001])
def test_scene_tuple_written_by_todays_torch_is_read_byte_for_byte(tmp_path, n):
    """prepare_data_inst.py:104: torch.save((coords, colors, sem_labels, instance_labels)) of float64 arrays, and
    prepare_superpoint.py:27: torch.save(spp) of one int64 array, with the installed torch's defaults."""
    rng = np.random.default_rng(n)
    tup = _scene_tuple(rng, n)
    p = str(tmp_path / "s.pth")
    torch.save(tup, p)
    got = pth_io.load_arrays(p)
    assert got is not None and got[1] is True and len(got[0]) == 4
    for a, b in zip(torch.load(p, weights_only=False), got[0]):
        _same(a, b)
    spp = rng.integers(0, 5000, n).astype(np.int64)
    torch.save(spp, p)
    arrays, is_seq = pth_io.load_arrays(p)
    assert not is_seq and len(arrays) == 1
    _same(spp, arrays[0])
    assert isinstance(pth_io.load(p), np.ndarray)


def test_every_decoder_tier_agrees_on_every_byte_value_and_block_boundary(tmp_path):
    """All 256 byte values at every offset modulo 64, so that two-byte characters straddle every block boundary of the
    AVX-512 (64) and BMI2 (8) tiers; each tier the CPU has runs in its own process (the choice is made once)."""
    rng = np.random.default_rng(5)
    arrays = []
    for off in range(0, 70):
        a = np.concatenate([np.zeros(off, np.uint8), np.arange(256, dtype=np.uint8),
                            rng.integers(0, 256, 300 + off, dtype=np.uint8), np.full(130, 0xC3, np.uint8),
                            np.full(67, 0x7F, np.uint8), np.full(129, 0x80, np.uint8)])
        arrays.append(a)
    p = str(tmp_path / "b.pth")
    torch.save(tuple(arrays), p)
    want = [a.tobytes() for a in arrays]
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from gapro_amd import pth_io\n"
            "import hashlib\n"
            "arrs, _ = pth_io.load_arrays(%r)\n"
            "print(hashlib.sha256(b''.join(a.tobytes() for a in arrs)).hexdigest())\n" % (ROOT, p))
    import hashlib

    h = hashlib.sha256(b"".join(want)).hexdigest()
    for tier in ("scalar", "bmi2", "avx512"):  # an absent tier falls back to the next one: still must agree
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                           env=dict(os.environ, GAPRO_PTH_DECODER=tier), timeout=120)
        assert r.returncode == 0, r.stderr
        assert r.stdout.strip() == h, tier


def test_other_dtypes_shapes_and_protocols(tmp_path):
    rng = np.random.default_rng(1)
    tup = (rng.normal(size=(5, 4, 3)).astype(np.float32), rng.integers(-5, 5, (6, 2)).astype(np.int32),
           rng.integers(0, 255, 9).astype(np.uint8), rng.random(11) < 0.5, rng.integers(0, 9, 3).astype(np.int16),
           np.float64(rng.normal(size=(2, 2, 2, 2))))
    p = str(tmp_path / "m.pth")
    for proto in (2, 3, 4):  # protocol >= 3 stores the buffers as raw bytes (BINBYTES), 4 adds frames and MEMOIZE
        torch.save(tup, p, pickle_protocol=proto)
        got = pth_io.load_arrays(p)
        assert got is not None, proto
        for a, b in zip(tup, got[0]):
            _same(np.asarray(a), b)
    torch.save(list(tup[:2]), p)  # a list is a sequence too
    arrays, is_seq = pth_io.load_arrays(p)
    assert is_seq and len(arrays) == 2


def test_numpy_1_module_path_is_read(tmp_path):
    """Files written in the reference's environment (NumPy 1.x) name numpy.core.multiarray; NumPy 2 names
    numpy._core.multiarray.  Re-label today's pickle and rebuild the archive."""
    rng = np.random.default_rng(2)
    tup = _scene_tuple(rng, 500)
    p = str(tmp_path / "a.pth")
    torch.save(tup, p)
    with zipfile.ZipFile(p) as z:
        name = [n for n in z.namelist() if n.endswith("data.pkl")][0]
        blob = z.read(name)
    assert b"numpy._core.multiarray" in blob or b"numpy.core.multiarray" in blob
    old = blob.replace(b"numpy._core.multiarray", b"numpy.core.multiarray")
    q = str(tmp_path / "old.pth")
    with zipfile.ZipFile(q, "w", zipfile.ZIP_STORED) as z:
        z.writestr("archive/data.pkl", old)
        z.writestr("archive/version", "3\n")
    arrays, _ = pth_io.load_arrays(q)
    for a, b in zip(tup, arrays):
        _same(a, b)


def test_unsupported_files_fall_back_to_torch_load(tmp_path):
    p = str(tmp_path / "t.pth")
    t = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    torch.save(t, p)  # tensor storages (persistent ids): export_features.py writes deep features like this
    assert pth_io.load_arrays(p) is None
    assert torch.equal(pth_io.load(p), t)
    torch.save((np.arange(3), "a string"), p)  # not only arrays
    assert pth_io.load_arrays(p) is None
    assert pth_io.load(p)[1] == "a string"
    torch.save(np.asfortranarray(np.arange(12.0).reshape(3, 4)), p)
    assert pth_io.load_arrays(p) is None
    np.testing.assert_array_equal(pth_io.load(p), np.arange(12.0).reshape(3, 4))
    torch.save(np.arange(5).astype(">i4"), p)  # big-endian
    assert pth_io.load_arrays(p) is None
    torch.save(np.zeros(0), p)  # empty arrays are pickled through bytes(): declined
    assert pth_io.load_arrays(p) is None
    with zipfile.ZipFile(p, "w", zipfile.ZIP_DEFLATED) as z:  # a compressed member
        z.writestr("x/data.pkl", pickle.dumps(np.arange(100.0), protocol=2))
        z.writestr("x/version", "3\n")
    assert pth_io.load_arrays(p) is None
    torch.save(np.arange(4.0), p, _use_new_zipfile_serialization=False)  # the legacy (non-zip) format
    assert pth_io.load_arrays(p) is None
    np.testing.assert_array_equal(pth_io.load(p), np.arange(4.0))
    with pytest.raises(OSError):
        pth_io.load_arrays(str(tmp_path / "missing.pth"))


def test_damaged_payload_is_an_error_not_garbage(tmp_path):
    p = str(tmp_path / "d.pth")
    torch.save(np.frombuffer(bytes(range(256)) * 4, dtype=np.uint8).copy(), p)
    raw = bytearray(open(p, "rb").read())
    i = raw.index(b"\xc3\xbf")  # a two-byte character of the payload: break its continuation byte
    raw[i + 1] = 0x41
    open(p, "wb").write(bytes(raw))
    with pytest.raises(OSError):
        pth_io.load_arrays(p)


def test_malformed_utf8_is_rejected_by_every_decoder_tier(tmp_path):
    """the AVX-512 tier accepted the overlong lead bytes 0xC0 / 0xC1 (0xC0 0x80 came out as 0x80) where the
    scalar and BMI2 tiers -- and Python's own decoder, i.e. torch.load -- reject them.  Every tier this CPU has, in a
    process of its own: overlong leads, a lead beyond latin-1, a stray continuation byte, a lead with no continuation;
    in the first 64-byte block and in the scalar tail.  A tier the CPU lacks is skipped, not aliased."""
    a = np.concatenate([np.full(200, 0x85, np.uint8), np.arange(7, dtype=np.uint8)])  # 0x85 -> C2 85
    p = str(tmp_path / "ok.pth")
    torch.save(a, p)
    raw = bytearray(open(p, "rb").read())
    start = raw.find(bytes([0xC2, 0x85] * 50))
    assert start > 0
    cases = {}
    for name, pos, new in (("overlong C0 (block)", 10, bytes([0xC0, 0x85])), ("overlong C1 (block)", 20, bytes([0xC1, 0x85])),
                           ("lead C4 (block)", 30, bytes([0xC4, 0x85])), ("stray continuation (block)", 40, bytes([0x85, 0x85])),
                           ("lead without continuation (block)", 50, bytes([0xC2, 0x41])),
                           ("overlong C0 (tail)", 396, bytes([0xC0, 0x85])), ("overlong C1 (tail)", 398, bytes([0xC1, 0x85]))):
        bad = bytearray(raw)
        bad[start + pos:start + pos + 2] = new
        q = str(tmp_path / ("bad%d.pth" % len(cases)))
        open(q, "wb").write(bytes(bad))
        cases[name] = q
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from gapro_amd import _lib, pth_io\n"
            "lib = _lib.load()\n"
            "print('TIER', lib.gapro_pth_decoder().decode())\n"
            "assert pth_io.load_arrays(%r) is not None\n"
            "for name, q in %r.items():\n"
            "    try:\n"
            "        pth_io.load_arrays(q)\n"
            "        print('ACCEPTED', name)\n"
            "    except Exception as e:\n"
            "        print('rejected', name, type(e).__name__)\n" % (ROOT, p, cases))
    ran = set()
    for tier in ("scalar", "bmi2", "avx512"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                           env=dict(os.environ, GAPRO_PTH_DECODER=tier), timeout=120)
        assert r.returncode == 0, r.stderr
        got = [ln.split()[1] for ln in r.stdout.splitlines() if ln.startswith("TIER")][0]
        if got != tier:
            continue  # this CPU lacks the tier
        ran.add(tier)
        assert "ACCEPTED" not in r.stdout and r.stdout.count("rejected") == len(cases), (tier, r.stdout)
    assert "scalar" in ran


def test_preallocated_destinations(tmp_path):
    rng = np.random.default_rng(3)
    tup = _scene_tuple(rng, 300)
    p = str(tmp_path / "s.pth")
    torch.save(tup, p)
    bufs = [np.full(a.shape, 7, a.dtype) for a in tup]
    arrays, _ = pth_io.load_arrays(p, out=lambda i, shape, dt: bufs[i])
    assert all(a is b for a, b in zip(arrays, bufs))
    for a, b in zip(tup, bufs):
        _same(a, b)
    with pytest.raises(ValueError):
        pth_io.load_arrays(p, out=lambda i, shape, dt: np.empty(3))


def test_writer_output_is_what_torch_load_and_zipfile_expect(tmp_path):
    """gen_ps.py:132: the 5-tuple (int32, int32, float32, float32, float32) the ISBNet / SPFormer datasets unpack
    (ISBNet/isbnet/data/scannetv2.py:46-48)."""
    rng = np.random.default_rng(4)
    n, s = 5000, 77
    tup = (rng.integers(-100, 19, n).astype(np.int32), rng.integers(-100, 40, n).astype(np.int32),
           rng.random(n).astype(np.float32), rng.normal(size=s).astype(np.float32), rng.random(s).astype(np.float32))
    p = str(tmp_path / "scene0000_00.pth")
    assert pth_io.save_arrays(p, tup)
    assert os.listdir(tmp_path) == ["scene0000_00.pth"]  # atomic: no temporary file left
    back = torch.load(p, weights_only=False)
    assert isinstance(back, tuple) and len(back) == 5
    for a, b in zip(tup, back):
        _same(a, b)
    with zipfile.ZipFile(p) as z:
        assert z.testzip() is None  # CRCs
        names = z.namelist()
        assert names[0] == "scene0000_00/data.pkl" and "scene0000_00/version" in names
        blob = z.read(names[0])
    assert b"numpy.core.multiarray" in blob and b"numpy._core" not in blob  # importable by NumPy 1.x and 2.x
    got = pickle.loads(blob)  # a plain protocol-2 pickle
    for a, b in zip(tup, got):
        _same(a, b)
    arrays, is_seq = pth_io.load_arrays(p)  # and the native reader reads the native writer
    assert is_seq
    for a, b in zip(tup, arrays):
        _same(a, b)
    # other dtypes / ranks, a bare array, sizes that need BININT2 / BININT
    more = (rng.random((3, 4, 5)), rng.integers(0, 2, 70000).astype(np.uint8), rng.random(300) < 0.5,
            rng.integers(0, 9, (2, 2, 2, 2)).astype(np.int64))
    assert pth_io.save_arrays(p, more)
    for a, b in zip(more, torch.load(p, weights_only=False)):
        _same(a, b)
    assert pth_io.save_arrays(p, [more[0]], as_tuple=False)
    _same(more[0], torch.load(p, weights_only=False))
    assert pth_io.save_arrays(p, (np.zeros(0, np.float32),)) is False  # declined: the caller uses torch.save


def test_read_scene_uses_the_native_reader_and_equals_the_torch_path(tmp_path, monkeypatch):
    from gapro_amd.gen_ps import read_scene
    from gapro_amd.synth import make_scene, write_scannet_layout

    root = str(tmp_path / "dataset" / "scannetv2")
    sc = make_scene(seed=3, n_points=3000, n_objects=6, with_walls_json=True, obj_patch=25, plane_patch=80,
                    scan_name="scene0700_00")
    write_scannet_layout(sc, root)
    fn = os.path.join(root, "train", sc.scan_name + "_inst_nostuff.pth")
    calls = []
    real = torch.load
    monkeypatch.setattr(torch, "load", lambda *a, **k: calls.append(a) or real(*a, **k))
    a = read_scene(fn, root)
    assert calls == []  # neither file went through torch.load
    monkeypatch.setenv("GAPRO_NATIVE_PTH", "0")
    b = read_scene(fn, root)
    assert len(calls) == 2
    for k in ("coords_float", "mask_feats", "spp", "semantic_label", "instance_label"):
        _same(np.asarray(a[k]), np.asarray(b[k]))


def test_abi_status_codes():
    lib = _lib.load()
    import ctypes as C

    h = C.c_void_p()
    assert lib.gapro_pth_open(b"/nonexistent/file.pth", C.byref(h)) == _lib.GAPRO_ERR_IO
    assert b"nonexistent" in lib.gapro_pth_last_error()
    assert lib.gapro_pth_open(None, C.byref(h)) == -1


def test_native_host_passes_equal_the_numpy_mirrors():
    """The loader threads' native passes: default features (gen_ps.py:55) and GT instance boxes (gen_ps_utils.py:195-239,
    without the corner labels) -- bit-equal to the NumPy mirrors that are themselves pinned on the reference's golden
    outputs (tests/test_host_golden.py), including id gaps, -100 labels and a scene without instances."""
    from gapro_amd.gen_ps_utils import getInstanceInfo, getInstanceInfo_native
    from gapro_amd.synth import make_scene

    for seed in range(4):
        sc = make_scene(seed=seed, n_points=9000 + 3000 * seed, n_objects=4 + 3 * seed, with_walls_json=False,
                        obj_patch=25, plane_patch=80)
        xyz = sc.aligned_xyz()
        inst = sc.inst.copy()
        if seed == 2:  # gaps in the id range and unlabelled points
            inst[inst == 1] = -100.0
            inst[inst == 3] = inst.max() + 4
        a, b = getInstanceInfo(xyz, inst, sc.sem), getInstanceInfo_native(xyz, inst, sc.sem)
        assert a[0] == b[0]
        for x, y in zip(a[1:4], b[1:4]):
            _same(np.asarray(x), np.asarray(y))
        feats = np.empty((len(xyz), 6), np.float32)
        assert _lib.load().gapro_scene_default_feats(np.ascontiguousarray(sc.xyz).ctypes.data,
                                                     np.ascontiguousarray(sc.rgb).ctypes.data, len(xyz),
                                                     feats.ctypes.data) == 0
        _same(feats, np.concatenate([sc.xyz, sc.rgb], -1).astype(np.float32))
    assert getInstanceInfo_native(xyz, np.full(len(xyz), -100.0), sc.sem) is None
    assert getInstanceInfo(xyz, np.full(len(xyz), -100.0), sc.sem) is None


def test_every_encoder_and_crc_tier_agrees_with_the_scalar_one(monkeypatch):
    """The writer's two primitives: latin-1 -> UTF-8 in the AVX-512 VBMI2 / BMI2 / scalar tiers and
    the zip CRC-32 by PCLMULQDQ folding / slice-by-8 tables, on every byte value at every offset modulo 64 and on every
    length around the tiers' block sizes -- against Python's own encoder and zlib.crc32.  (A tier the CPU lacks falls
    back to the next one and must still agree.)"""
    import ctypes as C
    import zlib

    from gapro_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(11)
    cases = [b"", b"\x80", b"\x7f" * 63 + b"\xff"]
    for off in range(0, 70):
        cases.append(bytes(off) + bytes(range(256)) + rng.integers(0, 256, 300 + off, dtype=np.uint8).tobytes()
                     + b"\xc3" * 130 + b"\x7f" * 67 + b"\x80" * 129 + rng.integers(0, 128, 200, dtype=np.uint8).tobytes())
    for n in list(range(0, 200)) + [255, 256, 257, 1023, 4096 + 17, 100_003]:
        cases.append(rng.integers(0, 256, n, dtype=np.uint8).tobytes())
    seen_enc, seen_crc = set(), set()
    for tier in ("scalar", "bmi2", "avx512"):
        monkeypatch.setenv("GAPRO_PTH_ENCODER", tier)
        seen_enc.add(lib.gapro_pth_encoder().decode())
        for raw in cases:
            want = raw.decode("latin1").encode("utf-8")
            dst = C.create_string_buffer(2 * len(raw) + 64)
            n = lib.gapro_pth_encode_latin1(raw, len(raw), dst, len(dst))
            assert n == len(want) and dst.raw[:n] == want, (tier, len(raw))
    for tier in ("table", "clmul"):
        monkeypatch.setenv("GAPRO_PTH_CRC", tier)
        seen_crc.add(lib.gapro_pth_crc().decode())
        for raw in cases:
            assert lib.gapro_pth_crc32(raw, len(raw)) == zlib.crc32(raw), (tier, len(raw))
    assert "scalar" in seen_enc and "table" in seen_crc


def test_label_files_of_every_writer_tier_are_byte_equal(tmp_path, monkeypatch):
    from gapro_amd import pth_io

    rng = np.random.default_rng(3)
    n, s = 50_000, 900
    arrs = (rng.integers(-100, 19, n).astype(np.int32), rng.integers(-100, 40, n).astype(np.int32),
            rng.random(n).astype(np.float32), rng.normal(size=s).astype(np.float32), rng.random(s).astype(np.float32))
    blobs = []
    for enc, crc in (("scalar", "table"), ("bmi2", "clmul"), ("avx512", "clmul")):
        monkeypatch.setenv("GAPRO_PTH_ENCODER", enc)
        monkeypatch.setenv("GAPRO_PTH_CRC", crc)
        p = str(tmp_path / "scene0000_00.pth")
        assert pth_io.save_arrays(p, arrs)
        with open(p, "rb") as fh:
            blobs.append(fh.read())
        for a, b in zip(torch.load(p, weights_only=False), arrs):
            _same(a, b)
    assert blobs[0] == blobs[1] == blobs[2]
