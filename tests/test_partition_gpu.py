"""HIP scene-partition kernels vs the oracle (bit-exact: integers, bit masks, pooled float32 means)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_partition(kw, thresh=0.999, spp_range_cap=None):
    import torch
    from gapro_amd.pipeline import Pipeline, make_job

    pipe = Pipeline(device=0, training_iter=0, spp_range_cap=spp_range_cap)
    job = make_job(kw["coords_float"], kw["mask_feats"], kw["spp"], kw["instance_cls"], kw["instance_box"],
                   kw["instance_box_volume"], kw["wall_box"], kw["wall_box_volume"], 18, 0.1, thresh)
    pipe._prepare(job)  # launches gapro_partition_prepare_async, one sync, checks the header status
    feats_spp = torch.empty((job.n_spps, job.feats.shape[1]), dtype=torch.float32, device=pipe.device)
    pipe._pool(job, feats_spp)
    torch.cuda.synchronize()
    return pipe, job


def _check_against_oracle(kw, job, thresh=0.999):
    from oracle import gen_ps_oracle as O

    boxes, cls, vol = O.assemble_boxes(kw["coords_float"], kw["instance_cls"], kw["instance_box"],
                                       kw["instance_box_volume"], kw["wall_box"], kw["wall_box_volume"])
    part = O.partition(kw["coords_float"], kw["mask_feats"], kw["spp"], boxes, cls, vol, thresh)
    h = job.header
    coords = np.asarray(kw["coords_float"], dtype=np.float64)
    np.testing.assert_array_equal(np.array(list(h.coord_min)), coords.min(0))
    np.testing.assert_array_equal(np.array(list(h.coord_max)), coords.max(0))
    assert (h.spp_min, h.spp_max) == (int(np.min(kw["spp"])), int(np.max(kw["spp"])))
    assert job.n_spps == part.n_spps
    np.testing.assert_array_equal(job.boxes, boxes)  # incl. the float64 floor box
    np.testing.assert_array_equal(job.boxes_cls, cls)
    np.testing.assert_array_equal(job.boxes_volume, vol)
    np.testing.assert_array_equal(job.spp_inv.cpu().numpy(), part.spp_inv)
    np.testing.assert_array_equal(job.dev["occ_count"].cpu().numpy(), part.occ_count)
    np.testing.assert_array_equal(job.dev["point_count"].cpu().numpy(), part.point_count)
    np.testing.assert_array_equal(job.dev["n_bbs"].cpu().numpy(), part.n_bbs_per_spp)
    bits = job.dev["occ_bits"].cpu().numpy().view(np.uint64)
    B = len(boxes)
    got = np.zeros((part.n_spps, B), dtype=bool)
    for b in range(B):
        got[:, b] = (bits[:, b // 64] >> np.uint64(b % 64)) & np.uint64(1)
    np.testing.assert_array_equal(got, part.occ_spp)
    assert int(h.fixed_shift) == O.fixed_point_shift(float(np.max(np.abs(np.asarray(kw["mask_feats"], np.float32)))),
                                                    len(coords))
    np.testing.assert_array_equal(job.dev["feats_spp"].cpu().numpy(), part.feats_spp)  # bit-exact
    return part


def test_partition_matches_oracle_on_golden(golden):
    kw = golden.api_inputs()
    pipe, job = _run_partition(kw)
    part = _check_against_oracle(kw, job)
    # and therefore the reference: occupancy decisions are the golden ones
    np.testing.assert_array_equal(part.occ_spp, golden["ref_occ_mean"] >= np.float32(0.999))


@pytest.mark.parametrize("n_points,feat_dim,seed", [(60000, 6, 3), (150000, 32, 4), (1000, 6, 9)])
def test_partition_synthetic_sizes(n_points, feat_dim, seed):
    from gapro_amd.gen_ps_utils import getInstanceInfo
    from gapro_amd.synth import make_scene

    sc = make_scene(seed=seed, n_points=n_points)
    xyz = sc.aligned_xyz()
    _, cls, box, vol, _ = getInstanceInfo(xyz, sc.inst, sc.sem)
    rng = np.random.default_rng(seed)
    feats = sc.default_feats().astype(np.float32) if feat_dim == 6 else \
        (rng.standard_normal((sc.n_points, feat_dim)) * 3).astype(np.float32)
    kw = dict(coords_float=xyz, mask_feats=feats, spp=sc.spp, instance_cls=cls.astype(np.int64),
              instance_box=box.astype(np.float32), instance_box_volume=vol.astype(np.float32), wall_box=[],
              wall_box_volume=[])
    for thresh in (0.999, 0.8):
        pipe, job = _run_partition(kw, thresh)
        _check_against_oracle(kw, job, thresh)


def test_partition_negative_and_sparse_superpoint_ids():
    rng = np.random.default_rng(0)
    n = 5000
    coords = rng.uniform(-2, 2, size=(n, 3))
    spp = (rng.integers(0, 40, size=n) * 997 - 20000).astype(np.int64)
    kw = dict(coords_float=coords, mask_feats=rng.standard_normal((n, 6)).astype(np.float32), spp=spp,
              instance_cls=np.array([3, 4]), instance_box=np.array([[-1, -1, -1, 0.5, 0.5, 0.5],
                                                                    [0, 0, 0, 1.5, 1.5, 1.5]], np.float32),
              instance_box_volume=np.array([3.375, 3.375], np.float32), wall_box=[], wall_box_volume=[])
    pipe, job = _run_partition(kw, 0.5)
    _check_against_oracle(kw, job, 0.5)


def test_partition_reports_superpoint_range_overflow():
    from gapro_amd._lib import GaproError

    n = 256
    kw = dict(coords_float=np.zeros((n, 3)), mask_feats=np.ones((n, 6), np.float32),
              spp=np.arange(n, dtype=np.int64) * (1 << 40), instance_cls=np.array([1]),
              instance_box=np.zeros((1, 6), np.float32), instance_box_volume=np.zeros(1, np.float32), wall_box=[],
              wall_box_volume=[])
    with pytest.raises(GaproError) as e:
        _run_partition(kw, spp_range_cap=1 << 16)
    assert e.value.code == -6


def _synth_kw(seed, n_points):
    from gapro_amd.gen_ps_utils import getInstanceInfo
    from gapro_amd.synth import make_scene

    sc = make_scene(seed=seed, n_points=n_points)
    xyz = sc.aligned_xyz()
    _, cls, box, vol, _ = getInstanceInfo(xyz, sc.inst, sc.sem)
    return dict(coords_float=xyz, mask_feats=sc.default_feats().astype(np.float32), spp=sc.spp,
                instance_cls=cls.astype(np.int64), instance_box=box.astype(np.float32),
                instance_box_volume=vol.astype(np.float32), wall_box=[], wall_box_volume=[])


def test_partition_batch_of_ragged_scenes_matches_oracle():
    """One batched launch over scenes of very different sizes (grid.y = scene): every scene bit-exact."""
    import torch
    from gapro_amd.pipeline import Pipeline, make_job

    kws = [_synth_kw(11, 40000), _synth_kw(12, 1500), _synth_kw(13, 90000), _synth_kw(14, 300)]
    pipe = Pipeline(device=0, training_iter=0)
    jobs = [make_job(kw["coords_float"], kw["mask_feats"], kw["spp"], kw["instance_cls"], kw["instance_box"],
                     kw["instance_box_volume"], kw["wall_box"], kw["wall_box_volume"], 18, 0.1, 0.999) for kw in kws]
    tasks, d_tasks = pipe._prepare_all(jobs)
    base = 0
    for job in jobs:
        job.feats_row_base = base
        base += job.n_spps
    feats_spp_all = torch.empty((base, 6), dtype=torch.float32, device=pipe.device)
    pipe._pool_all(jobs, tasks, d_tasks, feats_spp_all)
    torch.cuda.synchronize()
    for kw, job in zip(kws, jobs):
        _check_against_oracle(kw, job)


def test_single_scene_c_entry_points_match_the_batched_ones():
    """gapro_partition_prepare / _pool / gapro_broadcast_labels (one-scene ABI) against the batched path."""
    import torch
    from gapro_amd.pipeline import _ptr

    kw = _synth_kw(21, 20000)
    pipe, job = _run_partition(kw)  # batched path
    lib, ctx, dev = pipe.lib, pipe.ctx, pipe.device
    n, D, S, B = job.n_points, 6, job.n_spps, job.n_boxes
    cap = max(4 * n, 1 << 20)
    nbytes = lib.gapro_partition_prepare_workspace_bytes(n, cap)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    spp_inv = torch.empty(n, dtype=torch.int32, device=dev)
    from gapro_amd._lib import SceneHeader
    hdr = SceneHeader()
    ctx.check(lib.gapro_partition_prepare(ctx.handle, None, n, D, _ptr(job.coords), _ptr(job.feats), _ptr(job.spp),
                                          cap, _ptr(ws), nbytes, _ptr(spp_inv), C.byref(hdr)))
    assert (hdr.n_spps, hdr.fixed_shift, hdr.spp_min, hdr.spp_max) == (
        job.header.n_spps, job.header.fixed_shift, job.header.spp_min, job.header.spp_max)
    assert torch.equal(spp_inv, job.spp_inv)
    W = (B + 63) // 64
    boxes = torch.from_numpy(job.boxes).to(dev)
    feat_sum = torch.empty((S, D), dtype=torch.int64, device=dev)
    occ = torch.empty((S, B), dtype=torch.int32, device=dev)
    pc = torch.empty(S, dtype=torch.int32, device=dev)
    fs = torch.empty((S, D), dtype=torch.float32, device=dev)
    bits = torch.empty((S, W), dtype=torch.int64, device=dev)
    nbb = torch.empty(S, dtype=torch.int32, device=dev)
    ctx.check(lib.gapro_partition_pool(ctx.handle, None, n, D, B, S, int(hdr.fixed_shift), C.c_float(0.999),
                                       _ptr(job.coords), _ptr(job.feats), _ptr(spp_inv), _ptr(boxes), _ptr(feat_sum),
                                       _ptr(occ), _ptr(pc), _ptr(fs), _ptr(bits), _ptr(nbb)))
    torch.cuda.synchronize()
    for a, b in ((occ, job.dev["occ_count"]), (pc, job.dev["point_count"]), (fs, job.dev["feats_spp"]),
                 (bits, job.dev["occ_bits"]), (nbb, job.dev["n_bbs"])):
        assert torch.equal(a, b)
    sem_spp = torch.arange(S, dtype=torch.int32, device=dev)
    prob_spp = torch.rand(S, device=dev)
    sem = torch.empty(n, dtype=torch.int32, device=dev)
    ins = torch.empty(n, dtype=torch.int32, device=dev)
    prb = torch.empty(n, dtype=torch.float32, device=dev)
    ctx.check(lib.gapro_broadcast_labels(ctx.handle, None, n, _ptr(spp_inv), _ptr(sem_spp), _ptr(sem_spp),
                                         _ptr(prob_spp), _ptr(sem), _ptr(ins), _ptr(prb)))
    torch.cuda.synchronize()
    assert torch.equal(sem, sem_spp[spp_inv.long()]) and torch.equal(prb, prob_spp[spp_inv.long()])
