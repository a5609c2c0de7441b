"""End-to-end pseudo-label generation on the GPU vs the oracle, through the reference-named API."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_outputs(kw):
    from oracle import gen_ps_oracle as O
    from oracle.svgp_oracle import fit_gp_spp_oracle

    def fit_fn(feats_spp, b1, b2, it):
        return fit_gp_spp_oracle(feats_spp, b1, b2, it, 50, impl="autograd", dtype="f64")

    return O.gen_pseudo_label_gaussian_process(**kw, fit_fn=fit_fn, return_debug=True)


def _check(outs, ref, dbg):
    sem, ins, prob, mu, var = [o.cpu().numpy() for o in outs]
    r_sem, r_ins, r_prob, r_mu, r_var = ref
    assert sem.dtype == np.int32 and ins.dtype == np.int32 and prob.dtype == np.float32
    assert mu.shape == r_mu.shape and len(mu) == dbg["part"].n_spps  # superpoint length
    # integer masks bit-exact, unless a GP probability sits within float32 rounding of a tie
    tie = [np.min(np.abs(np.asarray(r[0], np.float64) - 0.5)) for r in dbg["results"]]
    assert not tie or min(tie) > 1e-5, "fixture has a GP tie; pick another seed"
    np.testing.assert_array_equal(sem, r_sem)
    np.testing.assert_array_equal(ins, r_ins)
    np.testing.assert_allclose(prob, r_prob, rtol=0, atol=3e-7)
    gp = r_mu != -100
    np.testing.assert_array_equal(mu == -100, ~gp)
    np.testing.assert_allclose(var[gp], r_var[gp], rtol=1e-4)  # north_star tolerance
    np.testing.assert_allclose(mu[gp], r_mu[gp], rtol=1e-4, atol=1e-6)
    np.testing.assert_array_equal(var[~gp], r_var[~gp])


def test_generator_matches_oracle_on_golden(golden):
    from gapro_amd import gen_pseudo_label_gaussian_process

    kw = golden.api_inputs()
    outs = gen_pseudo_label_gaussian_process(**kw)
    ref, dbg = _oracle_outputs(kw)
    _check(outs, ref, dbg)
    # reference-level anchor: pairs fitted == pairs the real reference fitted
    assert len(dbg["results"]) == int(golden["n_fits"])


def test_batch_of_scenes_equals_scene_by_scene():
    import torch
    from gapro_amd import gen_pseudo_label_gaussian_process, gen_pseudo_label_gaussian_process_batch
    from conftest import GOLDEN_NAMES, Golden

    scenes = [Golden(n).api_inputs() for n in GOLDEN_NAMES[:3]]
    batch = gen_pseudo_label_gaussian_process_batch(scenes, training_iter=50)
    for kw, got in zip(scenes, batch):
        one = gen_pseudo_label_gaussian_process(**kw, device="cuda:0")
        for a, b in zip(got, one):
            assert torch.equal(a.cpu(), b.cpu())


def test_api_conventions(golden):
    import torch
    from gapro_amd import gen_pseudo_label_gaussian_process

    kw = golden.api_inputs()
    cpu_out = gen_pseudo_label_gaussian_process(**kw)
    assert all(not o.is_cuda for o in cpu_out)
    tkw = dict(kw)
    tkw["coords_float"] = torch.from_numpy(kw["coords_float"]).cuda()
    tkw["mask_feats"] = torch.from_numpy(kw["mask_feats"]).cuda()
    tkw["spp"] = torch.from_numpy(kw["spp"]).cuda()
    dev_out = gen_pseudo_label_gaussian_process(**tkw, broadcast_mu_var=True)
    assert all(o.is_cuda for o in dev_out)
    n = len(kw["spp"])
    assert [len(o) for o in dev_out] == [n] * 5
    for a, b in zip(cpu_out[:3], dev_out[:3]):
        assert torch.equal(a, b.cpu())
    _, inv = np.unique(kw["spp"], return_inverse=True)
    assert torch.equal(dev_out[3].cpu(), cpu_out[3][torch.from_numpy(inv)])


def _bench_scene(seed, n_points=150000):
    from gapro_amd.gen_ps_utils import getInstanceInfo
    from gapro_amd.synth import make_scene

    sc = make_scene(seed=seed, n_points=n_points, n_objects=25, with_walls_json=False)
    xyz = sc.aligned_xyz()
    _, cls, box, vol, _ = getInstanceInfo(xyz, sc.inst, sc.sem)
    return dict(coords_float=xyz, mask_feats=sc.default_feats().astype(np.float32), spp=sc.spp,
                instance_cls=cls.astype(np.int64), instance_box=box.astype(np.float32),
                instance_box_volume=vol.astype(np.float32), wall_box=[], wall_box_volume=[])


def test_full_size_scenes_size_independent_properties():
    """bench.py-sized scenes (150k points, the configs[1] shape), checked through properties that do not need
    the CPU oracle at that size: pipelined == batch-by-batch (bitwise), repeatable, point-order invariant
    (the pooled sums are exact integers), per-superpoint consistency of the broadcast labels."""
    import torch
    from gapro_amd.pipeline import Pipeline, make_job

    kws = [_bench_scene(s) for s in range(4)]
    pipe = Pipeline(device=0, training_iter=50)

    def jobs(kw_list):
        return [make_job(kw["coords_float"], kw["mask_feats"], kw["spp"], kw["instance_cls"], kw["instance_box"],
                         kw["instance_box_volume"], kw["wall_box"], kw["wall_box_volume"], 18, 0.1, 0.999) for kw in kw_list]

    seq = [pipe.run(jobs(kws[:2])), pipe.run(jobs(kws[2:]))]
    pip = pipe.run_pipelined([jobs(kws[:2]), jobs(kws[2:])])
    again = pipe.run_pipelined([jobs(kws[:2]), jobs(kws[2:])])
    torch.cuda.synchronize()
    for b in range(2):
        for s in range(2):
            for x, y, z in zip(seq[b][s], pip[b][s], again[b][s]):
                assert torch.equal(x, y) and torch.equal(x, z)
    # per-superpoint consistency and value ranges
    for kw, outs in zip(kws, seq[0] + seq[1]):
        sem, ins, prob, mu, var = [o.cpu().numpy() for o in outs]
        uniq, inv = np.unique(kw["spp"], return_inverse=True)
        assert len(mu) == len(uniq) == len(var)
        for arr in (sem, ins, prob):
            first = np.zeros(len(uniq), dtype=arr.dtype)
            first[inv] = arr
            np.testing.assert_array_equal(arr, first[inv])  # one value per superpoint
        n_inst = len(kw["instance_cls"])
        assert ins.max() < n_inst and ((ins >= 0) | (ins == -100)).all()
        assert ((prob >= 0) & (prob <= 1)).all() and np.isfinite(prob).all()
        gp = mu != -100
        assert gp.any() and (var[gp] > 0).all() and (var[~gp] == -100).all()
    # point-order invariance: a permuted scene gives the permuted labels (and the same mu / var)
    kw = kws[0]
    rng = np.random.default_rng(0)
    perm = rng.permutation(len(kw["spp"]))
    kwp = dict(kw, coords_float=kw["coords_float"][perm], mask_feats=kw["mask_feats"][perm], spp=kw["spp"][perm])
    outp = pipe.run(jobs([kwp]))[0]
    base = seq[0][0]
    for a, b in zip(base[:3], outp[:3]):
        assert torch.equal(a.cpu()[torch.from_numpy(perm)], b.cpu())
    assert torch.equal(base[3], outp[3]) and torch.equal(base[4], outp[4])


def test_s3dis_shaped_scene_matches_oracle():
    """BASELINE configs[3]: a room-sized scene (1M points, 13 classes, objects of up to several hundred
    superpoints).  Its 66 fits (M up to 724) span every fit kernel inside one real schedule.

    Tolerance rule (round 6; no oracle-side quantity in it, nothing that depends on the oracle's BLAS or thread count):
    the PRODUCT says which fits are numerically soft -- `reproducibility_probe=True` runs every fit once more with the
    jitter on K_ZZ's diagonal scaled by (1 + 1e-11), a few ulps, and reports how far sigma^2 and p move.  A fit that moves by less than
    REPRO_SOFT (1e-5; the well-behaved fits move by 1e-7 .. 1e-6) must agree with the float64 autograd oracle to float32 rounding (p within
    3e-7, sigma^2 within 1e-5 relative); a fit that moves by more amplifies last-bit differences ~1e9-fold over its fifty
    Adam steps -- in ANY float64 implementation: on exactly these fits the oracle's own two implementations part by
    3e-6 .. 3e-5 and a one-ulp change of one input moves the oracle's sigma^2 by as much (tools/loose_fits.py,
    tests/test_svgp_oracle.py::test_the_one_ill_conditioned_s3dis_fit_...) -- and is held to 1e-3.
    Two of the 66 fits are of that kind (M = 58 and M = 144), and the count is asserted.  (Neither a pivot-ratio figure
    of the Cholesky factor nor cond_2(K_ZZ) singles them out: they rank 38th / 55th and 35th / 61st of 66 by those.)
    Integer masks are compared bit for bit wherever the GP probability is further from a tie than the deviation
    allowed above."""
    from gapro_amd import gen_pseudo_label_gaussian_process
    from gapro_amd._lib import Context
    from gapro_amd.gaussian_process_utils import fit_gp_spp_batch
    from gapro_amd.gen_ps_utils import getInstanceInfo
    from gapro_amd.pipeline import REPRO_SOFT
    from gapro_amd.synth import make_scene

    sc = make_scene(seed=7, n_points=1_000_000, n_objects=40, with_walls_json=False, obj_patch=60, plane_patch=400)
    xyz = sc.aligned_xyz()
    _, cls, box, vol, _ = getInstanceInfo(xyz, sc.inst, sc.sem)
    kw = dict(coords_float=xyz, mask_feats=sc.default_feats().astype(np.float32), spp=sc.spp,
              instance_cls=cls.astype(np.int64), instance_box=box.astype(np.float32),
              instance_box_volume=vol.astype(np.float32), wall_box=[], wall_box_volume=[], instance_classes=13,
              ground_h=0.1, training_iter=50, thresh_spp_occu=0.999)
    ref, dbg = _oracle_outputs(kw)
    fits = [e for e in dbg["events"] if e.kind == "fit"]
    lib = Context.get(0).lib
    routes = {int(lib.gapro_fit_route(len(e.b1_inds) + len(e.b2_inds), 6)) for e in fits}
    assert {0, 1, 3, 5} <= routes and (routes & {2, 4}), routes  # the scene really exercises every kernel family
    # fit by fit, on the oracle's pooled features (the partition is compared bit for bit through the masks below)
    got, res = fit_gp_spp_batch(dbg["part"].feats_spp, [(e.b1_inds, e.b2_inds, e.intersect_inds) for e in fits],
                                training_iter=50, reproducibility_probe=True)
    assert np.isfinite(res["cond"]).all() and (res["cond"] >= 1.0).all()
    soft = [k for k in range(len(fits)) if max(res["repro_dv"][k], res["repro_dp"][k]) > REPRO_SOFT]
    loose, worst_p = [], 3e-7
    for k, (g, r, e) in enumerate(zip(got, dbg["results"], fits)):
        dp = np.max(np.abs(g[0].astype(np.float64) - r[0]))
        dv = np.max(np.abs(g[4].astype(np.float64) - r[4]) / r[4])
        if k in soft:
            assert dv < 1e-3 and dp < 1e-3, (k, dv, dp)
            loose.append(k)
            worst_p = max(worst_p, dp)
        else:
            assert dp < 3e-7 and dv < 1e-5, "fit %d (M=%d): off by dv=%.2e dp=%.2e while its own perturbation probe " \
                "moves it by dv=%.1e dp=%.1e" % (k, len(e.b1_inds) + len(e.b2_inds), dv, dp, res["repro_dv"][k],
                                                 res["repro_dp"][k])
    assert sorted(len(fits[k].b1_inds) + len(fits[k].b2_inds) for k in soft) == [58, 144], soft
    outs = gen_pseudo_label_gaussian_process(**kw)
    sem, ins, prob, mu, var = [o.cpu().numpy() for o in outs]
    r_sem, r_ins, r_prob, r_mu, r_var = ref
    np.testing.assert_allclose(prob, r_prob, rtol=0, atol=max(2 * worst_p, 1e-6))
    safe = np.abs(r_prob.astype(np.float64) - 0.5) > 2 * worst_p
    assert safe.mean() > 0.999
    np.testing.assert_array_equal(sem[safe], r_sem[safe])
    np.testing.assert_array_equal(ins[safe], r_ins[safe])
    gp = r_mu != -100
    np.testing.assert_array_equal(mu == -100, ~gp)
    ok = np.abs(var[gp] - r_var[gp]) <= 1e-4 * r_var[gp]  # north_star tolerance ...
    n_loose_spp = sum(len(fits[k].intersect_inds) for k in loose)
    assert (~ok).sum() <= n_loose_spp  # ... everywhere but on superpoints labelled by the ill-conditioned fit(s)


@pytest.mark.parametrize("seed", [63, 20])
def test_train_split_shaped_scenes_match_oracle(seed):
    """BASELINE configs[2]'s scene stream as bench.py draws it (N ~ logN(150k, 0.5), 10 .. 40 objects, wall boxes
    read back from ScanNet-Planes quads through get_wall_boxes): the two smallest scenes with walls of the first 64
    seeds (50k and 70k points, 23 fits each, M up to 105), whole generator against the oracle."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from gapro_amd import gen_pseudo_label_gaussian_process

    kw = bench.build_scene_inputs(seed, 150000, 6, "stream")
    assert len(kw["wall_box"]) == 4
    kw["training_iter"] = 50
    ref, dbg = _oracle_outputs(kw)
    assert len(dbg["results"]) >= 20
    _check(gen_pseudo_label_gaussian_process(**kw), ref, dbg)


def test_scenes_without_any_gp_fit_and_mixed_batches():
    """Edge cases of the schedule: a scene whose boxes never overlap on a shared superpoint has no GP fit at all
    (labels come from containment alone, mu / var stay -100), alone and batched with a scene that does."""
    import torch
    from gapro_amd import gen_pseudo_label_gaussian_process, gen_pseudo_label_gaussian_process_batch

    rng = np.random.default_rng(4)
    n = 3000
    # two well separated blobs, one box each, superpoints never straddle the boxes
    a = rng.uniform([0, 0, 0.2], [1, 1, 1.2], size=(n // 2, 3))
    b = rng.uniform([5, 5, 0.2], [6, 6, 1.2], size=(n - n // 2, 3))
    xyz = np.concatenate([a, b])
    spp = np.concatenate([rng.integers(0, 20, n // 2), 100 + rng.integers(0, 20, n - n // 2)]).astype(np.int64)
    boxes = np.array([[0, 0, 0.2, 1, 1, 1.2], [5, 5, 0.2, 6, 6, 1.2]], np.float32)
    kw = dict(coords_float=xyz, mask_feats=rng.standard_normal((n, 6)).astype(np.float32), spp=spp,
              instance_cls=np.array([3, 7]), instance_box=boxes,
              instance_box_volume=np.prod(boxes[:, 3:] - boxes[:, :3], axis=1).astype(np.float32), wall_box=[],
              wall_box_volume=[], instance_classes=18, ground_h=0.1, training_iter=50, thresh_spp_occu=0.999)
    ref, dbg = _oracle_outputs(kw)
    assert len(dbg["results"]) == 0  # no fit in this scene
    out = gen_pseudo_label_gaussian_process(**kw)
    for got, want in zip(out, ref):
        np.testing.assert_array_equal(got.cpu().numpy(), want)
    assert (out[3].numpy() == -100).all() and (out[4].numpy() == -100).all()
    from conftest import GOLDEN_NAMES, Golden

    g = Golden(GOLDEN_NAMES[0]).api_inputs()
    both = gen_pseudo_label_gaussian_process_batch([kw, g], training_iter=50)
    alone = gen_pseudo_label_gaussian_process(**g, device="cuda:0")
    for x, y in zip(both[0], out):
        assert torch.equal(x.cpu(), y.cpu())
    for x, y in zip(both[1], alone):
        assert torch.equal(x.cpu(), y.cpu())


@pytest.mark.parametrize("n_points", [500_000, 2_000_000, 4_000_000])
def test_s3dis_sizes_size_independent_properties(n_points):
    """BASELINE configs[3]'s other sizes (N in {0.5, 1, 2, 4} M points; 1 M is compared with the oracle
    above).  At these sizes the CPU oracle is out of reach, so the size-independent properties: repeatable bit for bit,
    one value per superpoint, value ranges, point-order invariance (the pooled sums are exact integers), and the
    partition's per-superpoint occupancy re-derived on the host for a sample of superpoints."""
    import torch
    from gapro_amd import gen_pseudo_label_gaussian_process
    from gapro_amd.gen_ps_utils import getInstanceInfo
    from gapro_amd.synth import make_scene

    sc = make_scene(seed=5, n_points=n_points, n_objects=50, with_walls_json=False, obj_patch=150, plane_patch=400)
    xyz = sc.aligned_xyz()
    _, cls, box, vol, _ = getInstanceInfo(xyz, sc.inst, sc.sem)
    kw = dict(coords_float=xyz, mask_feats=sc.default_feats().astype(np.float32), spp=sc.spp,
              instance_cls=cls.astype(np.int64), instance_box=box.astype(np.float32),
              instance_box_volume=vol.astype(np.float32), wall_box=[], wall_box_volume=[], instance_classes=13,
              dataset_name="s3dis", ground_h=0.1, training_iter=50, thresh_spp_occu=0.999)
    a = [o.cpu().numpy() for o in gen_pseudo_label_gaussian_process(**kw)]
    b = [o.cpu().numpy() for o in gen_pseudo_label_gaussian_process(**kw)]
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    sem, ins, prob, mu, var = a
    uniq, inv = np.unique(sc.spp, return_inverse=True)
    assert len(sem) == n_points and len(mu) == len(uniq) == len(var)
    for arr in (sem, ins, prob):
        first = np.zeros(len(uniq), dtype=arr.dtype)
        first[inv] = arr
        np.testing.assert_array_equal(arr, first[inv])  # one value per superpoint
    assert ins.max() < len(cls) and ((ins >= 0) | (ins == -100)).all()
    assert set(np.unique(sem).tolist()) <= set(cls.tolist()) | {13, -100}  # a box's class, background, or ignore
    assert ((prob >= 0) & (prob <= 1)).all() and np.isfinite(prob).all()
    gp = mu != -100
    assert gp.any() and (var[gp] > 0).all() and np.isfinite(mu[gp]).all() and (var[~gp] == -100).all()
    # point-order invariance
    perm = np.random.default_rng(1).permutation(n_points)
    kwp = dict(kw, coords_float=xyz[perm], mask_feats=kw["mask_feats"][perm], spp=sc.spp[perm])
    p = [o.cpu().numpy() for o in gen_pseudo_label_gaussian_process(**kwp)]
    for x, y in zip(a[:3], p[:3]):
        np.testing.assert_array_equal(x[perm], y)
    np.testing.assert_array_equal(a[3], p[3])
    np.testing.assert_array_equal(a[4], p[4])
    # a superpoint labelled with an instance lies inside that instance's box (+- the 0.005 margin of the membership test)
    lab = np.flatnonzero(ins >= 0)[:: max(1, n_points // 2000)]
    bx = box[ins[lab]].astype(np.float64)
    assert ((xyz[lab] >= bx[:, :3] - 0.0051) & (xyz[lab] <= bx[:, 3:] + 0.0051)).all()
