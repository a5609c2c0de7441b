"""The oracle's non-GP half against the outputs of the imported reference (tests/golden/*.npz)."""
import numpy as np

from oracle import gen_ps_oracle as O


def _replay_fit(golden):
    fits = golden.fits
    state = {"i": 0}

    def fit_fn(feats_spp, b1, b2, it):
        f = fits[state["i"]]
        state["i"] += 1
        # the schedule itself is part of the contract: same pairs, same order, same index sets
        np.testing.assert_array_equal(b1, f["b1_inds"])
        np.testing.assert_array_equal(b2, f["b2_inds"])
        np.testing.assert_array_equal(it, f["intersect_inds"])
        return f["probs"], f["probs_new"], f["labels"], f["mu"], f["var"]

    return fit_fn, state


def test_partition_matches_reference(golden):
    kw = golden.api_inputs()
    boxes, cls, vol = O.assemble_boxes(kw["coords_float"], kw["instance_cls"], kw["instance_box"],
                                       kw["instance_box_volume"], kw["wall_box"], kw["wall_box_volume"])
    part = O.partition(kw["coords_float"], kw["mask_feats"], kw["spp"], boxes, cls, vol, 0.999)
    ref_occ_mean = golden["ref_occ_mean"]
    assert part.n_spps == ref_occ_mean.shape[0]
    np.testing.assert_array_equal(part.occ_spp, ref_occ_mean >= np.float32(0.999))
    # float32 occupancy means are one IEEE division of two integers: bit-exact
    occ_mean = part.occ_count.astype(np.float32) / part.point_count.astype(np.float32)[:, None]
    np.testing.assert_array_equal(occ_mean, ref_occ_mean)
    # pooled features: exact fixed-point mean vs the reference's sequential float32 mean
    np.testing.assert_allclose(part.feats_spp, golden["ref_feats_spp"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(part.coords_spp, golden["ref_coords_spp"], rtol=1e-12, atol=1e-12)


def test_schedule_merge_labels_match_reference(golden):
    kw = golden.api_inputs()
    fit_fn, state = _replay_fit(golden)
    sem, ins, prob, mu, var = O.gen_pseudo_label_gaussian_process(**kw, fit_fn=fit_fn)
    assert state["i"] == int(golden["n_fits"])
    for got, key in ((sem, "out_sem"), (ins, "out_inst"), (prob, "out_prob"), (mu, "out_mu"), (var, "out_var")):
        ref = golden[key]
        assert got.dtype == ref.dtype and got.shape == ref.shape, key
        np.testing.assert_array_equal(got, ref, err_msg=key)
    # mu/var stay superpoint-length, the other three are point-length
    assert len(mu) != len(sem)


def test_golden_covers_scheduler_branches():
    """Across the fixtures every branch of the pair loop fires at least once."""
    from conftest import GOLDEN_NAMES, Golden

    kinds = {"contain": 0, "fit": 0}
    skipped_iou = 0
    for name in GOLDEN_NAMES:
        g = Golden(name)
        kw = g.api_inputs()
        boxes, cls, vol = O.assemble_boxes(kw["coords_float"], kw["instance_cls"], kw["instance_box"],
                                           kw["instance_box_volume"], kw["wall_box"], kw["wall_box_volume"])
        part = O.partition(kw["coords_float"], kw["mask_feats"], kw["spp"], boxes, cls, vol, 0.999)
        for e in O.enumerate_schedule(boxes, part.occ_spp, part.n_bbs_per_spp):
            kinds[e.kind] += 1
        iou = O.batch_iou_cross(boxes, boxes)
        np.fill_diagonal(iou, 0)
        skipped_iou += int((iou >= 0.6).sum())
    assert kinds["contain"] > 0 and kinds["fit"] > 0
    assert skipped_iou > 0
