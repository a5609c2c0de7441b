"""Heuristic labelers: the NumPy oracle against the real reference's outputs on the golden
scenes (CPU), and the HIP kernels against both (GPU)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_NAMES, Golden

HERE = os.path.dirname(os.path.abspath(__file__))


def _inputs(name):
    g = Golden(name)
    return dict(coords=g["xyz_aligned"], spp=g["spp"], instance_cls=g["gi_cls"].astype(np.int64),
                instance_box=g["gi_box"].astype(np.float32), instance_box_volume=g["gi_vol"].astype(np.float32))


def _expected(name):
    return np.load(os.path.join(HERE, "golden", "labelers_" + name + ".npz"))


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_oracle_matches_the_reference(name):
    from oracle import labeler_oracle as L

    kw, exp = _inputs(name), _expected(name)
    for rule in ("volume", "dist", "none"):
        sem, ins = L.gen_pseudo_label(**kw, heuristic_rule=rule)
        np.testing.assert_array_equal(sem, exp[rule + "_sem"])
        np.testing.assert_array_equal(ins, exp[rule + "_inst"])
        sem, ins = L.gen_pseudo_label(**kw, heuristic_rule=rule, dataset_name="other")
        np.testing.assert_array_equal(sem, exp[rule + "_raw_sem"])
        np.testing.assert_array_equal(ins, exp[rule + "_raw_inst"])
    sem, ins = L.gen_pseudo_label_box2mask(**kw)
    np.testing.assert_array_equal(sem, exp["box2mask_sem"])
    np.testing.assert_array_equal(ins, exp["box2mask_inst"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_hip_labelers_match_the_reference(name):
    from gapro_amd.gen_ps_utils import gen_pseudo_label, gen_pseudo_label_box2mask

    kw, exp = _inputs(name), _expected(name)
    args = (kw["coords"], kw["spp"], kw["instance_cls"], kw["instance_box"], kw["instance_box_volume"])
    for rule in ("volume", "dist", "none"):
        sem, ins = gen_pseudo_label(*args, heuristic_rule=rule)
        np.testing.assert_array_equal(sem.cpu().numpy(), exp[rule + "_sem"])
        np.testing.assert_array_equal(ins.cpu().numpy(), exp[rule + "_inst"])
        sem, ins = gen_pseudo_label(*args, heuristic_rule=rule, dataset_name="other")
        np.testing.assert_array_equal(sem.cpu().numpy(), exp[rule + "_raw_sem"])
        np.testing.assert_array_equal(ins.cpu().numpy(), exp[rule + "_raw_inst"])
    sem, ins = gen_pseudo_label_box2mask(*args)
    np.testing.assert_array_equal(sem.cpu().numpy(), exp["box2mask_sem"])
    np.testing.assert_array_equal(ins.cpu().numpy(), exp["box2mask_inst"])


@pytest.mark.gpu
def test_hip_labelers_match_oracle_on_a_full_size_scene():
    from gapro_amd.gen_ps_utils import gen_pseudo_label, gen_pseudo_label_box2mask, getInstanceInfo
    from gapro_amd.synth import make_scene
    from oracle import labeler_oracle as L

    sc = make_scene(seed=3, n_points=150000, n_objects=25)
    xyz = sc.aligned_xyz()
    _, cls, box, vol, _ = getInstanceInfo(xyz, sc.inst, sc.sem)
    args = (xyz, sc.spp, cls.astype(np.int64), box.astype(np.float32), vol.astype(np.float32))
    for rule in ("volume", "dist", "none"):
        sem, ins = gen_pseudo_label(*args, heuristic_rule=rule)
        rs, ri = L.gen_pseudo_label(*args, heuristic_rule=rule)
        np.testing.assert_array_equal(sem.cpu().numpy(), rs)
        np.testing.assert_array_equal(ins.cpu().numpy(), ri)
    sem, ins = gen_pseudo_label_box2mask(*args)
    rs, ri = L.gen_pseudo_label_box2mask(*args)
    np.testing.assert_array_equal(sem.cpu().numpy(), rs)
    np.testing.assert_array_equal(ins.cpu().numpy(), ri)
