"""HIP label-side kernels against the CPU restatements in oracle/eval_oracle.py:
bit-exact for boxes / classes / volumes / counts, and for the float32 IoUs (same operation order)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(seed, n_points, n_objects=25):
    from gapro_amd.synth import make_scene

    return make_scene(seed=seed, n_points=n_points, n_objects=n_objects)


@pytest.mark.parametrize("seed,n_points", [(0, 50000), (1, 150000), (2, 700)])
def test_instance_info_matches_oracle_and_host_mirror(seed, n_points):
    from gapro_amd.gen_ps_utils import getInstanceInfo, getInstanceInfo_device
    from oracle.eval_oracle import get_instance_info

    sc = _scene(seed, n_points)
    xyz = sc.aligned_xyz()
    inst = sc.inst.astype(np.float64)
    sem = sc.sem.astype(np.float64)
    ref = get_instance_info(xyz, inst, sem)
    got = getInstanceInfo_device(xyz, inst, sem, return_corners=True)
    host = getInstanceInfo(xyz, inst, sem)
    assert got[0] == ref[0] == host[0]
    for a, b, c in zip(got[1:4], ref[1:4], host[1:4]):
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(a, np.asarray(c, dtype=np.float64))
    np.testing.assert_array_equal(got[4], ref[4])


def test_instance_info_edge_cases():
    from gapro_amd.gen_ps_utils import getInstanceInfo_device
    from oracle.eval_oracle import get_instance_info

    rng = np.random.default_rng(3)
    n = 4000
    xyz = rng.uniform(-5, 5, size=(n, 3))
    # sparse ids with gaps, negatives (-100 = unlabeled), one id beyond the first table size, sem -100 kept as is
    inst = rng.choice(np.array([-100.0, 0.0, 3.0, 4.0, 17.0, 1500.0]), size=n)
    sem = rng.choice(np.array([-100.0, 2.0, 5.0, 19.0]), size=n)
    ref = get_instance_info(xyz, inst, sem)
    got = getInstanceInfo_device(xyz, inst, sem, return_corners=True)
    assert got[0] == ref[0] == 1501
    for a, b in zip(got[1:], ref[1:]):
        np.testing.assert_array_equal(a, b)
    # no instance at all -> None, like the reference
    assert getInstanceInfo_device(xyz, np.full(n, -100.0), sem) is None
    # a single point per instance: zero volume, box = the point
    one = getInstanceInfo_device(xyz[:3], np.array([0.0, 1.0, 2.0]), np.array([4.0, 4.0, -100.0]), dataset_name="other")
    np.testing.assert_array_equal(one[2][:, :3], xyz[:3])
    np.testing.assert_array_equal(one[3], np.zeros(3))
    np.testing.assert_array_equal(one[1], np.array([4.0, 4.0, -100.0]))


def _labels(seed, n_points):
    import torch

    sc = _scene(seed, n_points)
    rng = np.random.default_rng(seed + 100)
    gt_sem = sc.sem.astype(np.int64).copy()
    gt_sem[gt_sem != -100] -= 2  # gen_ps.py:119-120
    gt_sem[(gt_sem == -1) | (gt_sem == -2)] = 18
    gt_ins = sc.inst.astype(np.int64)
    # a noisy pseudo labeling: permuted ids, some points dropped to -100, some reassigned
    perm = rng.permutation(int(gt_ins.max()) + 3)
    ps_ins = np.where(gt_ins >= 0, perm[np.clip(gt_ins, 0, None)], -100)
    ps_sem = gt_sem.copy()
    drop = rng.random(len(ps_ins)) < 0.1
    ps_ins[drop] = -100
    ps_sem[drop] = -100
    move = rng.random(len(ps_ins)) < 0.05
    ps_ins[move] = rng.integers(0, int(ps_ins.max()) + 1, size=int(move.sum()))
    return tuple(torch.from_numpy(a) for a in (gt_sem, gt_ins, ps_sem, ps_ins))


@pytest.mark.parametrize("seed,n_points", [(0, 60000), (5, 150000), (7, 900)])
def test_miou_and_confusion_match_oracle(seed, n_points):
    import torch
    from gapro_amd.eval_ps_labels import get_miou_scene, get_scene_sem_conf
    from oracle import eval_oracle as E

    gt_sem, gt_ins, ps_sem, ps_ins = _labels(seed, n_points)
    ref = E.get_miou_scene(gt_sem, gt_ins, ps_sem, ps_ins)
    got = get_miou_scene(gt_sem.cuda(), gt_ins.cuda(), ps_sem.cuda(), ps_ins.cuda())
    np.testing.assert_array_equal(got.cpu().numpy(), ref.numpy())  # float32, bit-exact
    conf_ref = E.get_scene_sem_conf(gt_sem, ps_sem)
    conf = get_scene_sem_conf(gt_sem.cuda(), ps_sem.cuda())
    assert torch.equal(conf.cpu(), conf_ref)
    assert int(conf.sum()) == int((gt_sem != -100).sum())


def test_miou_on_golden_matches_reference(golden):
    """The HIP evaluator on the reference's own outputs reproduces the IoUs the imported reference printed."""
    import torch
    from gapro_amd.eval_ps_labels import get_miou_scene

    gt_sem = torch.from_numpy(golden["sem_gt"]).int()
    gt_ins = torch.from_numpy(golden["inst_gt"]).int()
    gt_sem[gt_sem != -100] -= 2
    gt_sem[(gt_sem == -1) | (gt_sem == -2)] = 18
    ious = get_miou_scene(gt_sem.long().cuda(), gt_ins.long().cuda(), torch.from_numpy(golden["out_sem"]).long().cuda(),
                          torch.from_numpy(golden["out_inst"]).long().cuda())
    np.testing.assert_array_equal(ious.cpu().numpy(), golden["ref_ious"])


def test_miou_large_id_tables_and_empty_pseudo_labels():
    import torch
    from gapro_amd.eval_ps_labels import get_miou_scene
    from oracle import eval_oracle as E

    rng = np.random.default_rng(1)
    n = 20000
    gt_ins = torch.from_numpy(rng.integers(-1, 700, size=n))  # beyond the first table size and the LDS bins
    gt_sem = torch.from_numpy(rng.integers(0, 19, size=n))
    ps_ins = torch.from_numpy(rng.integers(-1, 900, size=n))
    ps_sem = torch.from_numpy(rng.integers(0, 19, size=n))
    ref = E.get_miou_scene(gt_sem, gt_ins, ps_sem, ps_ins)
    got = get_miou_scene(gt_sem.cuda(), gt_ins.cuda(), ps_sem.cuda(), ps_ins.cuda())
    np.testing.assert_array_equal(got.cpu().numpy(), ref.numpy())
    none = torch.full((n,), -100, dtype=torch.int64)
    got0 = get_miou_scene(gt_sem.cuda(), gt_ins.cuda(), none.cuda(), none.cuda())
    np.testing.assert_array_equal(got0.cpu().numpy(), E.get_miou_scene(gt_sem, gt_ins, none, none).numpy())
