import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
GOLDEN_NAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "s[0-9]*.npz")))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle's small float64 torch ops crawl when OpenMP spreads them over every hardware thread of a big
    # host (the GPU box shows 256 and grants far fewer): 8 threads are plenty for matrices of a few hundred rows
    try:
        import torch

        if (os.cpu_count() or 1) > 8:
            torch.set_num_threads(8)
    except ImportError:
        pass


class Golden:
    """One fixture file written by tests/golden/make_golden.py (outputs of the real reference)."""

    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)

    def __getitem__(self, k):
        return self.z[k]

    @property
    def fits(self):
        out = []
        for i in range(int(self.z["n_fits"])):
            out.append({k: self.z["fit%03d_%s" % (i, k)] for k in
                        ("b1_inds", "b2_inds", "intersect_inds", "probs", "probs_new", "labels", "mu", "var")})
        return out

    def api_inputs(self):
        """Arguments of gen_pseudo_label_gaussian_process exactly as reference gen_ps.py:79-111 casts them."""
        z = self.z
        wall_box = z["wall_box"].astype(np.float32) if z["wall_box"].size else []
        wall_vol = z["wall_vol"].astype(np.float32) if z["wall_vol"].size else []
        feats = np.concatenate([z["xyz_raw"], z["rgb"]], axis=-1).astype(np.float32)
        return dict(coords_float=z["xyz_aligned"], mask_feats=feats, spp=z["spp"],
                    instance_cls=z["gi_cls"].astype(np.int64), instance_box=z["gi_box"].astype(np.float32),
                    instance_box_volume=z["gi_vol"].astype(np.float32), wall_box=wall_box,
                    wall_box_volume=wall_vol, instance_classes=18, dataset_name="scannetv2", ground_h=0.1,
                    training_iter=50, thresh_spp_occu=0.999)


@pytest.fixture(params=GOLDEN_NAMES)
def golden(request):
    return Golden(request.param)
