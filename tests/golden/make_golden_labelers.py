#!/usr/bin/env python3
"""Golden outputs of the reference's heuristic labelers (gen_ps_utils.py:242-290 gen_pseudo_label_box2mask,
:485-569 gen_pseudo_label with rules volume / dist / none) on the golden scenes, by running the REAL reference
functions in this container (python tests/golden/make_golden_labelers.py).  Writes tests/golden/labelers_<scene>.npz
(outputs only; the inputs are the fields of <scene>.npz: xyz_aligned, spp, gi_cls, gi_box, gi_vol cast the way
gapro/gen_ps.py:79-89 casts them).  torch_scatter is not installed: the same pure-torch shim as make_golden.py,
extended to the [labels, points] scatter(dim=1 / -1) pattern these functions use.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def extend_shim():
    ts = sys.modules["torch_scatter"]
    base = ts.scatter

    def scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
        if src.dim() == 2 and dim in (1, -1):
            idx = index[0] if index.dim() == 2 else index
            n = int(idx.max()) + 1 if dim_size is None else dim_size
            res = torch.zeros((src.shape[0], n), dtype=src.dtype).index_add_(1, idx, src)
            if reduce == "mean":
                cnt = torch.zeros(n, dtype=src.dtype).index_add_(0, idx, torch.ones(len(idx), dtype=src.dtype)).clamp_(1)
                res = res / cnt[None, :]
            return res
        return base(src, index, dim=dim, out=out, dim_size=dim_size, reduce=reduce)

    ts.scatter = scatter


def main():
    mg.install_stubs()
    extend_shim()
    sys.path.insert(0, mg.REF)
    import gen_ps_utils as ref

    for name, _ in mg.SCENES:
        d = np.load(os.path.join(HERE, name + ".npz"), allow_pickle=True)
        coords = torch.from_numpy(d["xyz_aligned"])
        spp = torch.from_numpy(d["spp"])
        cls = torch.from_numpy(d["gi_cls"]).long()          # gen_ps.py:79
        box = torch.from_numpy(d["gi_box"]).float()         # :80
        vol = torch.from_numpy(d["gi_vol"]).float()         # :81
        out = {}
        for rule in ("volume", "dist", "none"):
            sem, ins = ref.gen_pseudo_label(coords, spp, cls, box, vol, instance_classes=18, dataset_name="scannetv2",
                                            heuristic_rule=rule)
            out[rule + "_sem"], out[rule + "_inst"] = sem.numpy().astype(np.int32), ins.numpy().astype(np.int32)
            sem, ins = ref.gen_pseudo_label(coords, spp, cls, box, vol, instance_classes=18, dataset_name="other",
                                            heuristic_rule=rule)
            out[rule + "_raw_sem"], out[rule + "_raw_inst"] = sem.numpy().astype(np.int32), ins.numpy().astype(np.int32)
        sem, ins = ref.gen_pseudo_label_box2mask(coords, spp, cls, box, vol, instance_classes=18, dataset_name="scannetv2")
        out["box2mask_sem"], out["box2mask_inst"] = sem.numpy().astype(np.int32), ins.numpy().astype(np.int32)
        np.savez_compressed(os.path.join(HERE, "labelers_" + name + ".npz"), **out)
        print(name, {k: (int((v >= 0).sum()) if k.endswith("inst") else None) for k, v in out.items() if k.endswith("inst")})


if __name__ == "__main__":
    main()
