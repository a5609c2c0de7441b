"""Multi-GPU plumbing of the pseudo-label path: scene sharding and the timing reduction.

The path has no exchange step: scenes are independent and results are per-scene files (SURVEY.md
section 8e), so N GPUs = N replicas over a sharded scene list and **no data-path collective**.  The only
collectives are control-plane: a barrier around the timed region and a MAX of the wall time
(bench.py), both through torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests).
"""
from __future__ import annotations

import os
from typing import List, Sequence


def shard_scenes(filenames: Sequence[str], rank: int, world: int) -> List[str]:
    """Round-robin shard of the *sorted* scene list: rank r takes scenes r, r+world, ...

    Every scene belongs to exactly one rank; combined with the driver's skip-if-exists
    (reference gen_ps.py:39-41) a restarted job redoes nothing."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(sorted(filenames))[rank::world]


def pending_scenes(filenames: Sequence[str], save_folder: str) -> List[str]:
    """Scenes whose output file does not exist yet (reference gen_ps.py:37-41: scan name = first 12 chars)."""
    out = []
    for fn in filenames:
        scan_name = fn.split("/")[-1][:12]
        if not os.path.exists(os.path.join(save_folder, scan_name + ".pth")):
            out.append(fn)
    return out


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def barrier_and_max(elapsed: float, device=None) -> float:
    """MAX over ranks of a wall-clock duration (identity when not distributed)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(elapsed)
    t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
