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


def scene_cost(filename: str) -> float:
    """Predicted cost of a scene before it is loaded.  The GP fits dominate (sum over pairs of M^3, M = training
    superpoints of a pair ~ points per object ~ N / K), and the only thing known without reading the file is its
    size, which is proportional to the point count N: cost ~ N^2 tracks the measured per-scene time of the synthetic
    stream (N^3 / K^2 summed over ~K pairs) much better than N or a scene count.  Missing file -> 0."""
    try:
        return float(os.path.getsize(filename)) ** 2
    except OSError:
        return 0.0


def shard_scenes_lpt(filenames: Sequence[str], rank: int, world: int, costs: Sequence[float] = None) -> List[str]:
    """Longest-processing-time-first shard: scenes sorted by decreasing predicted cost (ties by name) are dealt one by
    one to the rank with the least cost so far.  Deterministic, so every rank computes the same partition by itself;
    every scene belongs to exactly one rank.  The rank's scenes come back largest first, which also keeps the tail
    of its last batch short.  Reference behaviour being sharded: the serial scene loop of gapro/gen_ps.py:36-41."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    names = list(sorted(filenames))
    c = [scene_cost(f) for f in names] if costs is None else [float(v) for v in costs]
    if costs is not None and len(c) != len(names):
        raise ValueError("one cost per scene")
    if costs is not None:  # costs were given in the caller's order
        by_name = dict(zip(filenames, c))
        c = [by_name[f] for f in names]
    order = sorted(range(len(names)), key=lambda i: (-c[i], names[i]))
    load = [0.0] * world
    mine = []
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        load[r] += max(c[i], 1e-30)
        if r == rank:
            mine.append(names[i])
    return mine


class ClaimQueue:
    """Shared work queue over the scene list for the workers of ONE node, without a server: a worker claims a scene
    by creating `<claim_dir>/<scan>.claim` with O_CREAT | O_EXCL (atomic on a local file system), in the common
    cost-sorted order, so the expensive scenes start first and a fast worker simply claims more.  A claim directory
    belongs to one job: the parent creates a fresh one per run and removes it at the end; skip-if-exists on the
    OUTPUT files is what makes a restarted job redo nothing (reference gen_ps.py:39-41)."""

    def __init__(self, filenames: Sequence[str], claim_dir: str, costs: Sequence[float] = None):
        names = list(sorted(filenames))
        c = [scene_cost(f) for f in names] if costs is None else [dict(zip(filenames, costs))[f] for f in names]
        self.order = [names[i] for i in sorted(range(len(names)), key=lambda i: (-c[i], names[i]))]
        self.claim_dir = claim_dir
        self.pos = 0
        os.makedirs(claim_dir, exist_ok=True)

    def remaining(self) -> int:
        """Upper bound of the scenes nobody has claimed yet (scenes other workers claimed since this worker last looked
        are still counted)."""
        return len(self.order) - self.pos

    def guided(self, max_batch: int, n_workers: int, min_batch: int = 16) -> int:
        """Size of the next claim.  One worker: max_batch (a launch amortises its longest fits -- a floor or wall pair of
        M ~ 1000 lasts ~0.3 s whatever else is in the launch -- over as many scenes as it is given: 32-scene batches of
        the train-split mix ran at 90 scenes/s, 256-scene batches at 330).  Several workers: a share of what is left,
        remaining / (2 W), shrinking towards min_batch, so that the workers finish together instead of one of them
        holding the last full batch (guided self-scheduling)."""
        if n_workers <= 1:
            return max_batch
        return max(min_batch, min(max_batch, self.remaining() // (2 * n_workers)))

    def claim(self, n: int) -> List[str]:
        """Up to n scenes nobody has claimed yet (empty list: the queue is drained)."""
        got = []
        while len(got) < n and self.pos < len(self.order):
            fn = self.order[self.pos]
            self.pos += 1
            path = os.path.join(self.claim_dir, fn.split("/")[-1][:12] + ".claim")
            try:
                os.close(os.open(path, os.O_CREAT | os.O_EXCL | os.O_WRONLY))
            except FileExistsError:
                continue
            got.append(fn)
        return got


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


def effective_cpus() -> int:
    """CPUs this process can really use: the affinity mask, capped by the cgroup CPU quota (cpu.max of cgroup v2,
    cfs_quota_us / cfs_period_us of v1).  A container often sees every core of its host (os.cpu_count() = 256 on the
    MI355X boxes of this pool) behind a quota of a few (16 there): threads beyond the quota are not parallelism, they
    are throttling -- tools/host_probe.py shows compute scaling stop at 16 threads and fall beyond."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    return max(1, n)
