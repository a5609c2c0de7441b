"""The N>1 path on CPU: two gloo processes shard a scene list, write their outputs, resume, and reduce
the wall time with MAX -- the same helpers bench.py and gen_ps use with RCCL on the GPU box."""
import os
import subprocess
import sys
import textwrap

from gapro_amd.dist_utils import pending_scenes, shard_scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shards_partition_the_scene_list():
    names = ["train/scene%04d_00_inst_nostuff.pth" % i for i in range(1201)][::-1]
    for world in (1, 2, 4, 8):
        shards = [shard_scenes(names, r, world) for r in range(world)]
        flat = sorted(sum(shards, []))
        assert flat == sorted(names)
        assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1


def test_two_rank_gloo_job(tmp_path):
    script = tmp_path / "job.py"
    script.write_text(textwrap.dedent("""
        import os, sys, time
        sys.path.insert(0, %r)
        import torch, torch.distributed as dist
        from gapro_amd.dist_utils import barrier_and_max, env_rank_world, pending_scenes, shard_scenes
        rank, world, _ = env_rank_world()
        dist.init_process_group("gloo")
        out = sys.argv[1]
        names = ["train/scene%%04d_00_inst_nostuff.pth" %% i for i in range(11)]
        mine = pending_scenes(shard_scenes(names, rank, world), out)
        dist.barrier()
        t0 = time.perf_counter()
        for fn in mine:
            with open(os.path.join(out, fn.split("/")[-1][:12] + ".pth"), "w") as f:
                f.write(str(rank))
        time.sleep(0.05 * (rank + 1))
        dist.barrier()
        local = time.perf_counter() - t0
        worst = barrier_and_max(0.05 * (rank + 1))
        assert abs(worst - 0.05 * world) < 1e-12, worst
        n = torch.tensor([len(mine)]); dist.all_reduce(n)
        if rank == 0:
            print("TOTAL", int(n.item()), flush=True)
        dist.destroy_process_group()
    """) % ROOT)
    out = tmp_path / "labels"
    out.mkdir()
    (out / "scene0003_00.pth").write_text("done earlier")  # resume: must be skipped
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", str(script), str(out)]
    res = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "TOTAL 10" in res.stdout
    files = sorted(os.listdir(out))
    assert len(files) == 11
    assert (out / "scene0003_00.pth").read_text() == "done earlier"
    owners = {f: (out / f).read_text() for f in files if f != "scene0003_00.pth"}
    assert set(owners.values()) == {"0", "1"}
    assert pending_scenes(["train/scene%04d_00_inst_nostuff.pth" % i for i in range(11)], str(out)) == []


def _bench(*argv, env=None, timeout=300):
    e = dict(os.environ if env is None else env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        if env is None:
            e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True,
                          text=True, env=e, timeout=timeout)


def test_bench_gpus_n_starts_n_ranks_itself():
    """`python bench.py --gpus 2` with no launcher in the environment must start two ranks (round 1 parsed --gpus
    and ran one rank): the control path of the N > 1 bench -- child torch.distributed.run, rendezvous on 127.0.0.1,
    barrier + MAX over ranks, ONE JSON line from rank 0 -- driven here over gloo with the device pipeline stubbed
    (--dry-run); ms_per_step is the slower rank's sleep."""
    import json

    res = _bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run")
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["dry_run"] is True and rec["steps"] == 3 and rec["value"] is None
    assert rec["ms_per_step"] >= 4.0  # rank 1 sleeps 4 ms per step, rank 0 only 2: the MAX must win
    # with --gpus N the line carries `gen_ps_farm` -- N workers through `gen_ps --devices` over
    # scene FILES (here: --dry_run workers, the host side only; on the GPU box the real ones), two passes, start-up apart
    farm = rec["gen_ps_farm"]
    assert farm["workers"] == 2 and farm["devices"] == "0,1" and farm["dry_run"] is True and farm["scene_files"] == 6
    assert len(farm["passes"]) == 2
    for p in farm["passes"]:
        assert p["exit_status"] == 0 and p["scenes"] == 6 and p["written_files"] == 6 and p["failed"] == 0
        assert len(p["per_worker_scenes"]) == 2 and sum(p["per_worker_scenes"]) == 6
        assert p["file_io"].startswith("native") and p["loader_processes"] == 0
        assert "startup_s" in p and "wall_s" in p and p["scenes_per_s"] > 0


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = _bench("--gpus", "2", "--dry-run", env=env)
    assert res.returncode == 2 and "WORLD_SIZE=1" in res.stderr


def test_lpt_shard_balances_predicted_cost_and_partitions_the_list():
    import numpy as np

    from gapro_amd.dist_utils import shard_scenes_lpt

    rng = np.random.default_rng(3)
    names = ["train/scene%04d_00_inst_nostuff.pth" % i for i in range(1201)]
    # config 3's size distribution: N ~ logN(150k, 0.5) clipped, cost ~ N^2
    cost = np.clip(150000 * np.exp(0.5 * rng.standard_normal(len(names))), 40000, 450000) ** 2
    for world in (1, 2, 4, 8):
        shards = [shard_scenes_lpt(names, r, world, costs=cost) for r in range(world)]
        assert sorted(sum(shards, [])) == sorted(names)
        by = dict(zip(names, cost))
        loads = np.array([sum(by[f] for f in sh) for sh in shards])
        assert loads.max() / loads.mean() < 1.001  # LPT on 1201 items: essentially perfect
        rr = np.array([sum(by[f] for f in sorted(names)[r::world]) for r in range(world)])
        assert loads.max() <= rr.max() + 1e-9  # never worse than the round-robin shard it replaces
        if world > 1:
            first = [by[sh[0]] for sh in shards]
            assert min(first) >= np.sort(cost)[-world]  # every rank starts with one of the `world` largest scenes


def test_claim_queue_hands_every_scene_to_exactly_one_worker(tmp_path):
    """Two processes drain one ClaimQueue concurrently (O_EXCL claim files): disjoint, complete, cost-sorted."""
    script = tmp_path / "claim.py"
    script.write_text(textwrap.dedent("""
        import json, sys, time
        sys.path.insert(0, %r)
        from gapro_amd.dist_utils import ClaimQueue
        names = ["train/scene%%04d_00_inst_nostuff.pth" %% i for i in range(200)]
        q = ClaimQueue(names, sys.argv[1], costs=[(i * 7919) %% 200 for i in range(200)])
        got = []
        while True:
            c = q.claim(3)
            if not c:
                break
            got += c
            time.sleep(0.001)
        print(json.dumps(got))
    """) % ROOT)
    cdir = str(tmp_path / "claims")
    ps = [subprocess.Popen([sys.executable, str(script), cdir], stdout=subprocess.PIPE, text=True) for _ in range(2)]
    import json

    got = [json.loads(p.communicate(timeout=120)[0]) for p in ps]
    assert all(p.returncode == 0 for p in ps)
    assert len(got[0]) + len(got[1]) == 200 and not (set(got[0]) & set(got[1]))
    assert min(len(got[0]), len(got[1])) > 0
    cost = {"train/scene%04d_00_inst_nostuff.pth" % i: (i * 7919) % 200 for i in range(200)}
    for g in got:  # each worker sees the common order: non-increasing cost
        c = [cost[f] for f in g]
        assert c == sorted(c, reverse=True)


def test_bench_counts_overlapped_launch_time_once():
    """bench.py's launch duration: union of the launches' [first start, last end] intervals / number of launches."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    assert bench.union_length([]) == 0.0
    assert bench.union_length([(0.0, 10.0)]) == 10.0
    assert bench.union_length([(0.0, 10.0), (12.0, 20.0)]) == 18.0           # a gap is not launch time
    assert bench.union_length([(5.0, 20.0), (0.0, 10.0)]) == 20.0            # overlap counted once, any order
    assert bench.union_length([(0.0, 30.0), (5.0, 10.0), (29.0, 31.0)]) == 31.0
