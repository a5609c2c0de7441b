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
