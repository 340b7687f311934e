"""bench.py's multi-process bookkeeping on CPU: two ranks over gloo run the timed region with steps of different
length; the reported time must be the slowest rank's, the value the whole job's (weak scaling), and the ranks are
taken from the launcher's environment.  (The HIP solver under two processes: tests/test_gpu_configs.py.)"""
import os
import socket
import sys
import time

import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    import bench
    w, r, lr = bench.dist_env()
    assert (w, r, lr) == (world, rank, rank)
    assert bench.dist_env(same_device=True) == (world, rank, 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def step(timed):
        calls.append(timed)
        time.sleep(0.02 * (rank + 1))          # rank 1 is twice as slow

    elapsed = bench.run_timed(step, steps=5, warmup=2, barrier=dist.barrier)
    assert calls == [False, False] + [True] * 5
    slowest = bench.max_over_ranks(elapsed, dist)
    with open(os.path.join(out_dir, f"r{rank}.txt"), "w") as f:
        f.write(f"{elapsed} {slowest}\n")
    dist.barrier()
    dist.destroy_process_group()


def test_timed_region_takes_the_slowest_rank(tmp_path):
    import torch.multiprocessing as mp
    import bench
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    (e0, s0), (e1, s1) = [tuple(map(float, open(tmp_path / f"r{r}.txt").read().split())) for r in (0, 1)]
    assert s0 == s1 == pytest.approx(max(e0, e1))
    # the barrier on both sides of the timed steps makes every rank see (at least) the slowest rank's 5 x 40 ms
    assert s0 >= 5 * 0.04 * 0.9
    assert bench.job_throughput(2, 1000, 5, s0) == pytest.approx(2 * 1000 * 5 / s0)
    assert bench.max_over_ranks(1.5) == 1.5      # N = 1: no collective


def test_presets_and_flop_counts():
    import bench
    assert set(bench.PRESETS) >= {"C1", "C2", "C3", "C5", "STFT"}
    c2 = bench.PRESETS["C2"]
    assert (c2["bins"], c2["exemplars"], c2["iters"]) == (25, 4096, 100)
    # FACTORED K(4MN+3N) + 2MN + 2MbN = 42.6 Mflop per frame at C2
    assert bench.algorithmic_flops_per_frame(25, 4096, 100, 25, "factored") == 42598400
    assert bench.loop_flops_per_frame(25, 4096, 100, "factored") == 100 * (4 * 25 * 4096 + 3 * 4096)
    assert bench.cpu_sample_frames(4096, 688) == 688 and bench.cpu_sample_frames(16384, 688) == 43


def test_gpus_flag_spawns_a_launcher_child_and_relays_its_status():
    """`python bench.py --gpus N` without a launcher: the parent builds the torch.distributed.run command (one rank
    per GPU, rendezvous on 127.0.0.1), runs it as a child and hands back its exit status; nothing is exec'ed."""
    import bench
    cmd = bench.launcher_command(["--gpus", "4", "--steps", "3"], 4, 29511, python="python")
    assert cmd[:3] == ["python", "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5].endswith("bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    seen = {}

    class Done:
        returncode = 7

    def fake_run(c, env):
        seen["cmd"], seen["env"] = c, env
        return Done()

    assert bench.spawn_ranks(["--gpus", "2"], 2, run=fake_run) == 7
    assert "--nproc-per-node=2" in seen["cmd"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    p = bench.free_port()
    assert 1024 < p < 65536


def test_c4_is_one_set_split_over_the_ranks():
    import bench
    lens = [bench.C4_LENGTHS[i % len(bench.C4_LENGTHS)] for i in range(162)]
    assert sum(lens) == 109206
    for world in (1, 2, 4, 8):
        shards = [bench.shard_of_rank(lens, world, r) for r in range(world)]
        assert sorted(i for s in shards for i in s) == list(range(162))
        loads = [sum(lens[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= max(lens)          # LPT: even to within one utterance
    # strong scaling: the job's value counts the set once, whatever the number of ranks
    assert bench.job_throughput(8, 13000, 5, 2.0, total_units=109206) == pytest.approx(109206 * 5 / 2.0)
    assert bench.job_throughput(8, 1000, 5, 2.0) == pytest.approx(8 * 1000 * 5 / 2.0)
