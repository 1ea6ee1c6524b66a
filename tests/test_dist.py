"""CPU test of the N>1 path (-m "not gpu"): world_size 2 over gloo, 127.0.0.1 rendezvous.
The data path has no collective (batch shards); what is distributed is the measurement
bracket of bench.py (pffft_amd/sharding.py): barrier, MAX of elapsed, SUM of units — exercised here with a
host-side stand-in step, since the transform itself has no CPU implementation in the product."""
import os
import socket

import numpy as np
import pytest

from pffft_amd.sharding import combine, combine_stats, shard_range, timed_steps


def test_shard_range_partitions_the_batch():
    for total in (0, 1, 7, 8, 1 << 20, (1 << 23) + 5):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def _worker(rank, world, port, total, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    start, count = shard_range(total, rank, world)
    x = np.random.default_rng(rank).standard_normal((count, 64))
    done = [0]

    def step():  # stand-in for one pass over this rank's shard
        np.fft.fft(x, axis=1)
        done[0] += 1

    elapsed = timed_steps(step, steps=3, warmup=1, dist=dist, sync=None)
    mx, tot = combine(elapsed, float(count), dist, torch.device("cpu"))
    q.put((rank, start, count, elapsed, mx, tot, done[0]))
    dist.destroy_process_group()


def test_two_rank_gloo_bracket():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    total = 1001
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, c0, e0, m0, t0, d0), (r1, s1, c1, e1, m1, t1, d1) = res
    assert (s0, c0, s1, c1) == (0, 501, 501, 500)
    assert t0 == t1 == total                      # SUM over ranks of the units
    assert m0 == m1 == pytest.approx(max(e0, e1))  # MAX over ranks of the elapsed time
    assert d0 == d1 == 4                          # 1 warm-up + exactly 3 timed steps


def test_synthetic_input_is_addressed_by_global_index():
    """bench.py's inputs are a counter hash of (seed, GLOBAL element index): a strong-scaling shard holds the same vectors
    whatever the number of ranks, and the torch (device) and numpy (host) generators agree to the bit."""
    import torch
    from pffft_amd.sharding import global_uniform_, global_uniform_np
    vs, total = 96, 1001
    whole = global_uniform_np(0, total * vs, 5, np.float64).reshape(total, vs)
    assert -1.0 <= whole.min() and whole.max() < 1.0 and abs(whole.mean()) < 0.01
    for world in (1, 2, 3, 8):
        for r in range(world):
            start, count = shard_range(total, r, world)
            shard = torch.empty(count, vs, dtype=torch.float64)
            global_uniform_(shard, start * vs, 5, chunk=4099)
            assert np.array_equal(shard.numpy(), whole[start:start + count])
    # float32 carries the same values (multiples of 2^-23), far beyond 2^32 elements too
    a = global_uniform_np((1 << 34) + 5, 4096, 2, np.float32)
    b = global_uniform_(torch.empty(4096, dtype=torch.float32), (1 << 34) + 5, 2)
    assert np.array_equal(a, b.numpy()) and np.array_equal(a.astype(np.float64), global_uniform_np((1 << 34) + 5, 4096, 2, np.float64))
    assert not np.array_equal(a, global_uniform_np((1 << 34) + 5, 4096, 3, np.float32))     # the seed matters


# ------------------------------------------------------------------ bench.py's own launcher
def _bench(*args, env_extra=None, timeout=300):
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                          env=env, timeout=timeout)


def test_bench_gpus_2_spawns_two_ranks_over_gloo():
    """`python bench.py --gpus 2` with no launcher must itself start 2 ranks (torch.distributed.run, 127.0.0.1) and print
    ONE line with n_gpus: 2.  Driven here on CPU over gloo with the harness self-test step (the transform has no CPU
    implementation in the product); on a multi-GPU box the same path runs over RCCL."""
    import json
    p = _bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo", "--selftest-cpu")
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["selftest"] is True
    assert rec["value"] > 0
    # the line reports how many ranks actually met in the all-reduce, and the fastest / slowest rank
    assert rec["ranks_seen"] == 2
    assert 0 < rec["ms_per_step_fastest_rank"] <= rec["ms_per_step_slowest_rank"]


def test_bench_refuses_more_gpus_than_visible():
    """`--gpus 2` on a box that shows fewer than 2 GPUs must fail loudly, not report n_gpus: 1."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("this box really has 2 GPUs")
    p = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert p.returncode != 0
    assert "GPU(s) visible" in (p.stderr + p.stdout)
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_bench_refuses_world_size_mismatch():
    p = _bench("--gpus", "4", "--selftest-cpu", env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=2" in (p.stderr + p.stdout)
