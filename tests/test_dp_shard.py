"""The N>1 path on CPU: request sharding arithmetic and the world_size-2 gloo control group that
bench.py uses for its barrier / max-over-ranks timing (no data-path collective exists)."""
import os
import socket

import pytest
import torch.multiprocessing as mp

from swiftllm_amd import dp


@pytest.mark.parametrize("n,world", [(256, 8), (32, 1), (10, 4), (3, 8), (0, 2), (257, 8)])
def test_shard_bounds_partition_the_requests(n, world):
    seen = []
    sizes = []
    for r in range(world):
        b, e = dp.shard_bounds(n, r, world)
        assert 0 <= b <= e <= n
        seen.extend(range(b, e))
        sizes.append(e - b)
    assert seen == list(range(n))               # disjoint, ordered, complete
    assert max(sizes) - min(sizes) <= 1         # balanced
    if n == 256 and world == 8:
        assert sizes == [32] * 8                # BASELINE configs[4]: 256 sequences, 32 per GPU


def test_shard_bounds_rejects_bad_rank():
    with pytest.raises(ValueError):
        dp.shard_bounds(10, 2, 2)


def test_least_loaded_routing():
    assert dp.least_loaded([5, 3, 3, 9]) == 1
    assert dp.least_loaded([0]) == 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    assert dp.init_control_group(timeout_s=60)
    mine = dp.shard(list(range(7)), rank, world)
    dp.barrier()
    units, secs = dp.reduce_job(len(mine) * 10, 1.0 + rank)      # rank 1 is the slow one
    everything = dp.gather_lists(mine)
    dp.barrier()
    q.put((rank, mine, units, secs, everything))


def test_two_rank_gloo_control_group():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, m0, u0, s0, e0), (r1, m1, u1, s1, e1) = results
    assert m0 == [0, 1, 2, 3] and m1 == [4, 5, 6]
    assert u0 == u1 == 70.0         # whole-job units = sum over ranks
    assert s0 == s1 == 2.0          # whole-job time = the slowest rank
    assert e0 == e1 == [[0, 1, 2, 3], [4, 5, 6]]


def test_replica_core_sets_are_disjoint_and_cover_the_allowed_cores():
    """Per-replica CPU pinning: whatever the host topology says, the N replicas of a node get
    disjoint, non-empty core sets drawn from the allowed cores."""
    from swiftllm_amd import dp as _dp
    allowed = list(range(3, 67))
    for world in (1, 2, 4, 8):
        sets = [_dp.cpus_for_local_rank(r, world, allowed) for r in range(world)]
        assert all(s for s in sets)
        flat = [c for s in sets for c in s]
        assert set(flat) <= set(allowed)
        if world > 1:
            assert len(flat) == len(set(flat))
    assert _dp._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]


def test_spawn_local_ranks_gives_every_rank_its_gpu_and_a_working_control_group(tmp_path, capfd):
    """What `bench.py --gpus N` does without a launcher (dp.spawn_local_ranks): N processes with the torchrun
    environment, rank i pinned to GPU i, a gloo control group that works, rank 0's stdout passed through, the worst
    exit code returned."""
    import sys
    import textwrap
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        from swiftllm_amd import dp
        rank, local_rank, world = dp.env_rank_world()
        assert dp.init_control_group(timeout_s=60)
        units, secs = dp.reduce_job(10 * (rank + 1), 1.0 + rank)
        dp.barrier()
        print("RANK", rank, world, os.environ["HIP_VISIBLE_DEVICES"], os.environ["LOCAL_WORLD_SIZE"], units, secs, flush=True)
        sys.exit(int(os.environ.get("FAIL_RANK", "-1")) == rank)
    """ % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    env_keys = ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "WORLD_SIZE", "RANK")
    saved = {k: os.environ.pop(k) for k in env_keys if k in os.environ}
    try:
        assert dp.spawn_local_ranks([sys.executable, str(script)], 2, timeout_s=120) == 0
        out, err = capfd.readouterr()
        assert "RANK 0 2 0 2 30.0 2.0" in out and "RANK 1" not in out      # rank 0 owns stdout
        assert "RANK 1 2 1 2 30.0 2.0" in err
        os.environ["FAIL_RANK"] = "1"
        assert dp.spawn_local_ranks([sys.executable, str(script)], 2, timeout_s=120) == 1
    finally:
        os.environ.pop("FAIL_RANK", None)
        os.environ.update(saved)
