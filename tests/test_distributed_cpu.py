"""world_size-2 gloo test (CPU) of the N > 1 path: sharding is embarrassingly parallel, the only
collective is the aggregate report (max of the timed region, sums of the statistics)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from crazyflie_nmpc_amd import parallel
    from crazyflie_nmpc_amd.synthetic import sample_hover_x0
    lo, hi = parallel.shard_range(1001, rank, world)
    x0 = sample_hover_x0(np.random.default_rng(parallel.shard_seed(rank)), hi - lo)
    elapsed = 1.0 + rank                       # rank 1 is "slower"
    sums = [hi - lo, float(x0[:, 2].sum()), 3.0 * (rank + 1)]
    t, s = parallel.aggregate_report(elapsed, sums, dist)
    q.put((rank, lo, hi, t, s.tolist(), float(x0[:, 2].sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_aggregate_report():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, t0, s0, z0), (r1, lo1, hi1, t1, s1, z1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 501, 501, 1001)        # disjoint, covering shards
    assert t0 == t1 == 2.0                                      # MAX over ranks
    assert s0 == s1 and s0[0] == 1001 and abs(s0[1] - (z0 + z1)) < 1e-9 and s0[2] == 9.0
    assert z0 != z1                                             # different seeds -> different shards


def _lpt_numpy(horizons, world):
    """the rule cfnmpc_shard_by_horizon states (include/cfnmpc.h), restated: decreasing horizon (stable), each vehicle to
    the shard with the smallest sum N so far, ties to the lowest shard"""
    horizons = np.asarray(horizons)
    order = np.argsort(-horizons, kind="stable")
    load = np.zeros(world)
    out = [[] for _ in range(world)]
    for i in order:
        r = int(np.argmin(load))
        out[r].append(int(i))
        load[r] += horizons[i]
    return [np.array(sorted(ix), dtype=np.int64) for ix in out]


def test_horizon_bucketing_balances_cost():
    from crazyflie_nmpc_amd import parallel
    rng = np.random.default_rng(0)
    horizons = rng.choice([30, 50, 100], size=4096)
    shards = parallel.shard_by_horizon(horizons, 8)
    allidx = np.concatenate(shards)
    assert sorted(allidx.tolist()) == list(range(4096))
    loads = np.array([horizons[ix].sum() for ix in shards])
    assert loads.max() - loads.min() <= 100                    # balanced to one instance
    # config C5 at the metric's size over 8 GPUs: per-rank sum N within 1 % (in fact within one vehicle), every rank holds
    # its eighth of every bucket to within a few vehicles
    hz = np.random.default_rng(1).choice([30, 50, 100], size=65536)
    sh = parallel.shard_by_horizon(hz, 8)
    loads = np.array([hz[ix].sum() for ix in sh], dtype=np.float64)
    assert (loads.max() - loads.min()) / loads.mean() < 1e-3
    for n in (30, 50, 100):
        per = np.array([(hz[ix] == n).sum() for ix in sh])
        assert per.max() - per.min() <= 8, (n, per)
    # the library's partitioner IS the stated rule (also for worlds that do not divide the fleet, and one rank)
    for world in (1, 2, 3, 8):
        for a, b in zip(parallel.shard_by_horizon(horizons[:1001], world), _lpt_numpy(horizons[:1001], world)):
            assert np.array_equal(a, b)
    import pytest
    with pytest.raises(ValueError):
        parallel.shard_by_horizon([30, 0, 50], 2)
    assert parallel.shard_range(10, 3, 4) == (8, 10) and parallel.shard_range(10, 0, 4) == (0, 3)
