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


def test_horizon_bucketing_balances_cost():
    from crazyflie_nmpc_amd import parallel
    rng = np.random.default_rng(0)
    horizons = rng.choice([30, 50, 100], size=4096)
    shards = parallel.shard_by_horizon(horizons, 8)
    allidx = np.concatenate(shards)
    assert sorted(allidx.tolist()) == list(range(4096))
    loads = np.array([horizons[ix].sum() for ix in shards])
    assert loads.max() - loads.min() <= 100                    # balanced to one instance
    assert parallel.shard_range(10, 3, 4) == (8, 10) and parallel.shard_range(10, 0, 4) == (0, 3)
