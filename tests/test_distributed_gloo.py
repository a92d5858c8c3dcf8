"""N > 1 path on CPU: world_size 2 over gloo, each rank drives one shard of groups through ShardedDriver with the
one all-gather per migration event.  The compute engine is the CPU oracle injected by the test; the driver code is
the product's (differentialevolutionmcmc.jl_amd/distributed.py).  Result must equal the single-shard run bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    rng = np.random.default_rng(4)
    return rng.normal(size=25), np.stack([rng.normal(size=48), rng.uniform(0.5, 2, 48)], 1)


def _mk(orc, ng, off, total):
    data, _ = _problem()
    o = orc.Oracle(n_groups=ng, Np=6, D=2, n_rows=40, schedule=2, seed=99, alpha=0.4, group_offset=off,
                   n_groups_total=total, theta_snooker=0.1)
    o.set_model(0, data, [25])
    o.set_priors([1, 2], [0, 0], [10, 1])
    o.set_bounds([-np.inf, 0], [np.inf, np.inf])
    return o


def _worker(rank, world, port, q, async_migration=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from oracle import oracle as orc
    from demc_amd.distributed import ShardedDriver, gather_history
    dist.init_process_group("gloo", rank=rank, world_size=world)
    G = 8 // world
    eng = _mk(orc, G, rank * G, 8)
    _, th0 = _problem()
    eng.set_state(th0[rank * G * 6:(rank + 1) * G * 6])
    drv = ShardedDriver(eng, dist, async_migration=async_migration)
    drv.step(1, 25)
    drv.step(26, 15)  # split call: iteration numbering must carry over
    hist = gather_history(drv, 0, 40)
    st = eng.get_state()
    if rank == 0:
        q.put((hist, drv.n_exchanges))
    q.put(("state", rank, st))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("async_migration", [False, True])
def test_world2_equals_single_shard(orc, async_migration):
    """async_migration: the groups an exchange does not select update while the all-gather is in flight;
    same bits either way"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, async_migration)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(3)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    hist, n_ex = next(g for g in got if g[0] != "state")
    states = {g[1]: g[2] for g in got if g[0] == "state"}
    one = _mk(orc, 8, 0, 8)
    one.set_state(_problem()[1])
    one.step(1, 40)
    assert n_ex == sum(one.migration_due(i) for i in range(1, 41)) and n_ex >= 8
    t1, w1, i1 = one.get_state()
    np.testing.assert_array_equal(t1, np.concatenate([states[0][0], states[1][0]]))
    np.testing.assert_array_equal(w1, np.concatenate([states[0][1], states[1][1]]))
    np.testing.assert_array_equal(i1, np.concatenate([states[0][2], states[1][2]]))
    for a, b in zip(one.get_history(0, 40), hist):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("async_migration", [False, True])
def test_single_shard_driver_matches_step(orc, async_migration):
    from demc_amd.distributed import ShardedDriver
    a, b = _mk(orc, 8, 0, 8), _mk(orc, 8, 0, 8)
    th0 = _problem()[1]
    a.set_state(th0)
    b.set_state(th0)
    a.step(1, 30)
    ShardedDriver(b, async_migration=async_migration).step(1, 30)
    for x, y in zip(a.get_history(0, 30), b.get_history(0, 30)):
        np.testing.assert_array_equal(x, y)
