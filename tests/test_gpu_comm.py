"""The one collective behind the C-ABI (include/demc.h "The one collective"): migration! (migration.jl:11-19,
called from step!/pstep!, main.jl:85,103) with groups sharded over GPUs, reachable by a host that has neither Python nor
torch -- demc_comm_init / demc_step for one process per GPU, demc_create_multi / demc_multi_step for one host thread.

A 1-GPU box can run RCCL at world size 1 only (two ranks on one device are refused), so the communicator path is checked at
world 1 against demc_step's own on-device migration, and the rank logic (offsets, who receives whose candidate) through a
two-shard set whose shards share the device."""
import os
import subprocess

import numpy as np
import pytest

from conftest import make_problem, setup_engine

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def D():
    import demc_amd
    return demc_amd


def _run_single(D, prob, G, Np, n_it, th0, **kw):
    e = D.HipEngine(n_groups=G, Np=Np, D=prob["D"], n_rows=n_it, schedule=2, **kw)
    setup_engine(e, prob)
    e.set_state(th0)
    e.step(1, n_it)
    out = e.get_history(0, n_it) + e.get_state()
    e.close()
    return out


@pytest.mark.parametrize("family,kw", [("mvn_full", dict(loglike_mode=1)), ("mvn_full", dict(loglike_mode=0)), ("gaussian", dict()),
                                       ("hier_binomial", dict())])
def test_library_communicator_at_world_one_equals_demc_step(D, family, kw):
    """demc_comm_unique_id -> demc_comm_init -> demc_step: pack -> ncclAllGather on the handle's stream -> apply, through
    the C-ABI alone (no torch.distributed anywhere), reproduces demc_step's on-device migration bit for bit; so does the
    overlapped form (all-gather on the side stream, unselected groups updating meanwhile), and demc_migration_exchange +
    demc_update driven from the host."""
    prob = make_problem(family, np.random.default_rng(71), N=300, d=6, S=300)
    G, Np, n_it = 10, 8, 40
    th0 = prob["init"](G * Np)
    cfg = dict(seed=29, alpha=0.5, burnin=10, trace=0, **kw)
    ref = _run_single(D, prob, G, Np, n_it, th0, **cfg)
    for form in ("step", "overlap", "host_loop"):
        e = D.HipEngine(n_groups=G, Np=Np, D=prob["D"], n_rows=n_it, schedule=2, **cfg)
        setup_engine(e, prob)
        e.set_state(th0)
        e.comm_init(D.HipEngine.comm_unique_id(), 0, 1)
        if form == "overlap":
            e.comm_set_overlap(True)
        if form == "host_loop":
            for it in range(1, n_it + 1):
                if e.migration_due(it):
                    e.migration_exchange_enqueue(it)
                e.update(it, 1)
        else:
            e.step(1, n_it)
        st = e.comm_stats()
        assert st["world"] == 1 and st["rank"] == 0 and st["exchanges"] >= 10
        out = e.get_history(0, n_it) + e.get_state()
        e.comm_destroy()
        e.close()
        for i, (x, y) in enumerate(zip(ref, out)):
            if form == "overlap" and kw.get("loglike_mode") == 0 and i in (2, 5):
                # STREAMING on a small population: demc_step runs the streaming-resident form, the subset updates of the
                # overlapped exchange the K1 -> K2 -> K3 chain -- the observation sums are split differently (log-densities
                # to rounding, every draw and decision identical; include/demc.h, demc_comm_set_overlap)
                np.testing.assert_allclose(x, y, rtol=1e-10)
            else:
                assert np.array_equal(x, y), f"{form}: array {i}"


def test_history_partners_never_let_a_subset_of_groups_run_ahead(D):
    """DE-MC_Z draws its partners from the history rows of EVERY group, so a subset of the groups may not advance by more than one
    iteration before the others follow: demc_update_groups_async refuses n_iters > 1, and the overlapped exchange (which lets the
    groups a migration did not select run a whole stretch of iterations first) falls back to the plain stream-ordered one --
    with demc_comm_set_overlap(1) a world-1 run reproduces demc_step bit for bit (it read unwritten history rows before)."""
    from demc_amd import workloads as W
    w = W.cfg3(N=600, d=8, G=10, Np=16)
    G, Np, n_init, n_it = 10, 16, 4, 30
    rng = np.random.default_rng(8)
    rows0 = np.stack([w["init"](G * Np, rng) for _ in range(n_init)])
    th0 = w["init"](G * Np, rng)
    outs = []
    for overlap in (False, True):
        e = D.HipEngine(n_groups=G, Np=Np, D=w["D"], n_rows=n_init + n_it, schedule=1, seed=31, alpha=0.5, burnin=10, trace=0,
                        loglike_mode=1, partner_kind=1, n_initial=n_init, **w["engine"])
        W.configure(e, w)
        e.set_history_rows(0, rows0)
        e.set_state(th0)
        e.comm_init(D.HipEngine.comm_unique_id(), 0, 1)
        e.comm_set_overlap(overlap)
        if not overlap:
            with pytest.raises(D.DemcError) as err:
                e.update_groups_enqueue(1 + n_init, 2, [0, 1, 2])
            assert err.value.code == D._ffi.EINVAL and "history partners" in str(err.value)
        e.step(1 + n_init, n_it)
        assert e.comm_stats()["exchanges"] >= 8
        with pytest.raises(D.DemcError) as err:  # iteration t reads rows 1:(t-1): none of them may lie behind the history buffer
            e.step(1 + n_init + n_it, 2)
        assert err.value.code == D._ffi.EINVAL and "n_rows" in str(err.value)
        outs.append(e.get_history(n_init, n_init + n_it) + e.get_state())
        e.comm_destroy()
        e.close()
    for i, (x, y) in enumerate(zip(*outs)):
        assert np.array_equal(x, y), f"array {i}"


def test_communicator_shape_is_checked_and_a_sharded_step_needs_one(D):
    prob = make_problem("gaussian", np.random.default_rng(72))
    e = D.HipEngine(n_groups=4, Np=6, D=2, n_rows=8, schedule=2, seed=3, alpha=1.0, group_offset=4, n_groups_total=8)
    setup_engine(e, prob)
    e.set_state(prob["init"](24))
    with pytest.raises(D.DemcError) as err:  # migration due at once (alpha = 1) and nobody to exchange with
        e.step(1, 2)
    assert err.value.code == D._ffi.EINVAL and "communicator" in str(err.value)
    uid = D.HipEngine.comm_unique_id()
    assert len(uid) == 128 and uid != D.HipEngine.comm_unique_id()
    for rank, world in ((0, 2), (1, 3), (2, 2), (0, 1)):  # offset 4 of 8 groups is rank 1 of 2, nothing else
        with pytest.raises(D.DemcError) as err:
            e.comm_init(uid, rank, world)
        assert err.value.code == D._ffi.EINVAL
    with pytest.raises(D.DemcError):
        e.comm_set_overlap(True)  # no communicator yet
    with pytest.raises(D.DemcError):
        e.migration_exchange(1)
    np.testing.assert_array_equal(e.comm_allreduce([1.5, -2.0], "max"), [1.5, -2.0])  # one rank: the values stand
    e.close()


@pytest.mark.parametrize("n_shards", [1, 2, 4])
def test_multi_set_equals_one_handle(D, n_shards):
    """demc_create_multi / demc_multi_step: one host thread, n shards (here sharing the one device: rows change hands by
    event-ordered device-to-device copies), every shard sized with the geometry of the whole population: history and
    state of the set == the single handle of all groups, bit for bit -- offsets, the circular shift across shard borders
    and the stream ordering of pack / exchange / apply / update included."""
    prob = make_problem("mvn_full", np.random.default_rng(73), N=400, d=8)
    G, Np, n_it = 16, 16, 50
    th0 = prob["init"](G * Np)
    cfg = dict(seed=31, alpha=0.4, burnin=12, trace=0, loglike_mode=1, theta_snooker=0.1)
    ref = _run_single(D, prob, G, Np, n_it, th0, geometry_groups=G, **cfg)
    m = D.MultiEngine(n_shards, device_ids=[0] * n_shards, n_groups=G, Np=Np, D=prob["D"], n_rows=n_it, schedule=2, **cfg)
    m.each(lambda e: setup_engine(e, prob))
    m.set_state(th0)
    m.step(1, 20)
    m.step(21, n_it - 20)
    out = m.get_history(0, n_it) + m.get_state()
    if n_shards > 1:
        assert all(e.comm_stats()["exchanges"] >= 10 for e in m.shards)
        with pytest.raises(D.DemcError):
            m.shards[0].step(1, n_it)  # a shard cannot run a migration on its own
    m.close()
    for i, (x, y) in enumerate(zip(ref, out)):
        assert np.array_equal(x, y), f"array {i}"


def test_eight_shards_of_256_groups_equal_one_handle(D):
    """the shape of an 8-GPU run of BASELINE cfg3 -- eight shards of 256 groups, 2048 groups in the migration exchange buffer, the
    circular shift crossing every shard border -- as a set on ONE device against the single handle of all 2048 groups, bit for
    bit (small rows and N so that it takes seconds).  The first real 8-GPU run should not also be the first run of this size."""
    prob = make_problem("mvn_full", np.random.default_rng(77), N=200, d=4)
    G, Np, n_it = 2048, 6, 30
    th0 = prob["init"](G * Np)
    cfg = dict(seed=41, alpha=0.5, burnin=10, trace=0, loglike_mode=1)
    ref = _run_single(D, prob, G, Np, n_it, th0, geometry_groups=G, **cfg)
    m = D.MultiEngine(8, device_ids=[0] * 8, n_groups=G, Np=Np, D=prob["D"], n_rows=n_it, schedule=2, **cfg)
    m.each(lambda e: setup_engine(e, prob))
    m.set_state(th0)
    m.step(1, n_it)
    out = m.get_history(0, n_it) + m.get_state()
    assert all(e.comm_stats()["exchanges"] >= 8 for e in m.shards)
    m.close()
    for i, (x, y) in enumerate(zip(ref, out)):
        assert np.array_equal(x, y), f"array {i}"
    ids = out[-1]
    assert sorted(ids.tolist()) == list(range(G * Np)), "ids are a permutation: nothing was lost or duplicated across the shard borders"


def test_cfg4_sharded_over_eight_equals_one_handle(D):
    """BASELINE cfg4 as it is meant to run -- 128 groups of the hierarchical Binomial model over 8 GPUs, 16 groups each, two block
    sweeps per iteration, migration all-gathers in between -- as an 8-shard set on one device (rows of 2 102 scalars) against
    the single handle: the row-streaming kernel is chosen (over k_longrow) on the particle count of the WHOLE population in every shard
    (geometry_groups), so state and history agree bit for bit."""
    from demc_amd import workloads as W
    w = W.cfg4(S=2100, G=128, Np=32)
    G, Np, n_it = 128, 32, 6
    th0 = w["init"](G * Np, np.random.default_rng(78))
    cfg = dict(seed=43, alpha=0.5, burnin=3, trace=0)

    def run(make):
        e = make()
        (e.each if hasattr(e, "each") else (lambda f: f(e)))(lambda x: W.configure(x, w))
        e.set_state(th0)
        e.step(1, n_it)
        out = e.get_history(0, n_it) + e.get_state()
        kern = (e.shards[0] if hasattr(e, "shards") else e).last_kernels()
        e.close()
        return out, kern

    ref, k1 = run(lambda: D.HipEngine(n_groups=G, Np=Np, D=w["D"], n_rows=n_it, schedule=2, geometry_groups=G, **cfg))
    out, k8 = run(lambda: D.MultiEngine(8, device_ids=[0] * 8, n_groups=G, Np=Np, D=w["D"], n_rows=n_it, schedule=2, **cfg))
    assert k1 == k8 == "k_frozen_sweep<256,big>"  # (the last sweep: the subject block; the hyper-parameter sweep before it is k_frozen_sweep<256>)
    for i, (x, y) in enumerate(zip(ref, out)):
        assert np.array_equal(x, y), f"array {i}"


def test_cfg5_sharded_over_eight_equals_one_handle(D):
    """BASELINE cfg5's partition -- 512 groups of the LBA model (snooker 0.1) over 8 GPUs, 64 groups each -- as an 8-shard set on one
    device against the single handle (500 simulated trials, 16 particles per group): K1 -> k_lba_wave -> k_accept_store with
    the chunk counts of the whole population; ids, accept flags and theta bit for bit, log-densities too (same kernels, same sums)."""
    from demc_amd import workloads as W
    w = W.cfg5(N=500, G=512, Np=16)
    G, Np, n_it = 512, 16, 8
    th0 = w["init"](G * Np, np.random.default_rng(79))
    cfg = dict(seed=47, alpha=0.5, burnin=4, trace=0, **w["engine"])

    def run(e):
        (e.each if hasattr(e, "each") else (lambda f: f(e)))(lambda x: W.configure(x, w))
        e.set_state(th0)
        e.step(1, n_it)
        out = e.get_history(0, n_it) + e.get_state()
        e.close()
        return out

    ref = run(D.HipEngine(n_groups=G, Np=Np, D=w["D"], n_rows=n_it, schedule=2, geometry_groups=G, **cfg))
    out = run(D.MultiEngine(8, device_ids=[0] * 8, n_groups=G, Np=Np, D=w["D"], n_rows=n_it, schedule=2, **cfg))
    for i, (x, y) in enumerate(zip(ref, out)):
        if i in (2, 5):   # lp history / weights: the observation chunks are chosen from the launch's proposal count, which a shard
            np.testing.assert_allclose(x, y, rtol=1e-9)  # has an eighth of -- sums split differently, values equal to rounding
        else:
            assert np.array_equal(x, y), f"array {i}"


@pytest.mark.parametrize("snooker", [0.0, 0.1])
def test_multi_set_streaming_shards_share_a_device(D, snooker):
    """two 40-group shards of an 80-group STREAMING population on ONE device.  Sized from its own 40 groups a
    shard would cut the observation tiles into C = 4 chunks (the unsharded run: C = 2) and launch 160 spin-waiting
    workgroups next to its sibling's 160 on 256 CUs.  Chunking and kernel form now come from the whole population
    (geometry_groups) and shards of one device share a stream: the set equals the single handle bit for bit, in the lean
    streaming-resident kernel (default sampler) and in the general one (snooker)."""
    prob = make_problem("mvn_full", np.random.default_rng(75), N=1600, d=8)
    G, Np, n_it = 80, 16, 24
    th0 = prob["init"](G * Np)
    cfg = dict(seed=37, alpha=0.3, burnin=10, trace=0, loglike_mode=0, theta_snooker=snooker)
    e = D.HipEngine(n_groups=G, Np=Np, D=prob["D"], n_rows=n_it, schedule=2, geometry_groups=G, **cfg)
    setup_engine(e, prob)
    e.set_state(th0)
    e.step(1, n_it)
    lk = e.last_kernels()  # the streaming-resident form ran: k_res_mvn<256,true,8> (default sampler) or k_propose<...,true> (snooker)
    assert (lk.startswith("k_res_mvn<") and ",true," in lk) or (lk.startswith("k_propose<") and lk.endswith(",true>")), lk
    ref = e.get_history(0, n_it) + e.get_state()
    ref_kernels = e.last_kernels()
    e.close()
    m = D.MultiEngine(2, device_ids=[0, 0], n_groups=G, Np=Np, D=prob["D"], n_rows=n_it, schedule=2, **cfg)
    m.each(lambda s: setup_engine(s, prob))
    m.set_state(th0)
    m.step(1, 10)
    m.step(11, n_it - 10)
    assert m.shards[0].last_kernels() == ref_kernels
    with pytest.raises(D.DemcError):
        m.shards[1].comm_destroy()  # a shard's communicator belongs to the set
    out = m.get_history(0, n_it) + m.get_state()
    m.close()
    for i, (x, y) in enumerate(zip(ref, out)):
        assert np.array_equal(x, y), f"array {i}"


def test_multi_set_rejects_bad_shapes(D):
    with pytest.raises(D.DemcError) as err:
        D.MultiEngine(3, device_ids=[0, 0, 0], n_groups=8, Np=6, D=2)
    assert err.value.code == D._ffi.EINVAL and "divide" in str(err.value)
    with pytest.raises(D.DemcError):
        D.MultiEngine(2, device_ids=[0, 0], n_groups=8, Np=6, D=2, group_offset=2)
    # DE-MC_Z draws partner cells from the history of ALL particles (crossover.jl:113-124); a shard holds its own groups' history:
    # a sharded set cannot reproduce the single handle it stands for, so the combination is refused -- one shard is fine
    with pytest.raises(D.DemcError) as err:
        D.MultiEngine(2, device_ids=[0, 0], n_groups=8, Np=6, D=2, n_rows=8, n_initial=2, schedule=1, partner_kind=1)
    assert err.value.code == D._ffi.EUNSUPPORTED and "history" in str(err.value)
    one = D.MultiEngine(1, device_ids=[0], n_groups=8, Np=6, D=2, n_rows=8, n_initial=2, schedule=1, partner_kind=1)
    # the streams of a built set belong to the set (shards on one device borrow the first one's): re-pointing one is refused
    two = D.MultiEngine(2, device_ids=[0, 0], n_groups=8, Np=6, D=2)
    for e in two.shards:
        with pytest.raises(D.DemcError) as err:
            e.set_stream(None)
        assert err.value.code == D._ffi.EINVAL and "belong to the set" in str(err.value)
    two.close()
    one.close()


def test_sharded_driver_with_the_library_collective(D):
    """ShardedDriver(collective="library"): torch.distributed (gloo: no second RCCL communicator) only carries the id"""
    import torch
    import torch.distributed as dist
    from demc_amd.distributed import ShardedDriver
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        prob = make_problem("mvn_full", np.random.default_rng(74), N=200, d=6)
        G, Np, n_it = 6, 16, 40
        th0 = prob["init"](G * Np)
        cfg = dict(seed=17, alpha=0.4, burnin=10, trace=0, loglike_mode=1)
        ref = _run_single(D, prob, G, Np, n_it, th0, **cfg)
        for overlap in (False, True):
            e = D.HipEngine(n_groups=G, Np=Np, D=prob["D"], n_rows=n_it, schedule=2, **cfg)
            setup_engine(e, prob)
            e.set_state(th0)
            drv = ShardedDriver(e, dist, torch.device("cuda", 0), async_migration=overlap, collective="library")
            drv.step(1, n_it)
            drv.synchronize()
            assert drv.n_exchanges >= 5
            out = e.get_history(0, n_it) + e.get_state()
            e.close()
            for x, y in zip(ref, out):
                assert np.array_equal(x, y)
    finally:
        dist.destroy_process_group()


def test_plain_c_multi_rank_driver(tmp_path):
    """tools/demc_cdriver.c --ranks 1: fork before the GPU is touched, id through a pipe, demc_comm_init, demc_step with
    the exchange inside, demc_comm_allreduce for the posterior mean -- a host with neither Python nor MPI"""
    libdir = os.path.join(ROOT, "differentialevolutionmcmc.jl_amd")
    exe = str(tmp_path / "demc_cdriver")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "demc_cdriver.c"), "-o", exe,
                           "-L", libdir, "-ldemc_hip", "-lm", "-Wl,-rpath-link,/opt/rocm/lib"])
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    for extra in ([], ["--overlap"]):
        out = subprocess.run([exe, "--ranks", "1"] + extra, env=env, capture_output=True, text=True, timeout=180)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "C-ABI multi-rank driver OK" in out.stdout and "world 1" in out.stdout


def test_bench_under_the_drivers_launcher_with_the_library_collective():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port P bench.py --gpus 1`
    with DEMC_FORCE_DIST=1: the agent's store carries the communicator id, the engine's own RCCL communicator does the
    exchanges (world 1 is all a 1-GPU box can run), barrier and max-over-ranks go through demc_comm_allreduce -- the JSON
    line comes out with the collective named and all-gathers counted"""
    import json
    import socket
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, DEMC_FORCE_DIST="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "30", "--warmup", "10",
                          "--n-groups", "16", "--nobs", "4000", "--dim", "8", "--accuracy-iters", "0", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, env=env)  # (no --np: the launcher's parser claims it as an abbreviation)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    r, line = json.loads(lines[0])["detail"], json.loads(lines[-1])
    assert "demc_comm_init" in r["config"]["collective"] and r["config"]["all_gathers_rank0"] >= 1 and r["n_gpus"] == 1
    # the multi-rank line is the same compact one: what the first N > 1 run prints cannot outgrow the harness either
    assert len(lines[-1]) < 6000 and line["config"]["rccl_nranks"] == 1 and line["config"]["all_gathers_per_rank"][0] >= 1
    assert line["config"]["ms_per_step_min_over_ranks"] <= line["config"]["ms_per_step_max_over_ranks"]
    rec = json.load(open(os.path.join(ROOT, "gpurun_out", "rank0.json")))  # written before the first collective after the timed region
    assert rec["world"] == 1 and rec["all_gathers"] == line["config"]["all_gathers_per_rank"][0] and rec["rccl_nranks"] == 1


@pytest.mark.parametrize("collective", ["library", "torch"])
def test_bench_sharded_rows_of_the_8_gpu_configs_at_world_one(collective):
    """What `bench.py --gpus N` adds at N > 1 -- BASELINE's cfg4 (128 groups) and cfg5 (512 groups) partitioned over the ranks, one
    library all-gather per migration on the row engine's own communicator -- exercised on the one GPU there is: the driver's
    launcher, DEMC_FORCE_DIST=1, world 1 (each row is then the config's whole population on this GPU).  The rows are in the compact
    last line and in the rank file (written before the reduction that follows a row's timed region)."""
    import json
    import socket
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, DEMC_FORCE_DIST="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "3",
                          "--accuracy-iters", "0", "--no-cpu-baseline", "--collective", collective], capture_output=True, text=True, timeout=900,
                         env=env)  # (torch: the road the run takes when the library's communicator cannot be created -- same rows)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    det, line = json.loads(lines[0])["detail"], json.loads(lines[-1])
    nranks = 1 if collective == "library" else None  # (rccl_nranks is the LIBRARY communicator's size)
    assert len(lines[-1]) < 6000 and line["config"]["rccl_nranks"] == nranks
    rows = {r["name"]: r for r in det["rows"]}
    assert list(rows) == ["cfg4_sharded", "cfg5_sharded"], [(r["name"], r.get("error"), r.get("skipped")) for r in det["rows"]]
    for name, total, kern in (("cfg4_sharded", 128, "k_frozen_sweep"), ("cfg5_sharded", 512, "k_lba_wave")):
        r = rows[name]
        assert "error" not in r and r["value"] > 0 and r["finite_weights"] and r["scaling"] == "strong"
        assert r["groups_total"] == total == r["groups_per_rank"] * r["n_gpus"] and r["rccl_nranks"] == nranks and r["collective"] == collective
        assert r["all_gathers_per_rank"][0] >= 1, "50 iterations at alpha = 0.1 without a migration: the row measured no exchange"
        assert kern in r["kernels"]
    assert [c["name"] for c in line["rows"]] == ["cfg4_sharded", "cfg5_sharded"] and line["rows"][0]["all_gathers_per_rank"] == rows["cfg4_sharded"]["all_gathers_per_rank"]
    rec = json.load(open(os.path.join(ROOT, "gpurun_out", "rank0.json")))
    assert [x["name"] for x in rec["rows"]] == ["cfg4_sharded", "cfg5_sharded"] and rec["rows"][0]["all_gathers"] == rows["cfg4_sharded"]["all_gathers_per_rank"][0]


def test_a_shard_with_history_partners_says_that_its_pool_is_local(D):
    """DE-MC_Z under sharding is a documented deviation: `resample` (crossover.jl:116-124) draws from the
    history of all particles, a shard from its own groups' history.  The one-process-per-GPU road says so at run time: demc_create
    leaves a note in demc_last_error and the Python host turns it into a warning; an unsharded handle, or a sharded one with
    partners from the current population, says nothing."""
    import warnings
    cfg = dict(Np=8, D=3, n_rows=12, n_initial=2, schedule=1, seed=5)
    with pytest.warns(UserWarning, match="THIS shard's history only"):
        e = D.HipEngine(n_groups=4, n_groups_total=8, group_offset=4, partner_kind=1, **cfg)
    assert e.L.demc_last_error(e.h).decode().startswith("note: sharded handle (4 of 8 groups)")
    e.close()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        D.HipEngine(n_groups=4, partner_kind=1, **cfg).close()
        D.HipEngine(n_groups=4, n_groups_total=8, group_offset=4, partner_kind=0, **cfg).close()
