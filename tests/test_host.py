"""Host-side mirror of src/structs.jl / src/main.jl: constructor defaults and error behaviour, nested-Theta
flattening, names, history re-keying and bundle_samples bookkeeping (test/utility_tests.jl "Discard Burnin").
The engine is the CPU oracle INJECTED by the test (the product has no CPU path)."""
import os
import sys
import warnings

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

import demc_amd as D
from demc_amd import sampler as S


def _binomial_model(rng):
    data = dict(N=10, k=6)
    return D.DEModel(sample_prior=lambda: [rng.uniform()], prior_loglike=D.Priors(θ=D.Beta(1, 1)),
                     loglike=D.BinomialLikelihood(), data=data, names=("θ",))


def oracle_factory(orc, schedule=2):
    def make(**cfg):
        cfg = dict(cfg)
        cfg["schedule"] = schedule
        return orc.Oracle(**{k: v for k, v in cfg.items() if k in orc.CFG_KEYS})
    return make


def test_de_defaults_match_reference():
    de = D.DE(Np=6, bounds=((0, 1),), sample_prior=lambda: [0.5])
    assert (de.n_groups, de.burnin, de.discard_burnin, de.α, de.β, de.ϵ, de.σ, de.κ, de.θsnooker, de.n_initial, de.iter) == \
        (4, 1000, True, 0.1, 0.1, 0.001, 0.05, 1.0, 0.0, 0, 1)  # structs.jl:80-101, :122
    assert de.generate_proposal is D.random_gamma and de.update_particle is D.mh_update
    assert de.evaluate_fitness is D.compute_posterior and de.sample is D.sample_current
    assert de.blocking_on(de) is False
    with warnings.catch_warnings(record=True) as w:  # structs.jl:102-105
        warnings.simplefilter("always")
        de1 = D.DE(Np=6, n_groups=1, bounds=((0, 1),), sample_prior=lambda: [0.5])
        assert de1.α == 0.0 and any("n_groups == 1" in str(x.message) for x in w)


def test_unregistered_hooks_and_closures_are_rejected():
    with pytest.raises(D.DemcError):
        D.DE(Np=6, bounds=((0, 1),), sample_prior=lambda: [0.5], generate_proposal=lambda de, pt, g: pt)
    with pytest.raises(D.DemcError):
        D.DEModel(sample_prior=lambda: [0.5], loglike=lambda data, th: 0.0, names=("θ",), data=None)
    with pytest.raises(D.DemcError):
        D.HIPBackend(schedule="sequential")


def test_nested_theta_layout_names_bounds_blocks():
    """Theta = [mu(3-vector), sigma, B(2x2)] -> D = 8; bounds per top-level parameter; nested block masks
    (structs.jl:45); names as utilities.jl:131-149"""
    th0 = [np.zeros(3), 1.0, np.zeros((2, 2))]
    model = D.DEModel(sample_prior=lambda: th0, prior_loglike=D.Priors(mu=D.Normal(0, 1), sigma=D.TruncatedCauchy(0, 1)),
                      loglike=D.GaussianLikelihood(), data=np.zeros(3), names=("mu", "sigma", "B"))
    blocks = [[np.array([True, False, False]), False, np.array([[True, False], [False, True]])],
              [np.array([False, True, True]), True, np.array([[False, True], [True, False]])]]
    de = D.DE(Np=4, bounds=((-np.inf, np.inf), (0.0, np.inf)), sample_prior=lambda: th0, blocking_on=lambda de: True,
              blocks=blocks)
    lay = S._flat_layout(model, de, th0)
    assert lay["D"] == 8 and lay["sizes"] == [3, 1, 4]
    np.testing.assert_array_equal(lay["lo"], [-np.inf] * 3 + [0.0] + [-np.inf] * 4)  # zip truncates: B unbounded
    np.testing.assert_array_equal(lay["kind"], [1, 1, 1, 2, 0, 0, 0, 0])
    np.testing.assert_array_equal(lay["masks"][0], [1, 0, 0, 0, 1, 0, 0, 1])
    np.testing.assert_array_equal(lay["masks"][0] + lay["masks"][1], np.ones(8))
    assert S.get_names(model, lay["shapes"]) == ["mu[1]", "mu[2]", "mu[3]", "sigma", "B[1,1]", "B[2,1]", "B[1,2]",
                                                 "B[2,2]", "acceptance", "lp"]


def test_blocking_on_is_evaluated_per_iteration(orc):
    """de.blocking_on(de) is called on every iteration with de.iter set (main.jl:34,137): here blocks are used on even
    iterations only; the run must equal the same schedule driven by hand on the engine"""
    rng = np.random.default_rng(1)
    data = rng.normal(0, 1, 40)
    th_init = [list(x) for x in np.stack([rng.normal(0, 1, 24), rng.uniform(0.5, 2, 24)], 1)]
    it = iter(th_init)
    sp = lambda: next(it)
    model = D.DEModel(sample_prior=sp, prior_loglike=D.Priors(μ=D.Normal(0, 10), σ=D.TruncatedCauchy(0, 1)),
                      loglike=D.GaussianLikelihood(), data=data, names=("μ", "σ"))
    de = D.DE(sample_prior=sp, bounds=((-np.inf, np.inf), (0.0, np.inf)), burnin=5, Np=6, discard_burnin=False,
              blocking_on=lambda de: de.iter % 2 == 0, blocks=[[True, False], [False, True]])
    assert S._blocking_schedule(de, 5) == [[1, 1, False], [2, 1, True], [3, 1, False], [4, 1, True], [5, 1, False]]
    it = iter([th_init[0]] + th_init)  # sample() draws one extra prior sample to learn the layout
    ch = D.sample(model, de, D.HIPBackend(seed=4), 12, engine_factory=oracle_factory(orc))
    o = orc.Oracle(n_groups=4, Np=6, D=2, burnin=5, n_rows=12, schedule=2, seed=4)
    o.set_model(0, data, [40])
    o.set_priors([1, 2], [0, 0], [10, 1])
    o.set_bounds([-np.inf, 0], [np.inf, np.inf])
    o.set_state(np.array(th_init))
    for i in range(1, 13):
        o.set_blocks(np.array([[1, 0], [0, 1]], np.uint8) if i % 2 == 0 else np.zeros((0, 2), np.uint8))
        o.step(i, 1)
    th, acc, lp, idh = o.get_history(0, 12)
    exp_th, _, _ = S.rekey_by_id(th, acc, lp, idh)
    np.testing.assert_array_equal(np.transpose(ch.value[:, :2, :], (0, 2, 1)), exp_th)


def test_hierarchical_prior_reference_resolves_to_flat_index():
    th0 = [1.0, 1.0, np.zeros(5)]
    model = D.DEModel(sample_prior=lambda: th0, names=("mu_b0", "sd_b0", "b0"), data=np.zeros(5),
                      prior_loglike=D.Priors(mu_b0=D.Normal(1, 1), sd_b0=D.TruncatedCauchy(0, 1), b0=D.Normal(0, "sd_b0")),
                      loglike=D.HierBinomialLikelihood(50))
    de = D.DE(Np=4, bounds=((-np.inf, np.inf), (0, np.inf), (-np.inf, np.inf)), sample_prior=lambda: th0)
    lay = S._flat_layout(model, de, th0)
    np.testing.assert_array_equal(lay["kind"], [1, 2, 5, 5, 5, 5, 5])
    np.testing.assert_array_equal(lay["ref"][2:], [1] * 5)


def test_rekey_by_id():
    th = np.arange(2 * 3 * 1, dtype=float).reshape(2, 3, 1)
    idh = np.array([[0, 1, 2], [2, 0, 1]])
    o, a, l = S.rekey_by_id(th, np.ones((2, 3), np.uint8), th[..., 0], idh)
    np.testing.assert_array_equal(o[1, :, 0], [4, 5, 3])  # row 1: slot 0 held id 2, slot 1 id 0, slot 2 id 1


def test_discard_burnin_bookkeeping(orc):
    """test/utility_tests.jl:2-40: length(chains) == n_iter without discard, == n_iter - burnin with"""
    rng = np.random.default_rng(29542)
    model = _binomial_model(rng)
    kw = dict(sample_prior=model.sample_prior, Np=4, bounds=((0, 1),), burnin=150)
    ch = D.sample(model, D.DE(discard_burnin=False, **kw), D.HIPBackend(seed=1), 300, engine_factory=oracle_factory(orc))
    assert len(ch) == 300 and ch.value.shape == (300, 3, 16) and ch.names == ["θ", "acceptance", "lp"]
    ch = D.sample(model, D.DE(**kw), D.MCMCThreads(), 300, engine_factory=oracle_factory(orc))
    assert len(ch) == 150
    ch = D.sample(model, D.DE(**kw), 300, engine_factory=oracle_factory(orc))
    assert len(ch) == 150
    acc = ch["acceptance"]
    assert set(np.unique(acc)) <= {0.0, 1.0} and 0.2 < acc.mean() < 0.95
    assert np.all(np.isfinite(ch["lp"]))


def test_binomial_through_the_api(orc):
    """test/binomial_tests.jl end to end through DEModel / DE / sample -> Chains.describe()"""
    from scipy import stats
    rng = np.random.default_rng(5)
    model = _binomial_model(rng)
    de = D.DE(sample_prior=model.sample_prior, bounds=((0, 1),), burnin=1500, Np=4)
    ch = D.sample(model, de, D.HIPBackend(seed=3), 3000, engine_factory=oracle_factory(orc))
    d = ch.describe()["θ"]
    sol = stats.beta(7, 5)
    assert abs(d["mean"] - sol.mean()) < 0.02 * sol.mean() + 0.005
    assert abs(d["std"] - sol.std()) < 0.05 * sol.std()
    assert abs(d["rhat"] - 1.0) < 0.05


def test_optimize_rastrigin(orc):
    """test/optimization_tests.jl:1-44: greedy DE (minimize! + evaluate_fun!) reaches the global minimum"""
    rng = np.random.default_rng(78454111)
    sp = lambda: [rng.uniform(-5, 5, 2)]
    model = D.DEModel(sample_prior=sp, loglike=D.RastriginObjective(), data=None, names=("x",))
    de = D.DE(sample_prior=sp, bounds=((-5.0, 5.0),), Np=6, n_groups=1, update_particle=D.minimize,
              evaluate_fitness=D.evaluate_fun)
    # greedy DE with 6 particles can settle in a local minimum (f = 0.995); like the reference's test this run is
    # pinned by its seeds
    parts = D.optimize(model, de, D.HIPBackend(schedule="synchronous", seed=1), 10000, engine_factory=oracle_factory(orc, 1))
    best, val = D.get_optimal(de, model, parts)
    assert abs(val) < 1e-8 and np.allclose(best["x"], 0, atol=1e-4)


def test_particle_algebra_mirror():
    """test/utility_tests.jl:165-198 on the host Particle class"""
    p1, p2, p3 = D.Particle(Θ=[1.0, 2.0]), D.Particle(Θ=[-2.0, 3.0]), D.Particle(Θ=[-2.0, 3.0])
    np.testing.assert_allclose((p1 + 2).flat(), [3, 4])
    np.testing.assert_allclose((p1 * 4).flat(), [4, 8])
    np.testing.assert_allclose((3 * (p1 - p2)).flat(), [9, -3])
    np.testing.assert_allclose((3 * (p1 - p2) + p3).flat(), [7, 0])
    pr = D.project(D.Particle(Θ=[[-1.0], 4.0]), D.Particle(Θ=[[2.0], 7.0]))
    np.testing.assert_allclose(pr.flat(), [52 / 53, 182 / 53])


def test_bench_spawns_its_own_ranks_when_started_bare():
    """`python bench.py --gpus N` without a launcher starts N rank processes itself (before touching the GPU) and fails
    loudly if any of them fails.  Here (no GPU) both ranks stop at "needs an MI355X"; the parent returns non-zero."""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is visible: the multi-rank launch is the driver's to run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    # (the first rank to fail ends the other: the message appears once or twice; the parent names the rank and its status)
    assert out.stderr.count("bench.py needs an MI355X") >= 1, out.stderr[-1500:]
    assert "2-rank run FAILED" in out.stderr and "exited with status 1" in out.stderr and "---- rank 1 stderr" in out.stderr
    # a launcher that started a different number of ranks than --gpus is an error, not a silent 1-GPU run
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env2, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr


def _id_exchange_script():
    return (
        "import os, sys, hashlib\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "store = bench.control_plane_store(rank, world)\n"
        "uid = bench.exchange_comm_id(store, rank, lambda: bytes(range(128)))\n"
        "assert uid == bytes(range(128)), (rank, uid[:8])\n"
        "store.set('done/%%d' %% rank, b'1')\n"
        "[store.get('done/%%d' %% r) for r in range(world)]   # nobody leaves before everybody has the id (rank 0 may host the store)\n"
        "print('rank', rank, 'of', world, 'got the id', hashlib.sha1(uid).hexdigest()[:8], flush=True)\n" % ROOT)


def test_comm_id_travels_over_the_launchers_store_started_bare(tmp_path):
    """bench.py --gpus N with the library collective: the 128-byte communicator id goes from rank 0 to the others through a
    TCPStore that rank 0 hosts (spawn_ranks' environment) -- exercised here with two CPU processes and a stand-in id"""
    import socket
    import subprocess
    script = tmp_path / "idx.py"
    script.write_text(_id_exchange_script())
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(os.environ, RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1",
                                                                     MASTER_PORT=str(port)), stdout=subprocess.PIPE, text=True)
             for r in range(3)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("got the id" in o for o in outs)


@pytest.mark.timeout(300)
def test_comm_id_travels_over_the_agent_store_under_torch_distributed_run(tmp_path):
    """the same under the driver's launcher (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P`): the agent hosts the store (TORCHELASTIC_USE_AGENT_STORE), every rank connects as a client"""
    import socket
    import subprocess
    script = tmp_path / "idx.py"
    script.write_text(_id_exchange_script())
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(script)], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert out.stdout.count("got the id") == 2


def _bench_module():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_supervisor_ends_the_other_ranks_on_the_first_failure():
    """a rank that dies must not leave its peers blocked in a collective until somebody's timeout.  Fake
    ranks: rank 1 exits 3 at once, ranks 0 and 2 would sleep for minutes -> the parent is back within seconds, non-zero,
    names rank 1 and carries every rank's stderr tail."""
    import time
    bench = _bench_module()
    sleeper = [sys.executable, "-c", "import sys, time; sys.stderr.write('rank waiting in a collective\\n'); sys.stderr.flush(); time.sleep(300)"]
    dier = [sys.executable, "-c", "import sys; sys.stderr.write('ncclCommInitRank: unhandled system error\\n'); sys.exit(3)"]
    t0 = time.monotonic()
    res = bench.supervise([sleeper, dier, sleeper], [dict(os.environ)] * 3, deadline_s=120.0)
    assert time.monotonic() - t0 < 10.0
    assert res["rc"] == 3 and res["failed_rank"] == 1 and "rank 1 exited with status 3" in res["reason"]
    assert "ended rank(s) [0, 2]" in res["reason"]
    assert "unhandled system error" in res["stderr_tails"][1] and "waiting in a collective" in res["stderr_tails"][0]
    assert res["exit_codes"][1] == 3 and all(c != 0 for c in res["exit_codes"])


def test_supervisor_deadline_ends_ranks_that_never_return():
    """an ncclCommInitRank that never returns: the deadline ends every rank, the status is non-zero and says who was left"""
    import time
    bench = _bench_module()
    sleeper = [sys.executable, "-c", "import time; time.sleep(300)"]
    quick = [sys.executable, "-c", "print('{\"metric\": 1}')"]
    t0 = time.monotonic()
    res = bench.supervise([quick, sleeper], [dict(os.environ)] * 2, deadline_s=1.5)
    assert time.monotonic() - t0 < 10.0
    assert res["rc"] == bench.EXIT_STUCK and "deadline" in res["reason"] and "[1]" in res["reason"]
    assert res["exit_codes"][0] == 0 and res["exit_codes"][1] != 0
    # all ranks fine: rank 0's stdout (the JSON line) is what the parent passes on
    res = bench.supervise([quick, quick], [dict(os.environ)] * 2, deadline_s=30.0)
    assert res["rc"] == 0 and res["stdout0"].strip() == '{"metric": 1}'


def test_rank_watchdog_ends_a_rank_stuck_in_a_stage(tmp_path):
    """the in-rank half (what acts under torch.distributed.run, where the parent is not ours): a stage that overruns its
    limit ends the process with EXIT_INIT / EXIT_STUCK and a line that names the stage -- no re-exec"""
    import subprocess
    script = tmp_path / "wd.py"
    script.write_text("import sys, time\nsys.path.insert(0, %r)\nimport bench\nwd = bench.StageWatchdog(5)\n"
                      "wd.enter('comm_init', 0.5, bench.EXIT_INIT)\ntime.sleep(60)\n" % ROOT)
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 17 and "rank 5: stage 'comm_init' exceeded" in out.stderr


def test_bench_rows_are_commands_of_the_same_program():
    """every row of bench.ROWS parses as flags of bench.py itself (tools/collect_profiles.py profiles exactly these commands)
    and is recognised again from its flags (profile_tag), so a hand-run row quotes its own counters"""
    bench = _bench_module()
    names = [n for n, _ in bench.ROWS]
    assert len(set(names)) == len(names)
    for need in ("cfg3_streaming", "cfg3_suffstat", "cfg3_suffstat_post_burnin", "cfg2_streaming", "cfg2_streaming_post_burnin", "cfg4_share",
                 "cfg4_whole", "cfg5_share"):
        assert need in names
    assert bench.profile_tag(bench.parse([])) == "headline"
    for name, over in bench.ROWS:
        a = bench.parse(bench.row_flags(over))
        for k, v in over.items():
            assert getattr(a, k) == v, (name, k)
        assert bench.profile_tag(a) == name
    assert bench.profile_tag(bench.parse(["--config", "cfg2", "--nobs", "1280"])) is None


def test_cdriver_parent_supervises_its_ranks(tmp_path):
    """tools/demc_cdriver.c --ranks N: the first failing rank, or the deadline, ends the others; the fake ranks of
    the test hook touch no GPU"""
    import subprocess
    import time
    lib = os.path.join(ROOT, "differentialevolutionmcmc.jl_amd")
    if not os.path.exists(os.path.join(lib, "libdemc_hip.so")):
        pytest.skip("libdemc_hip.so not built")
    exe = str(tmp_path / "demc_cdriver")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "demc_cdriver.c"), "-o", exe,
                           "-L", lib, "-ldemc_hip", "-lm", "-Wl,-rpath," + lib, "-Wl,-rpath-link,/opt/rocm/lib"])
    env = dict(os.environ, LD_LIBRARY_PATH=lib + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    t0 = time.monotonic()
    out = subprocess.run([exe, "--ranks", "3"], env=dict(env, DEMC_CDRIVER_FAKE="1:3"), capture_output=True, text=True, timeout=60)
    assert out.returncode == 3 and "rank 1 failed with status 3" in out.stderr and time.monotonic() - t0 < 10
    out = subprocess.run([exe, "--ranks", "2", "--deadline", "1"], env=dict(env, DEMC_CDRIVER_FAKE="0:sleep"), capture_output=True,
                         text=True, timeout=60)
    assert out.returncode == 8 and "deadline passed" in out.stderr


def test_committed_counter_summaries_belong_to_the_sources_that_ship():
    """bench.py quotes `roofline.traffic` only from summaries collected on the kernel sources of the tree (source_sha16); at the end
    of a round the two must agree, or the driver's line silently carries no counter traffic.  (A round that has not collected
    yet -- profiles/<PROFILE_ROUND>/ empty -- is not judged here.)"""
    import glob
    import json
    bench = _bench_module()
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", bench.PROFILE_ROUND, "*.json")) if not f.endswith("_line.json")]
    stamped = [f for f in files if "source_sha16" in json.load(open(f))]
    if not stamped:
        pytest.skip("no summaries collected for this round yet")
    fp = bench.source_fingerprint()
    stale = [os.path.basename(f) for f in stamped if json.load(open(f))["source_sha16"] != fp]
    assert not stale, f"collected on other kernel sources (re-run tools/collect_profiles.py): {stale[:5]}"
    # every row of the driver's command has its summaries and the line of the un-profiled run
    for name in ["headline"] + [n for n, _ in bench.ROWS]:
        for suffix in ("pmc.json", "kernel_stats.csv", "line.json"):
            assert os.path.exists(os.path.join(ROOT, "profiles", bench.PROFILE_ROUND, f"bench_{name}_{suffix}")), (name, suffix)


def _fake_bench_result(bench, n_gpus=1, n_rows=None):
    """a result object shaped like main()'s `out`, with prose as long as the real one's (round 4's 23 KB line) and one row per
    entry of bench.ROWS"""
    prose = "x" * 600
    rf = dict(bound="mfma", kernel="k_cross_mfma<8,4> (v_mfma_f64_16x16x4_f64)" + prose, achieved=75.85414595078859, peak=78.6,
              unit="TFLOP/s", frac=0.9650654701118142, flop_counted=prose, what_is_streamed=prose, launch_ms=2.7647163826227192,
              launches=400, traffic=102952178.93877551, traffic_source=prose, wasted_traffic_ratio=3.0290201628373694, timing=prose,
              per_kernel_ms_per_iter=dict(propose=0.0431, loglike=5.5294), device_ms_per_iter=5.596939020240679)
    rows = [dict(name=name, workload=prose, value=4973723.868929676, unit="particle-updates/s", ms_per_step=13.176445200224407,
                 steps=20, warmup=5, kernels=prose, accept_rate=0.408, finite_weights=True, roofline=dict(rf), seconds=0.91)
            for name, _ in bench.ROWS][:n_rows]
    for r in rows:  # the fields only some rows carry: STREAMING rows both fractions, HBM rows the counters' fraction, two rows a CPU leg
        if "streaming" in r["name"]:
            r["roofline"].update(frac_executed=0.9650654701118142, frac_survey=1.4175654701118142)
        if "suffstat" in r["name"]:
            r["roofline"]["counter_frac"] = 0.21512345678
        if r["name"].startswith("cfg4_whole"):
            r["roofline"]["shape_frac"] = 0.63123456789
        if r["name"].startswith("cfg5") or r["name"] == "cfg2_direct":
            r["roofline"]["shader_clock_mhz"] = 2251.123456
        if r["name"] in bench.CPU_ROWS:
            r["cpu_baseline"] = dict(value=23877.65572005412, value_single_thread=1737.3869838033977, cores=16, sample=prose)
    rows.append(dict(name="a_row_that_failed", error="RuntimeError: " + prose))
    sharded = [dict(name=name, workload=prose, value=7822123.456789, unit="particle-updates/s", scaling="strong", ms_per_step=0.1309123456,
                    steps=40, warmup=10, n_gpus=n_gpus, groups_total=spec["total_groups"], groups_per_rank=spec["total_groups"] // n_gpus,
                    particles_per_gpu=512, block_sweeps_per_step=2, all_gathers_per_rank=[5] * n_gpus, rccl_nranks=n_gpus,
                    collective="library", collective_fallback=None, kernels=prose, finite_weights=True, seconds=3.2)
               for name, spec in bench.SHARDED_ROWS]
    rf = dict(rf, shader_clock_mhz=2051.123456, frac_at_clock=0.7012345678, frac_of_mix_ceiling_at_clock=0.93512345678,
              sclk_sysfs_mhz=dict(median=2100.0, min=2100.0, max=2400.0, samples=140, source=prose))
    return {"metric": "particle-updates/sec (proposal+loglike+accept) at D=32, N=1e5", "value": 11631628.466947727,
            "unit": "particle-updates/s", "n_gpus": n_gpus, "steps": 200, "warmup": 50, "ms_per_step": 5.634292754984926,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": prose, "particles_per_gpu": 65536, "block_sweeps_per_step": 1, "parallelism": prose,
                       "collective": None if n_gpus == 1 else prose, "collective_fallback": None,
                       "rccl_nranks": None if n_gpus == 1 else n_gpus, "all_gathers_rank0": 21, "all_gathers_per_rank": [21] * n_gpus,
                       "ms_per_step_min_over_ranks": 5.61, "ms_per_step_max_over_ranks": 5.634292754984926},
            "particle_parameter_updates_per_s": 372212110.94232726, "particle_iterations_per_s": 11631628.466947727,
            "accuracy": dict(timed_chain=dict(accept_rate=0.2288818359375, note=prose), posterior_mean_l1_rel=9.978460843456112e-06,
                             max_abs_err_in_posterior_sd=0.00976138608426626, leg=prose),
            "roofline": rf, "headline_context": dict(streaming_value=11631628.466947727, streaming_frac_executed=0.9650654701118142,
                                                     streaming_frac_survey=1.4175654701118142, suffstat_value=3736123456.789,
                                                     cpu_baseline_like_for_like_ratio=208.3003426819826, note=prose),
            "cpu_baseline": dict(value=23877.65572005412, unit="particle-updates/s", cores=16, kind="port",
                                 value_single_thread=1737.3869838033977, cpu_model="AMD EPYC 9575F 64-Core Processor", sample=prose),
            "rows": rows if n_gpus == 1 else sharded}


def test_bench_final_line_fits_what_a_harness_keeps(capsys):
    """round 4's record was lost to a 23 KB line: the LAST stdout line is one JSON object under bench.LINE_LIMIT bytes that
    carries the contract's fields, roofline, cpu_baseline, headline_context and every row as numbers; the prose goes to an earlier
    {"detail": ...} line and to gpurun_out/bench_detail.json"""
    import json
    bench = _bench_module()
    assert bench.LINE_LIMIT <= 6000
    for n_gpus in (1, 8):
        out = _fake_bench_result(bench, n_gpus)
        text = bench.compact_line(out)
        assert len(text) < bench.LINE_LIMIT and "\n" not in text, len(text)
        line = json.loads(text)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in line, k
        assert line["config"]["workload"] and line["n_gpus"] == n_gpus
        for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "launch_ms", "traffic", "wasted_traffic_ratio"):
            assert k in line["roofline"], k
        assert abs(line["roofline"]["frac"] - out["roofline"]["achieved"] / out["roofline"]["peak"]) < 1e-4
        assert abs(line["value"] / out["value"] - 1) < 1e-4  # 5 significant digits
        for k in ("value", "unit", "cores", "kind", "value_single_thread", "cpu_model", "sample"):
            assert k in line["cpu_baseline"], k
        assert set(line["headline_context"]) == {"streaming_value", "streaming_frac_executed", "streaming_frac_survey", "suffstat_value",
                                                 "cpu_baseline_like_for_like_ratio"}
        assert line["roofline"]["shader_clock_mhz"] == 2051.1 and line["roofline"]["sclk_sysfs_mhz"] == 2100.0
        assert line["config"]["all_gathers_per_rank"] == [21] * n_gpus
        assert line["config"]["ms_per_step_min_over_ranks"] == 5.61 and line["config"]["ms_per_step_max_over_ranks"] > 5.63
        if n_gpus == 1:
            assert [r["name"] for r in line["rows"]] == [n for n, _ in bench.ROWS] + ["a_row_that_failed"]
            for r in line["rows"][:-1]:
                assert {"name", "value", "ms_per_step", "steps", "frac", "launch_ms", "bound", "traffic"} <= set(r), r
                assert set(r) <= {"name", "value", "ms_per_step", "steps", "frac", "launch_ms", "bound", "traffic", "counter_frac", "shape_frac",
                                  "frac_executed", "frac_survey", "cpu", "cpu_1thread", "clock_mhz"}, r
                assert ("frac_survey" in r) == ("streaming" in r["name"]) and ("cpu" in r) == (r["name"] in bench.CPU_ROWS)
                assert all(not isinstance(v, str) or len(v) < 64 for v in r.values())
            assert len(line["rows"][-1]["error"]) <= 80
            assert "rows_truncated" not in line
        else:  # the N > 1 line: BASELINE's 8-GPU configs as sharded rows (strong partition), numbers only
            assert line["config"]["rccl_nranks"] == 8
            assert [r["name"] for r in line["rows"]] == [n for n, _ in bench.SHARDED_ROWS] == ["cfg4_sharded", "cfg5_sharded"]
            for r, (_, spec) in zip(line["rows"], bench.SHARDED_ROWS):
                assert set(r) == {"name", "value", "ms_per_step", "steps", "n_gpus", "groups_per_rank", "all_gathers_per_rank", "rccl_nranks"}
                assert r["groups_per_rank"] * 8 == spec["total_groups"] and r["all_gathers_per_rank"] == [5] * 8 and r["n_gpus"] == 8
    # what main() prints: the detail line first, the compact line last
    bench.DETAIL_FILE = os.path.join("gpurun_out", "bench_detail_test.json")
    try:
        bench.emit(_fake_bench_result(bench))
        lines = capsys.readouterr().out.strip().split("\n")
        assert len(lines) == 2 and "detail" in json.loads(lines[0]) and len(lines[0]) > 6000
        assert len(lines[1]) < bench.LINE_LIMIT and json.loads(lines[1])["detail"] == bench.DETAIL_FILE
        assert json.load(open(os.path.join(ROOT, bench.DETAIL_FILE)))["rows"][0]["workload"]
    finally:
        try:
            os.remove(os.path.join(ROOT, bench.DETAIL_FILE))
        except OSError:
            pass


def test_bench_rank_files_merge_into_one_short_line():
    """every rank writes gpurun_out/rank<r>.json before the first collective after the timed region; a run that dies later
    still yields one line (marked partial) with the whole-job value, rccl_nranks, the gathers of every rank and min/max ms"""
    import json
    bench = _bench_module()
    recs = [dict(rank=r, world=8, steps=200, warmup=50, seconds_timed=1.12 + 0.01 * r, ms_per_step=(1.12 + 0.01 * r) / 200 * 1e3,
                 particles=65536, block_sweeps_per_step=1, all_gathers=21, rccl_nranks=8, collective="library", collective_fallback=None,
                 workload="cfg3: " + "y" * 500, config="cfg3") for r in (3, 0, 1, 2, 7, 6, 5, 4)]
    text = bench.merge_rank_files(recs, "run failed after the timed region (rank 2 exited with status 18)")
    assert len(text) < bench.LINE_LIMIT
    line = json.loads(text)
    assert line["n_gpus"] == 8 and line["partial"] and line["config"]["rccl_nranks"] == 8
    assert line["config"]["all_gathers_per_rank"] == [21] * 8
    assert abs(line["value"] / (8 * 65536 * 200 / 1.19) - 1) < 1e-4  # whole job / MAX over ranks
    assert line["config"]["ms_per_step_min_over_ranks"] == 5.6 and line["config"]["ms_per_step_max_over_ranks"] == 5.95
    assert line["rows"] is None
    assert bench.merge_rank_files(recs[:7], "x") is None  # a rank without a file: no line is made up
    # ranks that also got through the sharded rows (BASELINE's cfg4 / cfg5 partitioned over the ranks) carry them in their files
    for r in recs:
        r["rows"] = [dict(name="cfg4_sharded", config="cfg4", steps=40, warmup=10, seconds_timed=0.0052 + 1e-4 * r["rank"], particles=512,
                          block_sweeps_per_step=2, all_gathers=4, groups=16, rccl_nranks=8)] + \
                    ([dict(name="cfg5_sharded", config="cfg5", steps=40, warmup=10, seconds_timed=0.1, particles=8192, block_sweeps_per_step=1,
                           all_gathers=3, groups=64, rccl_nranks=8)] if r["rank"] != 5 else [])  # (rank 5 died inside the second row)
    text = bench.merge_rank_files(recs, "late failure")
    line = json.loads(text)
    assert len(text) < bench.LINE_LIMIT and [r["name"] for r in line["rows"]] == ["cfg4_sharded"]
    assert abs(line["rows"][0]["value"] / (8 * 512 * 2 * 40 / 0.0059) - 1) < 1e-4 and line["rows"][0]["all_gathers_per_rank"] == [4] * 8


def test_bench_reads_the_traffic_shape_roof_of_the_long_row_kernel():
    """cfg4's rows are also quoted against what a bare kernel of the long-row kernel's traffic shape reaches
    (tools/longrow_traffic_probe.hip; the committed output of its run on the box): the average of a subject-sweep and a
    hyper-parameter-sweep launch at the whole cfg4's size, tens of microseconds -- parsed from the file, not typed into bench.py"""
    bench = _bench_module()
    path = os.path.join(ROOT, "profiles", bench.PROFILE_ROUND, "longrow_traffic_probe.txt")
    if not os.path.exists(path):
        pytest.skip("no probe output committed for this round")
    roof = bench.longrow_shape_roof_ms()
    assert roof is not None and 0.03 < roof < 0.2, roof
    txt = open(path).read()
    assert "hipMemcpy D2D" in txt and "subject sweep" in txt and "hyper sweep" in txt
