"""GPU tests through the public API (DEModel / DE / sample) and at BASELINE sizes.

Statistical gates mirror the reference's own tests with closed-form or grid targets (no Turing/NUTS here);
full-size tests use size-independent properties (weights re-evaluate to themselves, ids stay a permutation,
suffstat == streaming, same seed => same bits, two shards == one)."""
import numpy as np
import pytest
from scipy import stats

import demc_amd as D
from conftest import make_problem, setup_engine
from demc_amd import families as F

pytestmark = pytest.mark.gpu


def test_binomial_tests_jl():
    """test/binomial_tests.jl: Beta(1,1) prior, k ~ Binomial(10, theta): posterior Beta(k+1, N-k+1), rtol 0.02"""
    rng = np.random.default_rng(29542)
    N, k = 10, int(rng.binomial(10, 0.5))
    sp = lambda: [rng.uniform()]
    model = D.DEModel(sample_prior=sp, prior_loglike=D.Priors(θ=D.Beta(1, 1)), loglike=D.BinomialLikelihood(),
                      data=dict(N=N, k=k), names=("θ",))
    de = D.DE(sample_prior=sp, bounds=((0, 1),), burnin=1500, Np=4)
    ch = D.sample(model, de, D.HIPBackend(seed=11), 6000)
    d = ch.describe()["θ"]
    sol = stats.beta(k + 1, N - k + 1)
    assert abs(d["mean"] - sol.mean()) < 0.02 * sol.mean()
    assert abs(d["std"] - sol.std()) < 0.03 * sol.std()
    assert abs(d["rhat"] - 1.0) < 0.02
    assert len(ch) == 4500 and 0.2 < ch["acceptance"].mean() < 0.95


@pytest.mark.parametrize("schedule", ["two_colour", "synchronous"])
def test_gaussian_tests_jl(schedule):
    """test/gaussian_tests.jl: mu ~ N(0,10), sigma ~ Cauchy+(0,1), 50 obs, DE(burnin=1500, Np=6).  Target: the
    numerically integrated posterior (the north-star gate "posterior means within 1 %")."""
    rng = np.random.default_rng(973536)
    data = rng.normal(0, 1, 50)
    sp = lambda: [rng.normal(0, 10), abs(rng.standard_cauchy())]
    model = D.DEModel(sample_prior=sp, prior_loglike=D.Priors(μ=D.Normal(0, 10), σ=D.TruncatedCauchy(0, 1)),
                      loglike=D.GaussianLikelihood(), data=data, names=("μ", "σ"))
    de = D.DE(sample_prior=sp, bounds=((-np.inf, np.inf), (0.0, np.inf)), burnin=1500, Np=6)
    ch = D.sample(model, de, D.HIPBackend(schedule=schedule, seed=5), 21500)
    d = ch.describe()
    mu, sg = np.linspace(-1.2, 1.2, 481), np.linspace(0.5, 2.2, 481)
    M, S = np.meshgrid(mu, sg, indexing="ij")
    lp = (-0.5 * ((data[None, None] - M[..., None]) / S[..., None]) ** 2).sum(-1) - 50 * np.log(S) - 0.5 * (M / 10) ** 2 - np.log1p(S ** 2)
    p = np.exp(lp - lp.max())
    p /= p.sum()
    em, es = (p * M).sum(), (p * S).sum()
    sm, ss = np.sqrt((p * M ** 2).sum() - em ** 2), np.sqrt((p * S ** 2).sum() - es ** 2)
    assert abs(d["μ"]["mean"] - em) < 0.01 and abs(d["σ"]["mean"] - es) < 0.01 * es + 0.005
    assert abs(d["μ"]["std"] - sm) < 0.01 and abs(d["σ"]["std"] - ss) < 0.01
    assert abs(d["μ"]["rhat"] - 1) < 0.05 and abs(d["σ"]["rhat"] - 1) < 0.05


def test_multivariate_normal_tests_jl():
    """test/multivariate_normal_tests.jl (10 means instead of 30, 20k iterations instead of 50k): sample = resample,
    theta_snooker = 0.1, Np = 3, n_groups = 1, nested Theta [mu, sigma]; sds ~ 0.1, cor(data means, post means) > 0.98"""
    rng = np.random.default_rng(505514)
    n_mu, n_d = 10, 100
    data = rng.normal(0, 1, (n_mu, n_d))  # Julia layout: variables x observations
    sp = lambda: [rng.normal(0, 1, n_mu), abs(rng.standard_cauchy())]
    model = D.DEModel(sample_prior=sp, prior_loglike=D.Priors(μ=D.Normal(0, 1), σ=D.TruncatedCauchy(0, 1)),
                      loglike=D.MvNormalIsoLikelihood(), data=data, names=("μ", "σ"))
    with pytest.warns(UserWarning):
        de = D.DE(sample_prior=sp, bounds=((-np.inf, np.inf), (0.0, np.inf)), sample=D.resample, burnin=5000,
                  n_initial=(n_mu + 1) * 4, Np=3, n_groups=1, θsnooker=0.1)
    ch = D.sample(model, de, D.MCMCThreads(), 20000)
    d = ch.describe()
    sds = np.array([d[f"μ[{i + 1}]"]["std"] for i in range(n_mu)])
    means = np.array([d[f"μ[{i + 1}]"]["mean"] for i in range(n_mu)])
    np.testing.assert_allclose(sds, 0.1, atol=0.012)
    assert np.all(np.abs(means) < 0.3)
    assert np.corrcoef(data.mean(1), means)[0, 1] > 0.98


def test_blocking_tests_jl():
    """test/blocking_tests.jl: blocks [[true,false],[false,true]], blocking on every iteration; means ~ (0, 1)"""
    rng = np.random.default_rng(58122)
    data = rng.normal(0, 1, 1000)
    sp = lambda: [rng.normal(0, 10), abs(rng.standard_cauchy())]
    model = D.DEModel(sample_prior=sp, prior_loglike=D.Priors(μ=D.Normal(0, 10), σ=D.TruncatedCauchy(0, 1)),
                      loglike=D.GaussianLikelihood(), data=data, names=("μ", "σ"))
    de = D.DE(sample_prior=sp, bounds=((-np.inf, np.inf), (0.0, np.inf)), burnin=1000, Np=6,
              blocking_on=lambda de: True, blocks=[[True, False], [False, True]])
    ch = D.sample(model, de, D.HIPBackend(seed=2), 2000)
    d = ch.describe()
    assert abs(d["μ"]["mean"] - 0.0) < 0.1 and abs(d["σ"]["mean"] - 1.0) < 0.1
    assert abs(d["μ"]["rhat"] - 1) < 0.05


def test_optimization_tests_jl():
    """test/optimization_tests.jl: rastrigin minimum (minimize! + evaluate_fun!) and Gaussian MLE (maximize!)"""
    rng = np.random.default_rng(78454111)
    sp = lambda: [rng.uniform(-5, 5, 2)]
    model = D.DEModel(sample_prior=sp, loglike=D.RastriginObjective(), data=None, names=("x",))
    with pytest.warns(UserWarning):
        de = D.DE(sample_prior=sp, bounds=((-5.0, 5.0),), Np=6, n_groups=1, update_particle=D.minimize,
                  evaluate_fitness=D.evaluate_fun)
    vals = []
    for seed in (1, 4, 6):
        parts = D.optimize(model, de, D.HIPBackend(schedule="synchronous", seed=seed), 10000)
        vals.append(D.get_optimal(de, model, parts)[1])
    assert min(vals) < 1e-8
    data = rng.normal(0, 1, 100)
    sp2 = lambda: [rng.normal(0, 1), rng.uniform(0.5, 2)]
    model = D.DEModel(sample_prior=sp2, loglike=D.GaussianLikelihood(), data=data, names=("μ", "σ"))
    de = D.DE(sample_prior=sp2, bounds=((-np.inf, np.inf), (0.0, np.inf)), Np=6, n_groups=4, update_particle=D.maximize,
              evaluate_fitness=D.evaluate_fun)
    parts = D.optimize(model, de, D.MCMCThreads(), 5000)
    best, _ = D.get_optimal(de, model, parts)
    assert abs(best["μ"] - data.mean()) < 1e-4 and abs(best["σ"] - data.std()) < 1e-4


# ------------------------------------------------------------------------------------------------ full size


def _cfg3(N=100000, d=32, G=256, Np=256, **kw):
    rng = np.random.default_rng(20260002)
    A = rng.normal(0, 1, (d, d))
    Sigma = A @ A.T / d + 0.5 * np.eye(d)
    X = rng.normal(0, 1, d) + rng.normal(0, 1, (N, d)) @ np.linalg.cholesky(Sigma).T
    cfg = dict(n_groups=G, Np=Np, D=d, n_rows=6, schedule=2, seed=7)
    cfg.update(kw)
    eng = D.HipEngine(**cfg)
    eng.set_model(F.FAM_MVN_FULL, X, [N, d], Sigma)
    eng.set_priors([F.PRIOR_NORMAL] * d, [0.0] * d, [1.0] * d)
    eng.set_bounds([-np.inf] * d, [np.inf] * d)
    return eng, X, Sigma


def test_cfg2_posterior_matches_closed_form():
    """BASELINE cfg2 (MvNormal D=8, 32 x 64 particles, N=1e4): after burn-in the chain reproduces the conjugate
    Gaussian posterior -- means within 1 % (L1, the metric's accuracy half) and the right spread"""
    d, N, G, Np = 8, 10000, 32, 64
    rng = np.random.default_rng(20260001)
    A = rng.normal(0, 1, (d, d))
    Sigma = A @ A.T / d + 0.5 * np.eye(d)
    X = rng.normal(0, 1, d) + rng.normal(0, 1, (N, d)) @ np.linalg.cholesky(Sigma).T
    burn, n_it = 300, 700
    for schedule in (2, 1):
        eng = D.HipEngine(n_groups=G, Np=Np, D=d, n_rows=n_it, schedule=schedule, seed=3, burnin=burn, trace=0)
        eng.set_model(F.FAM_MVN_FULL, X, [N, d], Sigma)
        eng.set_priors([F.PRIOR_NORMAL] * d, [0.0] * d, [1.0] * d)
        eng.set_bounds([-np.inf] * d, [np.inf] * d)
        eng.set_state(rng.normal(0, 1, (G * Np, d)))
        eng.step(1, n_it)
        th = eng.get_history(burn + 100, n_it)[0].reshape(-1, d)
        eng.close()
        Ainv = np.linalg.inv(Sigma)
        cov = np.linalg.inv(N * Ainv + np.eye(d))
        mean = cov @ (N * Ainv @ X.mean(0))
        assert np.abs(th.mean(0) - mean).sum() / np.abs(mean).sum() < 0.01
        np.testing.assert_allclose(th.std(0), np.sqrt(np.diag(cov)), rtol=0.1 if schedule == 2 else 0.2)
        if schedule == 2:  # the valid schedule also reproduces the correlations
            np.testing.assert_allclose(np.corrcoef(th.T), cov / np.sqrt(np.outer(np.diag(cov), np.diag(cov))), atol=0.05)


def test_cfg3_shape_posterior_matches_closed_form():
    """cfg3's model (D=32 full Sigma, N=1e5 observations) with fewer particles (64 x 64), started 2x over-dispersed and
    run without the burn-in heuristic: the streaming MFMA path must converge to the conjugate posterior (mean L1 < 1 %,
    marginal sds within 5 %).  generate_proposal = variable_gamma (gamma = 2.38/sqrt(2 d), crossover.jl:213-226): with
    the default random_gamma (gamma in [0.5, 1]) a 32-dimensional target accepts < 1 % of the proposals after
    burn-in -- a property of the sampler (the reference's too), not of the kernels."""
    d, N, G, Np = 32, 100000, 64, 64
    rng = np.random.default_rng(20260002)
    A = rng.normal(0, 1, (d, d))
    Sigma = A @ A.T / d + 0.5 * np.eye(d)
    X = rng.normal(0, 1, d) + rng.normal(0, 1, (N, d)) @ np.linalg.cholesky(Sigma).T
    Ainv = np.linalg.inv(Sigma)
    cov = np.linalg.inv(N * Ainv + np.eye(d))
    mean = cov @ (N * Ainv @ X.mean(0))
    n_it = 1200
    eng = D.HipEngine(n_groups=G, Np=Np, D=d, n_rows=n_it, schedule=2, seed=11, burnin=0, trace=0, proposal_kind=2)
    eng.set_model(F.FAM_MVN_FULL, X, [N, d], Sigma)
    eng.set_priors([F.PRIOR_NORMAL] * d, [0.0] * d, [1.0] * d)
    eng.set_bounds([-np.inf] * d, [np.inf] * d)
    eng.set_state(mean + 2.0 * rng.normal(0, 1, (G * Np, d)) @ np.linalg.cholesky(cov).T)
    eng.step(1, n_it)
    th, acc, _, _ = eng.get_history(n_it - 300, n_it)
    eng.close()
    th = th.reshape(-1, d)
    assert np.abs(th.mean(0) - mean).sum() / np.abs(mean).sum() < 0.01
    np.testing.assert_allclose(th.std(0), np.sqrt(np.diag(cov)), rtol=0.05)
    np.testing.assert_allclose(np.corrcoef(th.T), cov / np.sqrt(np.outer(np.diag(cov), np.diag(cov))), atol=0.05)
    assert 0.1 < acc.mean() < 0.4


def test_cfg3_full_size_properties(orc):
    """BASELINE cfg3 (D=32, N=1e5, 256 x 256 particles): properties that need no full-size oracle run"""
    d, P = 32, 65536
    eng, X, Sigma = _cfg3()
    th0 = np.random.default_rng(1).normal(0, 1, (P, d))
    eng.set_state(th0)
    _, w0, _ = eng.get_state()
    # (1) streaming weights against the closed form with sufficient statistics, computed independently in numpy
    Ainv = np.linalg.inv(Sigma)
    xbar, N = X.mean(0), X.shape[0]
    Sc = (X - xbar).T @ (X - xbar)
    _, logdet = np.linalg.slogdet(Sigma)
    sub = np.arange(0, P, 257)
    dm = th0[sub] - xbar
    ll = -0.5 * N * (d * np.log(2 * np.pi) + logdet) - 0.5 * (N * np.einsum("pi,ij,pj->p", dm, Ainv, dm) + np.trace(Ainv @ Sc))
    pr = stats.norm(0, 1).logpdf(th0[sub]).sum(1)
    np.testing.assert_allclose(w0[sub], ll + pr, rtol=1e-9)
    # (2) a sample of rows against the CPU oracle (whitened O(N d) form)
    o = orc.Oracle(n_groups=1, Np=4, D=d, n_rows=0, store_history=0)
    o.set_model(F.FAM_MVN_FULL, X, [N, d], Sigma)
    o.set_priors([F.PRIOR_NORMAL] * d, [0.0] * d, [1.0] * d)
    np.testing.assert_allclose(w0[sub[:32]], o.logpost(th0[sub[:32]]), rtol=1e-9)
    # (3) after stepping: stored weights re-evaluate to themselves; ids stay a permutation; history row == state
    eng.step(1, 5)
    th, w, ids = eng.get_state()
    np.testing.assert_allclose(eng.logpost(th[sub]), w[sub], rtol=1e-10)
    assert np.array_equal(np.sort(ids), np.arange(P))
    hth, hacc, hlp, hid = eng.get_history(4, 5)
    assert np.array_equal(hth[0], th) and np.array_equal(hid[0], ids) and np.array_equal(hlp[0], w)
    assert 0.01 < hacc.mean() < 0.99
    eng.close()
    # (4) sufficient-statistic mode gives the same posterior values
    eng2, _, _ = _cfg3(loglike_mode=1)
    eng2.set_state(th0)
    np.testing.assert_allclose(eng2.get_state()[1], w0, rtol=1e-9)
    eng2.close()


def test_same_seed_same_bits_and_two_shards_equal_one():
    """determinism and SURVEY 8e on one GPU: two handles each owning half of the groups,
    exchanging through demc_migration_pack/apply, reproduce the single-handle run bit for bit"""
    import torch
    prob = make_problem("mvn_full", np.random.default_rng(3), N=2000, d=16)
    G, Np, d, n_it = 16, 32, 16, 30
    th0 = prob["init"](G * Np)

    def mk(ng, off):
        e = D.HipEngine(n_groups=ng, Np=Np, D=d, n_rows=n_it, schedule=2, seed=42, alpha=0.5, theta_snooker=0.1,
                        group_offset=off, n_groups_total=G, burnin=10)
        setup_engine(e, prob)
        return e
    runs = []
    for _ in range(2):
        e = mk(G, 0)
        e.set_state(th0)
        e.step(1, n_it)
        runs.append(e.get_history(0, n_it))
        e.close()
    for a, b in zip(*runs):
        assert np.array_equal(a, b)
    a, b = mk(G // 2, 0), mk(G // 2, G // 2)
    a.set_state(th0[: G // 2 * Np])
    b.set_state(th0[G // 2 * Np:])
    rows = torch.zeros((G, d + 3), dtype=torch.float64, device="cuda")
    half = G // 2 * (d + 3) * 8
    n_mig = 0
    for it in range(1, n_it + 1):
        if a.migration_due(it):
            a.migration_pack_dev(it, rows.data_ptr())
            b.migration_pack_dev(it, rows.data_ptr() + half)
            torch.cuda.synchronize()
            a.migration_apply_dev(it, rows.data_ptr())
            b.migration_apply_dev(it, rows.data_ptr())
            n_mig += 1
        a.update(it, 1)
        b.update(it, 1)
    assert n_mig >= 5
    ha, hb = a.get_history(0, n_it), b.get_history(0, n_it)
    for one, x, y in zip(runs[0], ha, hb):
        assert np.array_equal(one, np.concatenate([x, y], axis=1))
    a.close()
    b.close()


def test_sharded_driver_on_gpu_world1():
    """ShardedDriver on a CUDA device without a process group: pack -> (no gather) -> apply on device pointers"""
    import torch
    from demc_amd.distributed import ShardedDriver
    prob = make_problem("gaussian", np.random.default_rng(5))
    th0 = prob["init"](48)
    hs = []
    for use_driver in (False, True):
        e = D.HipEngine(n_groups=8, Np=6, D=2, n_rows=40, schedule=2, seed=9, alpha=0.4)
        setup_engine(e, prob)
        e.set_state(th0)
        if use_driver:
            e.set_stream(torch.cuda.current_stream().cuda_stream)
            drv = ShardedDriver(e, None, torch.device("cuda", 0))
            drv.step(1, 40)
            assert drv.n_exchanges >= 8
        else:
            e.step(1, 40)
        hs.append(e.get_history(0, 40))
        e.close()
    for a, b in zip(*hs):
        assert np.array_equal(a, b)


GAUSS_SRC = """
__device__ double demc_user_obs(const double* th, int D, const double* x, long long N, long long i,
                                const double* hyper, int nhyper) {
    const double z = (x[i] - th[0]) / th[1];
    return -0.5 * (z * z + 1.8378770664093453) - log(th[1]);
}
"""


def test_user_source_plugin_matches_registered_family():
    """demc_set_model_source: the Gaussian term of Examples/Gaussian_Example.jl:26-28 written as a HIP device function
    gives the same log-posteriors as the registered Gaussian family"""
    prob = make_problem("gaussian", np.random.default_rng(41), N=500)
    th = prob["init"](32)
    a = D.HipEngine(n_groups=4, Np=8, D=2, schedule=1)
    setup_engine(a, prob)
    b = D.HipEngine(n_groups=4, Np=8, D=2, schedule=1)
    b.set_model_source(GAUSS_SRC, prob["data"], [500])
    b.set_priors(prob["pk"], prob["pa"], prob["pb"], prob["pref"])
    b.set_bounds(prob["lo"], prob["hi"])
    np.testing.assert_allclose(b.logpost(th), a.logpost(th), rtol=1e-12)
    a.close()
    with pytest.raises(D.DemcError) as e:
        b.set_model_source("__device__ double demc_user_obs(int oops) { return undeclared; }", prob["data"], [500])
    assert "does not compile" in str(e.value) and "undeclared" in str(e.value)
    b.close()


def test_user_source_model_end_to_end():
    """a model the registry does not have: x_i ~ Exponential(rate), rate ~ Uniform(0, 50).  Posterior = Gamma(N+1, sum x)
    truncated far in the tail -> mean (N+1)/sum(x), sd sqrt(N+1)/sum(x); through DEModel / DE / sample"""
    rng = np.random.default_rng(42)
    x = rng.exponential(1 / 3.0, 200)
    src = """
    __device__ double demc_user_obs(const double* th, int D, const double* x, long long N, long long i,
                                    const double* hyper, int nhyper) { return log(th[0]) - th[0] * x[i]; }
    """
    sp = lambda: [rng.uniform(0.5, 10)]
    model = D.DEModel(sample_prior=sp, prior_loglike=D.Priors(rate=D.Uniform(0, 50)), loglike=D.SourceLikelihood(src),
                      data=x, names=("rate",))
    de = D.DE(sample_prior=sp, bounds=((0.0, 50.0),), burnin=1000, Np=6)
    ch = D.sample(model, de, D.HIPBackend(seed=8), 6000)
    d = ch.describe()["rate"]
    assert abs(d["mean"] - 201 / x.sum()) < 0.02 * 201 / x.sum()
    assert abs(d["std"] - np.sqrt(201) / x.sum()) < 0.05 * np.sqrt(201) / x.sum()


def test_plain_c_caller_of_the_abi(tmp_path):
    """tools/demc_cdriver.c: a host with no Python in it (what a Julia @ccall shim or any FFI does) runs
    Examples/Gaussian_Example.jl through include/demc.h"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "differentialevolutionmcmc.jl_amd")
    exe = str(tmp_path / "demc_cdriver")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "tools", "demc_cdriver.c"), "-o", exe,
                           "-L", libdir, "-ldemc_hip", "-lm", "-Wl,-rpath-link,/opt/rocm/lib"])
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "C-ABI driver OK" in out.stdout
    # second scenario: the call sequence of julia/DEMCHIP.jl's sample() for nested Theta with per-iteration blocking_on
    # (set_blocks / demc_step per run, layout-0 export checked element by element against the raw history)
    assert "nested/blocked sequence: 600 step calls, export mismatches 0" in out.stdout


def test_handles_of_different_sizes_coexist():
    """the dynamic-LDS ceiling of a kernel is per function, not per handle: a small handle created after a large one
    must not break the large one"""
    big = make_problem("mvn_full", np.random.default_rng(51), N=200, d=32)
    small = make_problem("gaussian", np.random.default_rng(52))
    a = D.HipEngine(n_groups=2, Np=256, D=32, n_rows=4, schedule=2, seed=1)   # 64 KB group tile in LDS
    setup_engine(a, big)
    a.set_state(big["init"](512))
    b = D.HipEngine(n_groups=2, Np=6, D=2, n_rows=4, schedule=2, seed=1)      # a few hundred bytes
    setup_engine(b, small)
    b.set_state(small["init"](12))
    for it in range(1, 4):
        a.step(it, 1)
        b.step(it, 1)
    assert np.isfinite(a.get_state()[1]).all() and np.isfinite(b.get_state()[1]).all()
    a.close()
    b.close()


def test_device_chain_export_equals_host_rekey():
    """demc_export_chains (both layouts) against the slot-keyed history re-keyed with numpy; alpha = 1 so that particle
    ids move between slots on every iteration"""
    import ctypes as C
    from demc_amd import sampler as S
    prob = make_problem("mvn_iso", np.random.default_rng(61), d=5)
    G, Np, Dd, n = 5, 8, 6, 12
    eng = D.HipEngine(n_groups=G, Np=Np, D=Dd, n_rows=n, schedule=2, seed=6, alpha=1.0)
    setup_engine(eng, prob)
    eng.set_state(prob["init"](G * Np))
    eng.step(1, n)
    th, acc, lp = S.rekey_by_id(*eng.get_history(2, n))
    exp = np.concatenate([np.transpose(th, (0, 2, 1)), acc[:, None, :].astype(float), lp[:, None, :]], axis=1)
    got = eng.export_chains(2, n)
    assert np.array_equal(got, exp)
    assert not np.array_equal(eng.get_history(0, n)[3][0], eng.get_history(0, n)[3][-1])  # ids really moved
    julia = np.empty((G * Np, Dd + 2, n - 2))  # Array{Float64,3}(n, D+2, P) column-major == C array [P][D+2][n]
    eng._ck(eng.L.demc_export_chains(eng.h, 2, n, 0, julia.ctypes.data_as(C.POINTER(C.c_double))))
    assert np.array_equal(np.transpose(julia, (2, 1, 0)), exp)
    eng.close()


def test_chain_export_is_repeatable():
    """regression: the staging buffer of demc_export_chains used to be zero-filled on the null stream while the export
    kernel ran on the handle's own (non-blocking) stream, so now and then part of an export came back as zeros.  A
    2000-row export (the size at which it showed about one time in five) repeated 40 times must not change."""
    prob = make_problem("gaussian", np.random.default_rng(62), N=400)
    G, Np, n = 4, 6, 2000
    eng = D.HipEngine(n_groups=G, Np=Np, D=2, n_rows=n, schedule=2, seed=2, trace=0)
    setup_engine(eng, prob)
    eng.set_state(prob["init"](G * Np))
    eng.step(1, n)
    first = eng.export_chains(0, n)
    assert np.isfinite(first).all() and (first[:, 1, :] > 0).all()  # sigma column: a zeroed patch would show here
    for _ in range(40):
        assert np.array_equal(eng.export_chains(0, n), first)
    eng.close()


def test_host_planned_migration_moves_whole_rows():
    """demc_apply_migration = shift_particles! (migration.jl:84-91) with the plan drawn by the caller: dst[k] receives the
    row (theta, weight, id) src[k] held before the call; a cycle is a rotation; everything else is untouched (bit-exact)."""
    rng = np.random.default_rng(5)
    G, Np, Dm = 6, 8, 5
    eng = D.HipEngine(n_groups=G, Np=Np, D=Dm, n_rows=0, store_history=0, seed=3)
    th0 = rng.standard_normal((G * Np, Dm))
    w0 = rng.standard_normal(G * Np)
    id0 = rng.permutation(G * Np).astype(np.int64)
    eng.set_state(th0, w0, id0)
    # the reference's plan: one particle from each of 4 selected groups, shifted circularly (utility_tests.jl:149-154)
    groups = rng.choice(G, 4, replace=False)
    slots = np.array([g * Np + rng.integers(Np) for g in groups], np.int32)
    eng.apply_migration(np.roll(slots, 1), slots)
    th, w, ids = eng.get_state()
    eth, ew, eid = th0.copy(), w0.copy(), id0.copy()
    eth[slots], ew[slots], eid[slots] = th0[np.roll(slots, 1)], w0[np.roll(slots, 1)], id0[np.roll(slots, 1)]
    assert np.array_equal(th, eth) and np.array_equal(w, ew) and np.array_equal(ids, eid)
    assert sorted(ids) == sorted(id0), "a migration permutes particles, it never duplicates one"
    assert np.array_equal(eng.get_weights(), ew)
    eng.apply_migration([], [])  # empty plan: nothing to do
    with pytest.raises(D.DemcError):
        eng.apply_migration([0, 1], [2, 2])  # two rows into one slot
    with pytest.raises(D.DemcError):
        eng.apply_migration([0], [G * Np])  # out of range
    assert np.array_equal(eng.get_state()[0], eth), "a rejected plan must leave the state untouched"
    eng.close()


@pytest.mark.parametrize("mode", ["direct", "streaming", "suffstat"])
def test_bench_contract_on_a_small_workload(mode):
    """bench.py prints ONE JSON line with the driver's keys, a `roofline` object for the dominant kernel and a
    `cpu_baseline` object; checked on a reduced workload so that the line's shape cannot rot unnoticed"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--mode", mode,
                          "--n-groups", "16", "--np", "32", "--nobs", "4000", "--dim", "8", "--accuracy-iters", "300", "--burnin", "100"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 2 and out.stdout.rstrip().endswith(lines[1])  # {"detail": ...} first, the compact line LAST
    r, det = json.loads(lines[1]), json.loads(lines[0])["detail"]
    assert len(lines[1]) < 6000 and r["detail"] == "gpurun_out/bench_detail.json"
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in r, key
    assert r["n_gpus"] == 1 and r["steps"] == 6 and r["warmup"] == 2 and r["dtype"] == "f64" and r["scaling"] == "weak"
    assert r["vs_baseline"] is None and r["higher_is_better"] is True and "workload" in r["config"]
    assert abs(det["value"] - 16 * 32 * 6 / (det["ms_per_step"] * 6e-3)) <= 1e-6 * det["value"]
    assert abs(r["value"] / det["value"] - 1) < 1e-4
    rf = r["roofline"]
    assert rf["bound"] == {"direct": "valu", "streaming": "mfma", "suffstat": "hbm"}[mode] and rf["unit"] == ("GB/s" if mode == "suffstat" else "TFLOP/s")
    if mode == "direct":  # the headline's mode: the fraction in SURVEY 8d's unit, and the clock the vector pipe held beside it
        assert "shader_clock_mhz" in rf and "frac_at_clock" in rf  # (a launch of one round of workgroups has nothing to difference: None)
        assert "3*N*D" in det["roofline"]["flop_counted"] and "THE KERNELS OF THE TIMED REGION" in det["accuracy"]["leg"]
    if mode == "streaming":  # a labelled row since round 6: both fractions, SURVEY's unit prices work the kernel does not do
        assert det["roofline"]["frac_survey"] > det["roofline"]["frac_executed"] == det["roofline"]["frac"]
    assert rf["achieved"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and "traffic" in rf  # (5 significant digits)
    assert abs(det["roofline"]["frac"] - det["roofline"]["achieved"] / det["roofline"]["peak"]) < 1e-12
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"] and cb["value_single_thread"] > 0
    assert rf["frac"] <= 1.0, "a roofline fraction above 1 means the numerator counts work the kernel does not execute"
    assert rf["device_ms_per_iter"] <= r["ms_per_step"] * 1.02, "kernel time of the timed iterations cannot exceed their wall time"
    acc = det["accuracy"]
    assert "timed_chain" in acc and "posterior_mean_l1_rel" in acc and "leg" in acc  # the steps-independent accuracy leg
    assert r["accuracy"]["posterior_mean_l1_rel"] is not None and r["accuracy"]["accept_rate"] is not None


@pytest.mark.parametrize("config,extra", [("cfg1", []), ("cfg3", ["--mode", "direct", "--n-groups", "16", "--np", "32", "--nobs", "4000", "--dim", "8"]),
                                          ("cfg2", ["--mode", "streaming", "--nobs", "2000"]), ("cfg4", ["--nobs", "600", "--n-groups", "4", "--np", "8"]),
                                          ("cfg5", ["--nobs", "500", "--n-groups", "4", "--np", "16"])])
def test_bench_lines_of_the_other_configs(config, extra):
    """--config cfg2 / cfg4 / cfg5 print the same contract with their own roofline definition"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--config", config,
                          "--accuracy-iters", "0"] + extra, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    r, det = json.loads(lines[-1]), json.loads(lines[0])["detail"]
    assert len(lines[-1]) < 6000
    rf = r["roofline"]
    assert r["n_gpus"] == 1 and config in r["config"]["workload"] and r["value"] > 0
    assert rf["bound"] == {"cfg1": "hbm", "cfg3": "valu", "cfg2": "mfma", "cfg4": "hbm", "cfg5": "valu"}[config]
    assert 0 < rf["frac"] <= 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    assert r["cpu_baseline"]["value"] > 0 and det["accuracy"]["timed_chain"]["finite_weights"]
    sweeps = r["config"]["block_sweeps_per_step"]
    assert sweeps == (2 if config == "cfg4" else 1)
    assert abs(det["value"] - r["config"]["particles_per_gpu"] * sweeps * 6 / (det["ms_per_step"] * 6e-3)) <= 1e-6 * det["value"]


def test_bench_rows_and_the_row_flags(tmp_path):
    """the driver's command (`python bench.py`, here with fewer steps) appends `rows`: every named row is there with a roofline
    fraction in (0, 1] and the kernels it ran; and the flags behind two of the newer rows run by hand -- DE-MC_Z past burn-in
    (ONE lean kernel) and the LBA from a converged population"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "10", "--warmup", "3",
                          "--accuracy-iters", "0", "--rows", "cfg3_streaming,cfg3_suffstat,cfg3_suffstat_history_partners_post_burnin,"
                                                            "cfg3_suffstat_history_partners_snooker,cfg4_share,cfg5_share_converged"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    line, r = json.loads(lines[-1]), json.loads(lines[0])["detail"]
    assert len(lines) == 2 and len(lines[-1]) < 6000
    # the compact rows of the last line are the detail's rows, numbers only
    for c, x in zip(line["rows"], r["rows"]):
        assert c["name"] == x["name"] and abs(c["value"] / x["value"] - 1) < 1e-4 and abs(c["frac"] / x["roofline"]["frac"] - 1) < 1e-4
        assert c["bound"] == x["roofline"]["bound"] and c["steps"] == x["steps"] and c["launch_ms"] > 0
    rows = {x["name"]: x for x in r["rows"]}
    assert set(rows) == {"cfg3_streaming", "cfg3_suffstat", "cfg3_suffstat_history_partners_post_burnin", "cfg3_suffstat_history_partners_snooker",
                         "cfg4_share", "cfg5_share_converged"}
    for name, x in rows.items():
        assert "error" not in x and x["value"] > 0 and x["finite_weights"], (name, x.get("error"))
        assert x["steps"] == dict(bench.ROWS)[name]["steps"]
        f = x["roofline"]["frac"]
        assert f is not None and 0 < f <= 1.0, (name, f)
    assert rows["cfg3_suffstat_history_partners_post_burnin"]["kernels"] == "k_res_mvn<512,false,32,1>"
    # (a row's sampler settings must reach its engine: the workload object is shared between the rows of a configuration)
    assert rows["cfg3_suffstat_history_partners_snooker"]["kernels"] == "k_res_mvn<512,false,32,3>"
    assert rows["cfg4_share"]["kernels"] == "k_longrow<512>" and "k_lba_wave" in rows["cfg5_share_converged"]["kernels"]
    # STREAMING is a labelled row: the executed MFMA flop at most the peak, SURVEY 8d's unit over the same time ABOVE it (the
    # [proposals x D].[D x N] product it executes is zero after centring: not the survey's per-pair work)
    st = rows["cfg3_streaming"]["roofline"]
    assert st["frac"] == st["frac_executed"] <= 1.0 < st["frac_survey"] and "null GEMM" in st["label"]
    ctx = r["headline_context"]
    assert ctx["streaming_frac_survey"] == st["frac_survey"] and ctx["suffstat_value"] == rows["cfg3_suffstat"]["value"]
    assert abs(ctx["cpu_baseline_like_for_like_ratio"] - r["value"] / r["cpu_baseline"]["value"]) < 1e-9 * ctx["cpu_baseline_like_for_like_ratio"]
    compact = {c["name"]: c for c in line["rows"]}
    assert compact["cfg3_streaming"]["frac_survey"] > 1.0 and "frac_survey" not in compact["cfg4_share"]
    # one GPU's share of BASELINE's 8-GPU cfg4 carries its own CPU leg (the oracle, reference schedule)
    cb = rows["cfg4_share"]["cpu_baseline"]
    assert cb["value"] > 0 and cb["value_single_thread"] > 0 and cb["kind"] == "port" and "cfg4" in cb["sample"]
    assert abs(compact["cfg4_share"]["cpu"] / cb["value"] - 1) < 1e-4 and rows["cfg4_share"]["gpu_over_cpu"] > 1
    # the headline: DIRECT, the fraction in SURVEY 8d's unit
    assert r["metric"].startswith("particle-updates/sec") and "cfg3" in r["config"]["workload"] and "loglike=direct" in r["config"]["workload"]
    assert r["roofline"]["bound"] == "valu" and 0.4 < r["roofline"]["frac"] <= 0.75 and "k_direct_mvn<32>" in r["kernels"]
    # ... next to the clock the vector pipe held under it (in-kernel: s_memtime over s_memrealtime, per XCD) -- what moves a VALU-bound
    # fraction between boxes; the fraction at that clock cannot exceed the instruction mix's 0.75 either
    rf = r["roofline"]
    assert 1000.0 < rf["shader_clock_mhz_min"] <= rf["shader_clock_mhz"] <= rf["shader_clock_mhz_max"] < 2450.0, rf
    assert rf["frac"] <= rf["frac_at_clock"] <= 0.76 and abs(rf["frac_of_mix_ceiling_at_clock"] - rf["frac_at_clock"] / 0.75) < 1e-9
    N, d, P = 100000, 32, 65536
    assert 3.0 * N * d * P / (r["ms_per_step"] * 1e-3) / 1e12 <= 78.6, "SURVEY 8d's flop per step over the step's wall time cannot exceed the peak"


def test_rccl_all_gather_path_at_world_size_one():
    """the ShardedDriver with a REAL process group (backend nccl = RCCL) at world_size 1: pack -> all_gather_into_tensor ->
    apply, enqueued stream-ordered on torch's current stream, reproduces demc_step's on-device migration bit for bit"""
    import os
    import torch
    import torch.distributed as dist
    from demc_amd.distributed import ShardedDriver
    from conftest import make_problem, setup_engine
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        prob = make_problem("mvn_full", np.random.default_rng(31), N=200, d=6)
        G, Np, d, n_it = 6, 16, 6, 40
        th0 = prob["init"](G * Np)
        outs = []
        for sharded in (False, "sync", "async"):  # async: the gather on a side stream, unselected groups update meanwhile
            eng = D.HipEngine(n_groups=G, Np=Np, D=d, n_rows=n_it, schedule=2, seed=17, alpha=0.4, burnin=10, trace=0, loglike_mode=1)
            setup_engine(eng, prob)
            eng.set_state(th0)
            if sharded:
                drv = ShardedDriver(eng, dist, torch.device("cuda", 0), stream_ordered=True, async_migration=sharded == "async")
                drv.step(1, n_it)
                torch.cuda.synchronize()
                assert drv.n_exchanges >= 5 and drv.dist is not None
            else:
                eng.step(1, n_it)
            outs.append(eng.get_history(0, n_it) + eng.get_state())
            eng.close()
        for other in outs[1:]:
            for x, y in zip(outs[0], other):
                assert np.array_equal(x, y)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("family,kw", [("mvn_full", dict(loglike_mode=1)), ("mvn_full", dict(loglike_mode=0)), ("gaussian", dict()),
                                       ("hier_binomial", dict())])
def test_asynchronous_migration_equals_the_synchronous_exchange(family, kw):
    """the groups an exchange did not select update (demc_update_groups_async) before the exchange is applied
    and the selected ones after it; groups never interact inside update!, so history and state equal demc_step's bit for bit.
    Covers the resident, per-phase and long-row kernels running on a SUBSET of the handle's groups."""
    import torch
    from demc_amd.distributed import ShardedDriver
    from conftest import make_problem, setup_engine
    prob = make_problem(family, np.random.default_rng(51), N=300, d=6, S=2100)
    G, Np, n_it = 10, 8, 30
    Dd = prob["D"]
    th0 = prob["init"](G * Np)
    outs = []
    for mode in ("step", "async"):
        eng = D.HipEngine(n_groups=G, Np=Np, D=Dd, n_rows=n_it, schedule=2, seed=23, alpha=0.5, burnin=10, trace=0, **kw)
        setup_engine(eng, prob)
        eng.set_state(th0)
        if mode == "step":
            eng.step(1, n_it)
        else:
            drv = ShardedDriver(eng, None, torch.device("cuda", 0), stream_ordered=True, async_migration=True)
            drv.step(1, n_it)
            drv.synchronize()
            assert drv.n_exchanges >= 8
        outs.append(eng.get_history(0, n_it) + eng.get_state())
        eng.close()
    for i, (x, y) in enumerate(zip(*outs)):
        if i in (2, 5) and kw.get("loglike_mode") == 0:
            # STREAMING: demc_step runs the streaming-resident form, the subset updates the K1 -> K2 -> K3 chain: the
            # observation sums are split differently (log-densities to rounding, everything else exact)
            np.testing.assert_allclose(x, y, rtol=1e-10)
        else:
            assert np.array_equal(x, y), f"array {i}"


def test_geometry_groups_makes_shards_reproduce_the_unsharded_run():
    """Lanes per particle are chosen from the population, so a shard of 128 groups would pick a different
    split than the 256-group run it is a part of, and its log-densities would differ in their last bits (sum order).
    demc_config.geometry_groups pins the geometry of the unsharded run: two handles of 128 groups == one of 256, bit for bit,
    at a size that straddles a threshold (per-phase form, D = 32, Np = 128: 256 groups -> 8 lanes per particle, 128 groups
    -> 16)."""
    from conftest import make_problem, setup_engine
    prob = make_problem("mvn_full", np.random.default_rng(41), N=300, d=32)
    G, Np, d, n_it = 256, 128, 32, 6
    th0 = prob["init"](G * Np)

    def run(groups, offset, geometry_groups):
        eng = D.HipEngine(n_groups=groups, Np=Np, D=d, n_rows=n_it, schedule=2, seed=5, alpha=0.0, burnin=3, trace=0, loglike_mode=1,
                          fuse=2, group_offset=offset, n_groups_total=G, geometry_groups=geometry_groups)
        setup_engine(eng, prob)
        eng.set_state(th0[offset * Np:(offset + groups) * Np])
        eng.update(1, n_it)
        out = eng.get_history(0, n_it)[:3] + eng.get_state()[:2]
        eng.close()
        return out

    whole = run(G, 0, 0)
    halves = [run(G // 2, 0, G), run(G // 2, G // 2, G)]
    for i, x in enumerate(whole):
        y = np.concatenate([h[i] for h in halves], axis=1 if i < 3 else 0)
        assert np.array_equal(x, y), f"array {i}"
    # without the pin the accept decisions still agree, the log-densities only to rounding
    loose = [run(G // 2, 0, 0), run(G // 2, G // 2, 0)]
    np.testing.assert_allclose(np.concatenate([h[4] for h in loose]), whole[4], rtol=1e-11)


# ---------------------------------------------------------------------------------------------------------------------
# Statistical gates of the race models (the reference's only multi-group, threaded, non-Gaussian test) and the BASELINE
# cfg4 / cfg5 shapes at one GPU's full share
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_sequential(orc):
    def make(**cfg):
        cfg = dict(cfg)
        cfg["schedule"] = 0  # the reference's own sequential in-place sweep
        cfg["n_threads"] = 8
        return orc.Oracle(**{k: v for k, v in cfg.items() if k in orc.CFG_KEYS})
    return make


def test_lognormal_race_tests_jl(orc):
    """test/lognormal_race_tests.jl:1-66: LNR(nu = [-2,-2,-3,-3], sigma = 1, tau = .5), 100 trials, nu ~ N(0,3), tau ~ U(0, min rt),
    DE(burnin = 2000, Np = 24, n_groups = 4), 5000 iterations under MCMCThreads.  The reference compares mean and sd with NUTS at
    rtol 0.05 (needs Turing); here the comparison chain is the CPU oracle running the reference's own schedule (sequential
    sweep) on the same model -- a different engine, schedule and random stream -- at the same tolerance, plus rhat."""
    rng = np.random.default_rng(9918)
    nu, tau, N = np.array([-2.0, -2.0, -3.0, -3.0]), 0.5, 100
    t = np.exp(rng.normal(nu, 1.0, (N, 4)))
    choice, rt = t.argmin(1) + 1.0, t.min(1) + tau
    min_rt = float(rt.min())

    def run(factory, seed):
        r2 = np.random.default_rng(seed)
        sp = lambda: [r2.normal(0, 3, 4), r2.uniform(0, min_rt)]
        model = D.DEModel(sample_prior=sp, prior_loglike=D.Priors(ν=D.Normal(0, 3), τ=D.Uniform(0.0, min_rt)),
                          loglike=D.LNRLikelihood(sigma=1.0), data=(choice, rt), names=("ν", "τ"))
        de = D.DE(sample_prior=sp, bounds=((-np.inf, np.inf), (0.0, min_rt)), burnin=2000, Np=24, n_groups=4)
        return D.sample(model, de, D.MCMCThreads(), 5000, engine_factory=factory).describe()

    gpu = run(None, 68541)
    ref = run(_oracle_sequential(orc), 1234)
    assert len(gpu) == 5 and set(gpu) == set(ref)
    for nm in gpu:
        assert abs(gpu[nm]["rhat"] - 1.0) < 0.05, (nm, gpu[nm])
        assert abs(gpu[nm]["mean"] - ref[nm]["mean"]) <= 0.05 * abs(ref[nm]["mean"]) + 0.01, (nm, gpu[nm], ref[nm])
        assert abs(gpu[nm]["std"] - ref[nm]["std"]) <= 0.08 * ref[nm]["std"], (nm, gpu[nm], ref[nm])
    # and the posterior sits where the data were generated (wide tolerances: 100 trials)
    for j in range(4):
        assert abs(gpu[f"ν[{j + 1}]"]["mean"] - nu[j]) < 4 * gpu[f"ν[{j + 1}]"]["std"]


def test_run_lba_jl_parameter_recovery(orc):
    """Examples/Run_LBA.jl:6-47: LBA(nu = [3,2], A = .8, k = .2, tau = .3), 100 trials, the example's priors and bounds,
    DE(burnin = 1500, n_groups = 3, Np = 15), 3000 iterations under MCMCThreads: the chain recovers the generating parameters
    (each inside the chain's central mass) and agrees with the oracle's sequential-schedule chain on the same model"""
    from demc_amd.workloads import simulate_lba
    rng = np.random.default_rng(88484)
    truth = dict(nu=(3.0, 2.0), A=0.8, k=0.2, tau=0.3)
    choice, rt = simulate_lba(rng, 100, truth["nu"], truth["A"], truth["k"], truth["tau"])
    min_rt = float(rt.min())

    def run(factory, seed):
        r2 = np.random.default_rng(seed)
        sp = lambda: [np.abs(r2.normal(1, 5, 2)), abs(r2.normal(0.8, 0.2)), abs(r2.normal(0.2, 0.1)), r2.uniform(0, min_rt)]
        model = D.DEModel(sample_prior=sp, loglike=D.LBALikelihood(), data=(choice, rt), names=("ν", "A", "k", "τ"),
                          prior_loglike=D.Priors(ν=D.Normal(1, 5), A=D.Normal(0.8, 0.2), k=D.Normal(0.2, 0.1), τ=D.Uniform(0, min_rt)))
        de = D.DE(sample_prior=sp, bounds=((0.0, np.inf), (0.0, np.inf), (0.0, np.inf), (0.0, min_rt)), burnin=1500, n_groups=3, Np=15)
        return D.sample(model, de, D.MCMCThreads(), 3000, engine_factory=factory)

    ch = run(None, 5)
    d = ch.describe()
    flat = dict(zip(["ν[1]", "ν[2]", "A", "k", "τ"], [3.0, 2.0, 0.8, 0.2, 0.3]))
    for nm, tv in flat.items():
        x = ch[nm].ravel()
        lo, hi = np.quantile(x, [0.005, 0.995])
        assert lo < tv < hi, (nm, tv, lo, hi)
        assert abs(d[nm]["rhat"] - 1.0) < 0.1, (nm, d[nm])
    assert d["ν[1]"]["mean"] > d["ν[2]"]["mean"]          # the faster accumulator is recovered as the faster one
    ref = run(_oracle_sequential(orc), 77).describe()
    for nm in flat:
        assert abs(d[nm]["mean"] - ref[nm]["mean"]) <= 0.1 * abs(ref[nm]["mean"]) + 0.02, (nm, d[nm], ref[nm])
        assert abs(d[nm]["std"] - ref[nm]["std"]) <= 0.2 * ref[nm]["std"] + 0.005, (nm, d[nm], ref[nm])


def _full_share(name, orc, n_iter, spot_rows, rtol):
    """one GPU's share of a BASELINE config at FULL size: size-independent properties + an oracle spot check"""
    from demc_amd import workloads as W
    w = W.BUILDERS[name]()
    G, Np, Dd = w["G"], w["Np"], w["D"]
    P = G * Np
    eng = D.HipEngine(n_groups=G, Np=Np, D=Dd, n_rows=n_iter, schedule=2, seed=20260001, burnin=1000, trace=0, **w["engine"])
    W.configure(eng, w)
    th0 = w["init"](P, np.random.default_rng(3))
    eng.set_state(th0)
    _, w0, _ = eng.get_state()
    assert np.isfinite(w0).all()
    o = orc.Oracle(n_groups=1, Np=4, D=Dd, n_rows=0, store_history=0, n_threads=8, **w["engine"])
    W.configure(o, w)
    sub = np.linspace(0, P - 1, spot_rows).astype(int)
    np.testing.assert_allclose(w0[sub], o.logpost(th0[sub]), rtol=rtol)       # initial evaluation vs the oracle
    eng.step(1, n_iter)
    th, wts, ids = eng.get_state()
    assert np.array_equal(np.sort(ids), np.arange(P))                          # ids stay a permutation
    fin = np.isfinite(wts)
    assert fin.all()
    np.testing.assert_allclose(eng.logpost(th[sub]), wts[sub], rtol=1e-10)     # stored weights re-evaluate to themselves
    np.testing.assert_allclose(wts[sub], o.logpost(th[sub]), rtol=rtol)        # ... and to the oracle's value
    hth, hacc, hlp, hid = eng.get_history(n_iter - 1, n_iter)
    assert np.array_equal(hth[0], th) and np.array_equal(hid[0], ids) and np.array_equal(hlp[0], wts)
    assert 0.02 < hacc.mean() < 0.98
    lo, hi = np.asarray(w["lo"]), np.asarray(w["hi"])
    assert ((th >= lo) & (th <= hi)).all()                                     # accepted rows are in bounds
    eng.close()
    o.close()
    return wts.mean() - w0.mean()


def test_cfg4_full_share_properties(orc):
    """BASELINE cfg4 at one GPU's share: 16 groups x 32 particles, hierarchical Binomial with 1e4 subjects (D = 10 002),
    two block sweeps [hyper ; subject] per iteration -- the long-row kernel at the size bench.py --config cfg4 runs"""
    gain = _full_share("cfg4", orc, n_iter=6, spot_rows=6, rtol=1e-9)
    assert gain > 0  # prior draws climb towards the posterior


def test_cfg5_full_share_properties(orc):
    """BASELINE cfg5 at one GPU's share: 64 groups x 128 particles, LBA with 3 accumulators, 5e4 trials simulated from
    nu = (3,2,1), A = .8, k = .2, tau = .3, snooker 0.1 (the data bench.py --config cfg5 runs)"""
    gain = _full_share("cfg5", orc, n_iter=4, spot_rows=8, rtol=1e-5)
    assert gain > 0


# ---------------------------------------------------------------------------------------------------------------------
# Whole-row plug-in (demc_set_model_source_row): user-written prior_loglike / loglike of ANY shape (structs.jl:176-189)
# ---------------------------------------------------------------------------------------------------------------------
HIER_ROW_SRC = r"""
// Examples/Hierarchical_Example.jl:26-44 written by a user: theta = (mu_b0, sd_b0, b0[1:S], sigma), data Y[S][n]
//   prior_loglike: mu_b0 ~ Normal(1,1), sd_b0 ~ truncated(Cauchy(0,1),0,Inf), b0_s ~ Normal(0, sd_b0), sigma ~ truncated(Cauchy(0,1),0,Inf)
//   loglike:       y_{s,i} ~ Normal(mu_b0 + b0_s, sigma)
__device__ double norm_lpdf(double x, double m, double s) { const double z = (x - m) / s; return -(z * z + 1.8378770664093453) / 2.0 - log(s); }
__device__ double hcauchy_lpdf(double x) { return x < 0.0 ? -INFINITY : 0.6931471805599453 - 1.1447298858494002 - log1p(x * x); }
__device__ double demc_user_loglike_row(const double* th, int D, const double* Y, const long long* dims, int ndims,
                                        const double* hyper, int nhyper, int lane, int n_lanes) {
    const long long S = dims[0], n = dims[1];
    const double mu0 = th[0], sg = th[2 + S];
    double acc = 0.0;
    for (long long s = lane; s < S; s += n_lanes)
        for (long long i = 0; i < n; ++i) acc += norm_lpdf(Y[s * n + i], mu0 + th[2 + s], sg);
    return acc;
}
__device__ double demc_user_prior_row(const double* th, int D, const double* hyper, int nhyper, int lane, int n_lanes) {
    const int S = D - 3;
    double acc = 0.0;
    if (lane == 0) acc = norm_lpdf(th[0], 1.0, 1.0) + hcauchy_lpdf(th[1]) + hcauchy_lpdf(th[2 + S]);
    for (int s = lane; s < S; s += n_lanes) acc += norm_lpdf(th[2 + s], 0.0, th[1]);
    return acc;
}
"""


def test_user_written_hierarchical_model_equals_the_registered_family():
    """the reference's hierarchical Gaussian (a prior that depends on another parameter, a likelihood
    indexed by subject) registered as USER SOURCE, no built-in family involved: log-posteriors equal FAM_HIER_GAUSSIAN's to
    1e-12, out-of-bounds rows stay -Inf, and a sampling run with block updates makes the same accept decisions."""
    from conftest import make_problem, setup_engine
    rng = np.random.default_rng(81)
    prob = make_problem("hier_gaussian", rng, S=37, n=9)
    Dd, G, Np = prob["D"], 4, 8
    th = prob["init"](G * Np)
    th[3, 1] = -0.5  # sd_b0 < 0: out of bounds -> -Inf, neither function is called (utilities.jl:92-99)
    a = D.HipEngine(n_groups=G, Np=Np, D=Dd, schedule=2, n_rows=12, seed=4, burnin=6)
    setup_engine(a, prob)
    b = D.HipEngine(n_groups=G, Np=Np, D=Dd, schedule=2, n_rows=12, seed=4, burnin=6)
    b.set_model_source_row(HIER_ROW_SRC, prob["data"], prob["dims"], has_prior=True)
    b.set_bounds(prob["lo"], prob["hi"])   # priors: the table stays flat, the user's function is the prior
    la, lb = a.logpost(th), b.logpost(th)
    assert la[3] == -np.inf and lb[3] == -np.inf
    fin = np.isfinite(la)
    assert fin.sum() == G * Np - 1
    np.testing.assert_allclose(lb[fin], la[fin], rtol=1e-12)
    masks = np.zeros((2, Dd), np.uint8)   # blocks [hyper ; subject] (Examples/Hierarchical_Example.jl:88-92)
    masks[0, :2] = 1
    masks[0, -1] = 1
    masks[1] = 1 - masks[0]
    th0 = prob["init"](G * Np)
    for e in (a, b):
        e.set_blocks(masks)
        e.set_state(th0)
        e.step(1, 12)
    assert "k_user_row" in b.last_kernels()
    ha, hb = a.get_history(0, 12), b.get_history(0, 12)
    assert np.array_equal(ha[1], hb[1]) and np.array_equal(ha[3], hb[3]) and ha[1].mean() > 0.05
    np.testing.assert_allclose(hb[0], ha[0], rtol=1e-12)
    np.testing.assert_allclose(hb[2], ha[2], rtol=1e-11)
    # a source that lacks the entry point fails loudly, with the compiler's message
    c = D.HipEngine(n_groups=G, Np=Np, D=Dd, schedule=2)
    with pytest.raises(D.DemcError) as err:
        c.set_model_source_row("__device__ double something_else() { return 0.0; }", prob["data"], prob["dims"])
    assert err.value.code in (D._ffi.EINVAL, D._ffi.EHIP)
    for e in (a, b, c):
        e.close()


def test_user_row_model_through_sample():
    """the same through DEModel / DE / sample: SourceLikelihood(row=True, has_prior=True)"""
    rng = np.random.default_rng(82)
    S, n = 6, 30
    b0 = rng.normal(0, 0.7, S)
    Y = 1.0 + b0[:, None] + rng.normal(0, 0.5, (S, n))
    sp = lambda: [rng.normal(1, 1), abs(rng.standard_cauchy()) + 0.2, rng.normal(0, 1, S), abs(rng.standard_cauchy()) + 0.2]
    model = D.DEModel(sample_prior=sp, prior_loglike=D.Priors(mu_b0=D.Flat(), sd_b0=D.Flat(), b0=D.Flat(), sigma=D.Flat()),
                      loglike=D.SourceLikelihood(HIER_ROW_SRC, row=True, has_prior=True), data=Y, names=("mu_b0", "sd_b0", "b0", "sigma"))
    de = D.DE(sample_prior=sp, bounds=((-np.inf, np.inf), (0.0, np.inf), (-np.inf, np.inf), (0.0, np.inf)), burnin=1500, Np=12, n_groups=4)
    ch = D.sample(model, de, D.HIPBackend(seed=9), 4000)
    d = ch.describe()
    assert abs(d["sigma"]["mean"] - 0.5) < 0.06
    got = np.array([d[f"b0[{s + 1}]"]["mean"] + d["mu_b0"]["mean"] for s in range(S)])
    np.testing.assert_allclose(got, Y.mean(1), atol=0.08)   # subject means are well identified (30 observations each)


def test_padded_history_cells_round_trip_through_the_abi():
    """With history partners the device pads a history cell to whole cache lines (KParams::hist_ld: D = 31 -> 32 doubles); the C-ABI
    keeps its dense [rows][P][D] layout: rows written with demc_set_history_rows come back from demc_get_history bit for bit, also
    when the transfer takes several chunks of the staging buffer (40 rows of 16 384 x 31 doubles = 162 MB through 64 MB), at an offset,
    and the chain export reads the padded cells."""
    from conftest import make_problem, setup_engine
    G, Np, d, n_rows = 256, 64, 30, 48
    prob = make_problem("mvn_iso", np.random.default_rng(5), N=50, d=d)
    e = D.HipEngine(n_groups=G, Np=Np, D=d + 1, n_rows=n_rows, n_initial=4, schedule=1, partner_kind=1, seed=3, loglike_mode=1)
    setup_engine(e, prob)
    rows = np.random.default_rng(6).normal(0, 1, (40, G * Np, d + 1))
    e.set_history_rows(3, rows)
    back = e.get_history(3, 43)[0]
    assert np.array_equal(back, rows)
    assert np.array_equal(e.get_history(10, 12)[0], rows[7:9])
    e.set_history_rows(0, rows[:4] * 2.0)
    assert np.array_equal(e.get_history(0, 5)[0], np.concatenate([rows[:3] * 2.0, rows[3:4] * 2.0, rows[1:2]]))
    e.close()
