"""End-to-end gates of the CPU oracle, mirroring the reference's statistical tests with closed-form targets
(Turing/NUTS is not available here): test/binomial_tests.jl, test/gaussian_tests.jl, test/blocking_tests.jl."""
import numpy as np
import pytest
from scipy import stats

from demc_amd import families as F


def _binomial(orc, schedule, seed):
    N, k = 10, 6
    o = orc.Oracle(n_groups=4, Np=4, D=1, burnin=1500, n_rows=3000, schedule=schedule, seed=seed)
    o.set_model(F.FAM_BINOMIAL, [N, k], [1])
    o.set_priors([F.PRIOR_BETA], [1.0], [1.0])
    o.set_bounds([0.0], [1.0])
    o.set_state(np.random.default_rng(seed).uniform(0, 1, (16, 1)))
    o.step(1, 3000)
    th = o.get_history(1500, 3000)[0]
    return th.ravel(), stats.beta(k + 1, N - k + 1)


@pytest.mark.parametrize("schedule", [0, 1, 2])
def test_binomial_posterior_is_beta(orc, schedule):
    """test/binomial_tests.jl:35-37: mean and sd vs Beta(k+1, N-k+1), rtol 0.02 (Np=4 so that two_colour is defined)"""
    x, sol = _binomial(orc, schedule, 29542)
    assert abs(x.mean() - sol.mean()) <= 0.02 * sol.mean()
    assert abs(x.std() - sol.std()) <= 0.03 * sol.std()


@pytest.mark.parametrize("schedule", [0, 1, 2])
def test_gaussian_posterior_matches_grid(orc, schedule):
    """test/gaussian_tests.jl:57-59 with the NUTS target replaced by a numerically integrated grid posterior
    (no conjugate form with the Cauchy+ prior): means and sds atol 0.01 would need ~1e5 effective draws, so the
    gate here is 3 Monte-Carlo standard errors + 0.01"""
    rng = np.random.default_rng(973536)
    data = rng.normal(0, 1, 50)
    o = orc.Oracle(n_groups=4, Np=6, D=2, burnin=1500, n_rows=6000, schedule=schedule, seed=7)
    o.set_model(F.FAM_GAUSSIAN, data, [50])
    o.set_priors([F.PRIOR_NORMAL, F.PRIOR_HALFCAUCHY], [0, 0], [10, 1])
    o.set_bounds([-np.inf, 0], [np.inf, np.inf])
    o.set_state(np.stack([rng.normal(0, 10, 24), np.abs(rng.standard_cauchy(24))], 1))
    o.step(1, 6000)
    h = o.get_history(1500, 6000)[0].reshape(-1, 2)
    mu, sg = np.linspace(-1.2, 1.2, 481), np.linspace(0.5, 2.2, 481)
    M, S = np.meshgrid(mu, sg, indexing="ij")
    lp = (-0.5 * ((data[None, None] - M[..., None]) / S[..., None]) ** 2).sum(-1) - 50 * np.log(S) - 0.5 * (M / 10) ** 2 - np.log1p(S ** 2)
    p = np.exp(lp - lp.max())
    p /= p.sum()
    em, es = (p * M).sum(), (p * S).sum()
    sm, ss = np.sqrt((p * M ** 2).sum() - em ** 2), np.sqrt((p * S ** 2).sum() - es ** 2)
    assert abs(h[:, 0].mean() - em) < 0.02 and abs(h[:, 1].mean() - es) < 0.02
    assert abs(h[:, 0].std() - sm) < 0.02 and abs(h[:, 1].std() - ss) < 0.02


def test_blocking_updates_both_blocks(orc):
    """test/blocking_tests.jl: blocks [[true,false],[false,true]] on Normal(0,1) data, means ~ (0,1) atol 0.1"""
    rng = np.random.default_rng(58122)
    data = rng.normal(0, 1, 1000)
    o = orc.Oracle(n_groups=4, Np=6, D=2, burnin=1000, n_rows=2000, schedule=2, seed=5)
    o.set_model(F.FAM_GAUSSIAN, data, [1000])
    o.set_priors([F.PRIOR_NORMAL, F.PRIOR_HALFCAUCHY], [0, 0], [10, 1])
    o.set_bounds([-np.inf, 0], [np.inf, np.inf])
    o.set_blocks(np.array([[1, 0], [0, 1]], np.uint8))
    o.set_state(np.stack([rng.normal(0, 10, 24), np.abs(rng.standard_cauchy(24))], 1))
    o.step(1, 2000)
    h = o.get_history(1000, 2000)[0].reshape(-1, 2)
    assert abs(h[:, 0].mean() - data.mean()) < 0.1 and abs(h[:, 1].mean() - 1.0) < 0.1


def test_resample_snooker_mvn_iso(orc):
    """test/multivariate_normal_tests.jl (reduced): sample = resample, snooker, n_groups = 1, Np = 3"""
    rng = np.random.default_rng(505514)
    d, nd = 6, 100
    X = rng.normal(0, 1, (nd, d))
    n_init = (d + 1) * 4
    n_iter = 12000
    o = orc.Oracle(n_groups=1, Np=3, D=d + 1, burnin=3000, n_initial=n_init, n_rows=n_iter + n_init, schedule=1,
                   partner_kind=1, theta_snooker=0.1, seed=9)
    o.set_model(F.FAM_MVN_ISO, X, [nd, d])
    o.set_priors([1] * d + [2], [0] * (d + 1), [1] * (d + 1))
    o.set_bounds([-np.inf] * d + [0], [np.inf] * (d + 1))
    rows = np.concatenate([rng.normal(0, 1, (n_init, 3, d)), np.abs(rng.standard_cauchy((n_init, 3, 1)))], 2)
    o.set_history_rows(0, rows)
    o.set_state(rows[0])
    o.step(1 + n_init, n_iter)
    h = o.get_history(n_init + 3000, n_init + n_iter)[0].reshape(-1, d + 1)
    np.testing.assert_allclose(h[:, :d].std(0), 0.1, atol=0.015)     # posterior sd of each mean = 1/sqrt(100)
    assert np.corrcoef(X.mean(0), h[:, :d].mean(0))[0, 1] > 0.98


def test_determinism_and_seed_sensitivity(orc):
    def run(seed, threads):
        rng = np.random.default_rng(3)
        o = orc.Oracle(n_groups=6, Np=5, D=2, n_rows=40, schedule=0, seed=seed, n_threads=threads)
        o.set_model(F.FAM_GAUSSIAN, rng.normal(size=20), [20])
        o.set_priors([1, 2], [0, 0], [10, 1])
        o.set_bounds([-np.inf, 0], [np.inf, np.inf])
        o.set_state(np.stack([rng.normal(size=30), rng.uniform(0.5, 2, 30)], 1))
        o.step(1, 40)
        return o.get_history(0, 40)
    a, b, c = run(1, 1), run(1, 4), run(2, 1)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)  # thread count never changes results (per-group addressed RNG)
    assert not np.array_equal(a[0], c[0])


def test_two_shards_equal_one(orc):
    """sharding groups + one exchange per migration event reproduces the single-shard run exactly"""
    rng = np.random.default_rng(4)
    data = rng.normal(size=25)
    th0 = np.stack([rng.normal(size=32), rng.uniform(0.5, 2, 32)], 1)

    def mk(ng, off):
        o = orc.Oracle(n_groups=ng, Np=4, D=2, n_rows=30, schedule=2, seed=77, alpha=0.5, group_offset=off,
                       n_groups_total=8)
        o.set_model(F.FAM_GAUSSIAN, data, [25])
        o.set_priors([1, 2], [0, 0], [10, 1])
        o.set_bounds([-np.inf, 0], [np.inf, np.inf])
        return o
    one = mk(8, 0)
    one.set_state(th0)
    one.step(1, 30)
    a, b = mk(4, 0), mk(4, 4)
    a.set_state(th0[:16])
    b.set_state(th0[16:])
    n_mig = 0
    for it in range(1, 31):
        if a.migration_due(it):
            rows = np.concatenate([a.migration_pack(it), b.migration_pack(it)])
            a.migration_apply(it, rows)
            b.migration_apply(it, rows)
            n_mig += 1
        a.update(it, 1)
        b.update(it, 1)
    assert n_mig > 5
    t1, w1, i1 = one.get_state()
    ta, wa, ia = a.get_state()
    tb, wb, ib = b.get_state()
    np.testing.assert_array_equal(t1, np.concatenate([ta, tb]))
    np.testing.assert_array_equal(i1, np.concatenate([ia, ib]))
    h1 = one.get_history(0, 30)
    ha, hb = a.get_history(0, 30), b.get_history(0, 30)
    for x, y, z in zip(h1, ha, hb):
        np.testing.assert_array_equal(x, np.concatenate([y, z], axis=1))
