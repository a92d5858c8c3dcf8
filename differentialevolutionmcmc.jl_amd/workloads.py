"""Synthetic workloads of the BASELINE.json configs, shared by bench.py, the full-size GPU tests and tools/.

Each builder returns a dict that an engine (HipEngine, or the CPU oracle in bench.py's cpu_baseline leg / in tests) is
configured from with `configure(engine, w)`:
    fam, data, dims, hyper     demc_set_model arguments
    pk, pa, pb, pref           per-scalar prior table (demc_set_priors)
    lo, hi                     per-scalar bounds
    masks                      block masks or None (demc_set_blocks)
    D, G, Np, engine           geometry and extra demc_config fields (theta_snooker, ...)
    init(P, rng)               starting rows: prior draws, like sample_init (main.jl:263-271)
There is no network: all data are generated here from the seeds SURVEY 8d names.
"""
import numpy as np

from . import families as F

INF = np.inf


def mvn_full(d, N, seed):
    """cfg2 / cfg3: theta = mu in R^d, known Sigma = A A'/d + 0.5 I, X rows ~ N(mu*, Sigma), prior mu_j ~ N(0, 1)"""
    rng = np.random.default_rng(seed)
    A = rng.normal(0, 1, (d, d))
    Sigma = A @ A.T / d + 0.5 * np.eye(d)
    mu = rng.normal(0, 1, d)
    X = np.ascontiguousarray(mu + rng.normal(0, 1, (N, d)) @ np.linalg.cholesky(Sigma).T)
    Ainv = np.linalg.inv(Sigma)
    post_cov = np.linalg.inv(N * Ainv + np.eye(d))  # conjugate posterior (prior N(0, I), known Sigma)
    return dict(fam=F.FAM_MVN_FULL, data=X, dims=[N, d], hyper=Sigma, D=d, pk=[F.PRIOR_NORMAL] * d, pa=[0.0] * d,
                pb=[1.0] * d, pref=[0] * d, lo=[-INF] * d, hi=[INF] * d, masks=None, engine={},
                init=lambda P, rng: rng.normal(0, 1, (P, d)), truth=mu,
                posterior_mean=post_cov @ (N * Ainv @ X.mean(0)), posterior_sd=np.sqrt(np.diag(post_cov)))


def simulate_lba(rng, N, nu, A, k, tau):
    """N trials of the linear ballistic accumulator (Brown & Heathcote 2008; SequentialSamplingModels conventions used by
    Examples/Run_LBA.jl:33-37: b = A + k, sigma = 1, drifts redrawn until one is positive): start a_i ~ U(0, A),
    drift d_i ~ N(nu_i, 1), finishing time (b - a_i)/d_i for d_i > 0; choice = first finisher, rt = its time + tau."""
    nu = np.asarray(nu, dtype=np.float64)
    na = nu.size
    b = A + k
    d = rng.normal(nu, 1.0, (N, na))
    bad = ~(d > 0).any(1)
    while bad.any():
        d[bad] = rng.normal(nu, 1.0, (int(bad.sum()), na))
        bad = ~(d > 0).any(1)
    a = rng.uniform(0, A, (N, na))
    with np.errstate(divide="ignore"):
        t = np.where(d > 0, (b - a) / np.where(d > 0, d, 1.0), np.inf)
    choice = t.argmin(1) + 1
    rt = t.min(1) + tau
    return choice.astype(np.float64), rt


def cfg1(seed=20260000):
    """Examples/Gaussian_Example.jl:11-28: data rand(Normal(0,1), 50), mu ~ N(0,1), sigma ~ Cauchy+(0,1)"""
    rng = np.random.default_rng(seed)
    data = rng.normal(0, 1, 50)
    return dict(name="cfg1", G=4, Np=10, fam=F.FAM_GAUSSIAN, data=data, dims=[50], hyper=None, D=2,
                pk=[F.PRIOR_NORMAL, F.PRIOR_HALFCAUCHY], pa=[0, 0], pb=[1, 1], pref=[0, 0], lo=[-INF, 0], hi=[INF, INF],
                masks=None, engine={},
                init=lambda P, rng: np.stack([rng.normal(0, 1, P), np.abs(rng.standard_cauchy(P)) + 0.1], 1))


def cfg2(N=10000, d=8, G=32, Np=64, seed=20260001):
    w = mvn_full(d, N, seed)
    w.update(name="cfg2", G=G, Np=Np)
    return w


def cfg3(N=100000, d=32, G=256, Np=256, seed=20260002):
    w = mvn_full(d, N, seed)
    w.update(name="cfg3", G=G, Np=Np)
    return w


def cfg4(S=10000, G=16, Np=32, seed=20260004):
    """hierarchical Binomial in the shape of Examples/Hierarchical_Example.jl:26-44,88-92: theta = (mu_b0, sd_b0, b0[1:S]),
    k_s ~ Binomial(50, logistic(mu_b0 + b0_s)), mu_b0 ~ N(1,1), sd_b0 ~ Cauchy+(0,1), b0_s ~ N(0, sd_b0); two blocks
    [hyper ; subject].  BASELINE: 128 groups over 8 GPUs -> G = 16 is one GPU's share."""
    rng = np.random.default_rng(seed)
    n = 50.0
    b0 = rng.normal(0, 1, S)
    k = rng.binomial(int(n), 1 / (1 + np.exp(-(1.0 + b0)))).astype(np.float64)
    D = S + 2
    m0 = np.zeros(D, np.uint8)
    m0[:2] = 1
    return dict(name="cfg4", G=G, Np=Np, fam=F.FAM_HIER_BINOMIAL, data=k, dims=[S], hyper=[n], D=D,
                pk=[F.PRIOR_NORMAL, F.PRIOR_HALFCAUCHY] + [F.PRIOR_NORMAL_REF] * S, pa=[1, 0] + [0] * S, pb=[1, 1] + [1] * S,
                pref=[0, 0] + [1] * S, lo=[-INF, 0] + [-INF] * S, hi=[INF] * D, masks=np.stack([m0, 1 - m0]), engine={},
                # the generating parameters and (roughly) the posterior's spread around them: a converged population (`--start posterior`)
                truth=np.concatenate([[1.0, 1.0], b0]), truth_sd=np.concatenate([[0.01, 0.01], np.full(S, 0.3)]),
                init=lambda P, rng: np.concatenate([rng.normal(1, 1, (P, 1)), np.abs(rng.standard_cauchy((P, 1))) + 0.3,
                                                    rng.normal(0, 1, (P, S))], 1))


def cfg5(N=50000, G=64, Np=128, seed=20260005):
    """LBA with 3 accumulators (theta = nu[3], A, k, tau; BASELINE's "6 params"), data SIMULATED from
    nu = (3,2,1), A = .8, k = .2, tau = .3, priors and bounds of Examples/Run_LBA.jl:10-31, snooker 0.1.
    BASELINE: 512 groups over 8 GPUs -> G = 64 is one GPU's share."""
    rng = np.random.default_rng(seed)
    na = 3
    choice, rt = simulate_lba(rng, N, (3.0, 2.0, 1.0), 0.8, 0.2, 0.3)
    mr = float(rt.min())
    D = na + 3
    return dict(name="cfg5", G=G, Np=Np, fam=F.FAM_LBA, data=np.concatenate([choice, rt]), dims=[N, na], hyper=None, D=D,
                # Run_LBA.jl:10-17: nu ~ N(1,5), A ~ N(.8,.2), k ~ N(.2,.1), tau ~ U(0, min_rt); bounds :31
                pk=[F.PRIOR_NORMAL] * na + [F.PRIOR_NORMAL, F.PRIOR_NORMAL, F.PRIOR_UNIFORM],
                pa=[1.0] * na + [0.8, 0.2, 0.0], pb=[5.0] * na + [0.2, 0.1, mr], pref=[0] * D,
                lo=[0.0] * D, hi=[INF] * (D - 1) + [mr], masks=None, engine=dict(theta_snooker=0.1),
                truth=np.array([3.0, 2.0, 1.0, 0.8, 0.2, 0.3]),
                # sample_prior (Run_LBA.jl:19-25), reflected into the bounds so that every start is finite
                init=lambda P, rng: np.concatenate([np.abs(rng.normal(1, 5, (P, na))) + 0.05, np.abs(rng.normal(0.8, 0.2, (P, 1))) + 0.05,
                                                    np.abs(rng.normal(0.2, 0.1, (P, 1))) + 0.02, rng.uniform(0.02, mr * 0.98, (P, 1))], 1))


def mvn30(N=100, d=30, G=256, Np=64, seed=20260006):
    """test/multivariate_normal_tests.jl:6-38,50-59: data rand(MvNormal(0, I), 100), theta = (mu[1:30], sigma), mu_j ~ N(0,1),
    sigma ~ Cauchy+(0,1), loglike = sum logpdf(MvNormal(mu, sigma^2 I), data), bounds ((-Inf,Inf),(0,Inf)); the reference runs
    it as DE-MC_Z with snooker (sample = resample, theta_snooker = 0.1, n_initial = (n_mu + 1) * 4) on Np = 3, n_groups = 1 --
    here with enough groups to fill the chip (the sampler's settings come from the caller: bench.py's row)"""
    rng = np.random.default_rng(seed)
    X = rng.normal(0, 1, (N, d))
    D = d + 1
    return dict(name="mvn30", G=G, Np=Np, fam=F.FAM_MVN_ISO, data=X, dims=[N, d], hyper=None, D=D,
                pk=[F.PRIOR_NORMAL] * d + [F.PRIOR_HALFCAUCHY], pa=[0.0] * D, pb=[1.0] * D, pref=[0] * D,
                lo=[-INF] * d + [0.0], hi=[INF] * D, masks=None, engine={},
                truth=np.concatenate([X.mean(0), [1.0]]),
                init=lambda P, rng: np.concatenate([rng.normal(0, 1, (P, d)), np.abs(rng.standard_cauchy((P, 1))) + 0.1], 1))


BUILDERS = dict(cfg1=cfg1, cfg2=cfg2, cfg3=cfg3, cfg4=cfg4, cfg5=cfg5, mvn30=mvn30)


def configure(engine, w):
    """model, priors, bounds and blocks of workload `w` on an engine (HipEngine or an object with the same methods)"""
    engine.set_model(w["fam"], w["data"], w["dims"], w["hyper"])
    engine.set_priors(w["pk"], w["pa"], w["pb"], w["pref"])
    engine.set_bounds(w["lo"], w["hi"])
    if w["masks"] is not None:
        engine.set_blocks(w["masks"])
