"""Registered model family = what replaces the user closures of DEModel on the device.

The reference's `prior_loglike(theta...)` and `loglike(data, theta...)` are arbitrary Julia closures
(src/structs.jl:176-189); closures cannot run inside a HIP kernel, so the drop-in path accepts the
compositions of Distributions.logpdf that the reference's own tests and examples use, as data:
a per-parameter prior table and a likelihood family id.  Anything else raises DEMC_EUNSUPPORTED.
"""
import numpy as np

# likelihood family ids (include/demc.h)
FAM_GAUSSIAN, FAM_MVN_ISO, FAM_MVN_FULL, FAM_BINOMIAL, FAM_HIER_BINOMIAL, FAM_HIER_GAUSSIAN, FAM_LBA, FAM_LNR, \
    FAM_RASTRIGIN = range(9)
FAM_USER = 100
PRIOR_FLAT, PRIOR_NORMAL, PRIOR_HALFCAUCHY, PRIOR_UNIFORM, PRIOR_BETA, PRIOR_NORMAL_REF, PRIOR_GAMMA, \
    PRIOR_EXPONENTIAL, PRIOR_LOGNORMAL, PRIOR_CAUCHY = range(10)


# ---- priors: named like Distributions.jl -------------------------------------------------------
class Prior:
    kind = PRIOR_FLAT
    a = 0.0
    b = 1.0
    ref = None  # name of the parameter that supplies the scale (hierarchical)


class Flat(Prior):
    pass


class Normal(Prior):
    """Normal(mu, sigma); sigma may be the NAME of another scalar parameter (Normal(0, sigma_b0))."""

    def __init__(self, mu=0.0, sigma=1.0):
        self.a = float(mu)
        if isinstance(sigma, str):
            self.kind, self.ref, self.b = PRIOR_NORMAL_REF, sigma, 1.0
        else:
            self.kind, self.b = PRIOR_NORMAL, float(sigma)


class TruncatedCauchy(Prior):
    """truncated(Cauchy(loc, scale), 0, Inf) (Examples/Gaussian_Example.jl:14)."""
    kind = PRIOR_HALFCAUCHY

    def __init__(self, loc=0.0, scale=1.0):
        self.a, self.b = float(loc), float(scale)


class Uniform(Prior):
    kind = PRIOR_UNIFORM

    def __init__(self, a=0.0, b=1.0):
        self.a, self.b = float(a), float(b)


class Beta(Prior):
    kind = PRIOR_BETA

    def __init__(self, a=1.0, b=1.0):
        self.a, self.b = float(a), float(b)


class Gamma(Prior):
    """Gamma(shape, scale) (Distributions.jl parameterisation)"""
    kind = PRIOR_GAMMA

    def __init__(self, shape=1.0, scale=1.0):
        self.a, self.b = float(shape), float(scale)


class Exponential(Prior):
    """Exponential(scale)"""
    kind = PRIOR_EXPONENTIAL

    def __init__(self, scale=1.0):
        self.a, self.b = 0.0, float(scale)


class LogNormal(Prior):
    kind = PRIOR_LOGNORMAL

    def __init__(self, mu=0.0, sigma=1.0):
        self.a, self.b = float(mu), float(sigma)


class Cauchy(Prior):
    kind = PRIOR_CAUCHY

    def __init__(self, loc=0.0, scale=1.0):
        self.a, self.b = float(loc), float(scale)


class Priors:
    """prior_loglike as data: one Prior per top-level parameter (applied element-wise to array parameters),
    e.g. Priors(mu=Normal(0, 1), sigma=TruncatedCauchy(0, 1))."""

    def __init__(self, **by_name):
        self.by_name = by_name


# ---- likelihoods -------------------------------------------------------------------------------
class Likelihood:
    family = None

    def pack(self, data, shapes):
        """-> (data array, dims, hyper or None); shapes = list of np.shape of each top-level parameter."""
        raise NotImplementedError


class GaussianLikelihood(Likelihood):
    """sum(logpdf.(Normal(mu, sigma), data)) (Examples/Gaussian_Example.jl:26-28); theta = (mu, sigma)."""
    family = FAM_GAUSSIAN

    def pack(self, data, shapes):
        x = np.asarray(data, dtype=np.float64).ravel()
        return x, [x.size], None


class MvNormalIsoLikelihood(Likelihood):
    """sum(logpdf(MvNormal(mu, sigma^2 I), data)) with data d x N (test/multivariate_normal_tests.jl:31-33);
    theta = (mu[d], sigma)."""
    family = FAM_MVN_ISO

    def pack(self, data, shapes):
        x = np.asarray(data, dtype=np.float64)  # Julia layout: d x N (columns are observations)
        return np.ascontiguousarray(x.T), [x.shape[1], x.shape[0]], None


class MvNormalFullLikelihood(Likelihood):
    """sum(logpdf(MvNormal(mu, Sigma), data)), known full Sigma, data d x N; theta = mu[d] (BASELINE cfg2/cfg3)."""
    family = FAM_MVN_FULL

    def __init__(self, Sigma):
        self.Sigma = np.ascontiguousarray(Sigma, dtype=np.float64)

    def pack(self, data, shapes):
        x = np.asarray(data, dtype=np.float64)
        return np.ascontiguousarray(x.T), [x.shape[1], x.shape[0]], self.Sigma


class BinomialLikelihood(Likelihood):
    """logpdf(Binomial(data.N, theta), data.k) (test/binomial_tests.jl:15-17); data = (N=..., k=...) or arrays."""
    family = FAM_BINOMIAL

    def pack(self, data, shapes):
        n = np.atleast_1d(np.asarray(data["N"] if isinstance(data, dict) else data.N, dtype=np.float64))
        k = np.atleast_1d(np.asarray(data["k"] if isinstance(data, dict) else data.k, dtype=np.float64))
        return np.concatenate([n, k]), [n.size], None


class HierBinomialLikelihood(Likelihood):
    """k_s ~ Binomial(n, logistic(mu_b0 + b0_s)); theta = (mu_b0, sigma_b0, b0[S]) -- the shape of
    Examples/Hierarchical_Example.jl with a Binomial observation model (BASELINE cfg4)."""
    family = FAM_HIER_BINOMIAL

    def __init__(self, n):
        self.n = float(n)

    def pack(self, data, shapes):
        k = np.asarray(data, dtype=np.float64).ravel()
        return k, [k.size], [self.n]


class HierGaussianLikelihood(Likelihood):
    """Examples/Hierarchical_Example.jl:36-44; theta = (mu_b0, sigma_b0, b0[S], sigma); data = S vectors of n."""
    family = FAM_HIER_GAUSSIAN

    def pack(self, data, shapes):
        y = np.ascontiguousarray(np.asarray(data, dtype=np.float64))
        return y, [y.shape[0], y.shape[1]], None


class LBALikelihood(Likelihood):
    """sum(logpdf.(LBA(nu, A, k, tau), choice, rt)) (Examples/Run_LBA.jl:33-37); data = (choice, rt)."""
    family = FAM_LBA

    def pack(self, data, shapes):
        c = np.asarray(data[0], dtype=np.float64).ravel()
        rt = np.asarray(data[1], dtype=np.float64).ravel()
        return np.concatenate([c, rt]), [c.size, int(np.prod(shapes[0])) if shapes[0] else 1], None


class LNRLikelihood(Likelihood):
    """sum(logpdf(LNR(nu, sigma=1, tau), data)) (test/lognormal_race_tests.jl:9-12); data = (choice, rt)."""
    family = FAM_LNR

    def __init__(self, sigma=1.0):
        self.sigma = float(sigma)

    def pack(self, data, shapes):
        c = np.asarray(data[0], dtype=np.float64).ravel()
        rt = np.asarray(data[1], dtype=np.float64).ravel()
        return np.concatenate([c, rt]), [c.size, int(np.prod(shapes[0])) if shapes[0] else 1], [self.sigma]


class RastriginObjective(Likelihood):
    """objective of test/optimization_tests.jl:15-23 (optimize mode only)."""
    family = FAM_RASTRIGIN

    def pack(self, data, shapes):
        return None, [], None


class SourceLikelihood(Likelihood):
    """Plug-in for models outside the registered family.  The reference's `loglike(data, theta...)` closures are sums
    of per-observation log-densities (e.g. Examples/Gaussian_Example.jl:26-28); a Julia closure cannot run in a kernel,
    but the same term written as a HIP device function can:

        __device__ double demc_user_obs(const double* theta, int D, const double* data, long long N, long long i,
                                        const double* hyper, int nhyper);   // log-density of observation i

    `data` is flattened row-major with shape (N, ...) (observations first); the source is JIT-compiled for gfx950 by
    demc_set_model_source (include/demc.h)."""
    family = FAM_USER

    def __init__(self, source, hyper=None, row=False, has_prior=False):
        """row=True: the whole-row form (demc_set_model_source_row): `source` defines demc_user_loglike_row(theta, D, data, dims,
        ndims, hyper, nhyper, lane, n_lanes) -- and demc_user_prior_row(...) with has_prior=True -- evaluated by one workgroup
        per proposal; for likelihoods that are not a flat sum over observations and priors the per-scalar table cannot
        express (Examples/Hierarchical_Example.jl:26-44)."""
        self.source, self.hyper, self.row, self.has_prior = source, hyper, bool(row), bool(has_prior)

    def pack(self, data, shapes):
        x = np.ascontiguousarray(np.asarray(data, dtype=np.float64))
        return x, list(x.shape), self.hyper
