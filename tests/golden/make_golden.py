#!/usr/bin/env python3
"""Generates tests/golden/logpdf_golden.npz: inputs + expected log-densities computed with scipy.stats, an
implementation independent of both the oracle and the HIP kernels.  (The reference's own arithmetic lives in
Distributions.jl / SequentialSamplingModels.jl, which are neither vendored in /root/reference nor runnable here
-- no Julia -- so scipy's closed forms are the pin.  SURVEY.md 8c.)

Run:  python tests/golden/make_golden.py     (deterministic: fixed seeds)
"""
import os

import numpy as np
from scipy import integrate, special, stats

rng = np.random.default_rng(20261003)
out = {}

# ---- priors (per scalar) -------------------------------------------------------------------------------------
xs = np.array([-2.0, -0.3, 0.0, 0.2, 0.7, 1.0, 3.5])
out["prior_x"] = xs
out["prior_normal_1_2"] = stats.norm(1.0, 2.0).logpdf(xs)
with np.errstate(divide="ignore"):
    out["prior_halfcauchy_0_1"] = np.where(xs >= 0, stats.halfcauchy(0, 1).logpdf(np.abs(xs)), -np.inf)
    out["prior_halfcauchy_0_2p5"] = np.where(xs >= 0, stats.halfcauchy(0, 2.5).logpdf(np.abs(xs)), -np.inf)
    out["prior_uniform_m1_2"] = stats.uniform(-1, 3).logpdf(xs)
    out["prior_beta_2_3"] = stats.beta(2, 3).logpdf(xs)
    out["prior_beta_1_1"] = stats.beta(1, 1).logpdf(xs)
    out["prior_gamma_2p5_1p5"] = stats.gamma(2.5, scale=1.5).logpdf(xs)
    out["prior_exponential_0p7"] = stats.expon(scale=0.7).logpdf(xs)
    out["prior_lognormal_0p3_0p8"] = np.where(xs > 0, stats.lognorm(s=0.8, scale=np.exp(0.3)).logpdf(np.where(xs > 0, xs, 1.0)), -np.inf)
    out["prior_cauchy_m1_2"] = stats.cauchy(-1.0, 2.0).logpdf(xs)

# ---- Gaussian (Examples/Gaussian_Example.jl:26-28) ---------------------------------------------------------
x = rng.normal(0.2, 1.1, 50)
th = np.stack([rng.normal(0, 1, 8), rng.uniform(0.3, 2.5, 8)], 1)
out["gauss_x"], out["gauss_theta"] = x, th
out["gauss_ll"] = np.array([stats.norm(m, s).logpdf(x).sum() for m, s in th])

# ---- MvNormal(mu, sigma^2 I) (test/multivariate_normal_tests.jl:31-33) --------------------------------------
N, d = 40, 5
X = rng.normal(0, 1, (N, d)) + rng.normal(0, 1, d)
th = np.concatenate([rng.normal(0, 1, (6, d)), rng.uniform(0.4, 2.0, (6, 1))], 1)
out["iso_X"], out["iso_theta"] = X, th
out["iso_ll"] = np.array([stats.multivariate_normal(t[:d], t[d] ** 2 * np.eye(d)).logpdf(X).sum() for t in th])

# ---- MvNormal(mu, Sigma) full ------------------------------------------------------------------------------
N, d = 60, 6
A = rng.normal(0, 1, (d, d))
Sigma = A @ A.T / d + 0.5 * np.eye(d)
X = rng.multivariate_normal(rng.normal(0, 1, d), Sigma, N)
th = rng.normal(0, 1, (6, d))
out["full_X"], out["full_Sigma"], out["full_theta"] = X, Sigma, th
out["full_ll"] = np.array([stats.multivariate_normal(t, Sigma).logpdf(X).sum() for t in th])

# ---- Binomial (test/binomial_tests.jl:15-17) ----------------------------------------------------------------
n = rng.integers(5, 30, 7).astype(float)
k = np.floor(n * rng.uniform(0.1, 0.9, 7))
k[0], k[1] = 0.0, n[1]  # edge cases: k = 0 and k = n
p = np.array([0.05, 0.3, 0.5, 0.77, 0.99])
out["binom_n"], out["binom_k"], out["binom_p"] = n, k, p
out["binom_ll"] = np.array([stats.binom(n.astype(int), q).logpmf(k.astype(int)).sum() for q in p])

# ---- hierarchical Binomial / Gaussian ----------------------------------------------------------------------
S, ntr = 9, 50
kk = rng.integers(0, ntr + 1, S).astype(float)
th = np.concatenate([rng.normal(1, 1, (5, 1)), rng.uniform(0.3, 2, (5, 1)), rng.normal(0, 1, (5, S))], 1)
out["hb_k"], out["hb_n"], out["hb_theta"] = kk, np.array([float(ntr)]), th
out["hb_ll"] = np.array([stats.binom(ntr, special.expit(t[0] + t[2:])).logpmf(kk.astype(int)).sum() for t in th])
out["hb_prior"] = np.array([stats.norm(1, 1).logpdf(t[0]) + stats.halfcauchy(0, 1).logpdf(t[1]) +
                            stats.norm(0, t[1]).logpdf(t[2:]).sum() for t in th])
S, nd = 7, 6
Y = rng.normal(1, 1, (S, nd))
th = np.concatenate([rng.normal(1, 1, (5, 1)), rng.uniform(0.3, 2, (5, 1)), rng.normal(0, 1, (5, S)),
                     rng.uniform(0.3, 2, (5, 1))], 1)
out["hg_Y"], out["hg_theta"] = Y, th
out["hg_ll"] = np.array([stats.norm(0, t[2 + S]).logpdf(Y - (t[0] + t[2:2 + S])[:, None]).sum() for t in th])

# ---- LNR (test/lognormal_race_tests.jl:9-12): winner density x product of the others' survival --------------
na, N = 4, 25
choice = rng.integers(1, na + 1, N)
rt = rng.uniform(0.5, 2.0, N)
th = np.concatenate([rng.normal(-1, 1, (5, na)), rng.uniform(0.05, 0.45, (5, 1))], 1)


def lnr_ll(t):
    nu, tau = t[:na], t[na]
    ll = 0.0
    for c, r in zip(choice, rt):
        for i in range(na):
            dist = stats.lognorm(s=1.0, scale=np.exp(nu[i]))
            ll += dist.logpdf(r - tau) if i + 1 == c else dist.logsf(r - tau)
    return ll


out["lnr_choice"], out["lnr_rt"], out["lnr_theta"] = choice.astype(float), rt, th
out["lnr_ll"] = np.array([lnr_ll(t) for t in th])

# ---- LBA (Examples/Run_LBA.jl:33-37): Brown & Heathcote (2008) with b = A + k, s = 1, conditioned on at least
# one positive drift, density floored at 1e-10 (SequentialSamplingModels.jl convention, recalled) --------------
na, N = 3, 25
choice = rng.integers(1, na + 1, N)
rt = rng.uniform(0.45, 1.8, N)
th = np.concatenate([rng.uniform(0.5, 4, (5, na)), rng.uniform(0.5, 1.1, (5, 1)), rng.uniform(0.05, 0.4, (5, 1)),
                     rng.uniform(0.05, 0.4, (5, 1))], 1)


def lba_pdf(v, b, A, t):
    n1, n2 = (b - A - t * v) / t, (b - t * v) / t
    return (1 / A) * (-v * stats.norm.cdf(n1) + stats.norm.pdf(n1) + v * stats.norm.cdf(n2) - stats.norm.pdf(n2))


def lba_cdf(v, b, A, t):
    n1, n2 = (b - A - t * v) / t, (b - t * v) / t
    return (1 + ((b - A - t * v) / A) * stats.norm.cdf(n1) - ((b - t * v) / A) * stats.norm.cdf(n2)
            + (t / A) * stats.norm.pdf(n1) - (t / A) * stats.norm.pdf(n2))


def lba_trial(nu, A, k, tau, c, r):
    if r < tau:
        return -np.inf
    b, t = A + k, r - tau
    den = 1.0
    for i in range(len(nu)):
        den *= lba_pdf(nu[i], b, A, t) if i + 1 == c else 1 - lba_cdf(nu[i], b, A, t)
    den /= 1 - np.prod(stats.norm.cdf(-np.asarray(nu)))
    return np.log(max(den, 1e-10))


out["lba_choice"], out["lba_rt"], out["lba_theta"] = choice.astype(float), rt, th
out["lba_ll"] = np.array([sum(lba_trial(t[:na], t[na], t[na + 1], t[na + 2], c, r) for c, r in zip(choice, rt)) for t in th])
# defective densities integrate to 1 over (choice, t): the restated LBA is a proper distribution
nu, A, k, tau = np.array([3.0, 2.0, 1.0]), 0.8, 0.2, 0.3
tot = sum(integrate.quad(lambda r: np.exp(lba_trial(nu, A, k, tau, c, r)), tau, 60, limit=400)[0] for c in (1, 2, 3))
out["lba_total_mass"] = np.array([tot])

np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "logpdf_golden.npz"), **out)
print({k: np.shape(v) for k, v in out.items()})
print("LBA total mass", tot)
