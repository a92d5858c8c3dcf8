#!/usr/bin/env python3
"""Examples/Multivariate_Guassian_Example.jl: 30 means + one sigma, nested Theta [mu, sigma], history partners
(sample = resample, DE-MC_Z) with snooker updates, one group of three particles."""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import demc_amd as D  # noqa: E402

rng = np.random.default_rng(50514)
n_mu, n_d = 30, 100
mus = rng.normal(0.0, 1.0, n_mu)
data = mus[:, None] + rng.normal(0, 1, (n_mu, n_d))  # variables x observations, like rand(MvNormal(mus, I), n_d)


def sample_prior():
    return D.as_union([rng.normal(0, 1, n_mu), abs(rng.standard_cauchy())])


model = D.DEModel(sample_prior=sample_prior, names=("μ", "σ"), data=data,
                  prior_loglike=D.Priors(μ=D.Normal(0, 1), σ=D.TruncatedCauchy(0, 1)),
                  loglike=D.MvNormalIsoLikelihood())        # sum(logpdf(MvNormal(mus, sigma^2 I), data))
with warnings.catch_warnings():
    warnings.simplefilter("ignore")                         # alpha forced to 0 for a single group (structs.jl:102-105)
    de = D.DE(sample_prior=sample_prior, bounds=((-np.inf, np.inf), (0.0, np.inf)), sample=D.resample, burnin=5000,
              n_initial=(n_mu + 1) * 4, Np=3, n_groups=1, θsnooker=0.1)
chains = D.sample(model, de, D.MCMCThreads(), 20000, progress=True)
d = chains.describe()
means = np.array([d[f"μ[{i + 1}]"]["mean"] for i in range(n_mu)])
print("cor(true means, posterior means) =", np.corrcoef(mus, means)[0, 1])
print("posterior sd of the means ~ 0.1:", np.round([d[f"μ[{i + 1}]"]["std"] for i in range(5)], 3), "…  σ:", d["σ"])
