#!/usr/bin/env python3
"""Examples/Hierarchical_Example.jl: hierarchical Gaussian with block updates [hyper-parameters ; subject effects]."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import demc_amd as D  # noqa: E402

rng = np.random.default_rng(9528)
n_data, n_subj = 50, 50
beta0 = rng.normal(0.0, 1.0, n_subj)
data = 1.0 + beta0[:, None] + rng.normal(0, 0.5, (n_subj, n_data))


def sample_prior():
    sd_b0 = abs(rng.standard_cauchy())
    return D.as_union([rng.normal(1, 1), sd_b0, rng.normal(0.0, sd_b0, n_subj), abs(rng.standard_cauchy())])


model = D.DEModel(sample_prior=sample_prior, names=("μβ0", "σβ0", "β0", "σ"), data=data,
                  prior_loglike=D.Priors(μβ0=D.Normal(1, 1), σβ0=D.TruncatedCauchy(0, 1), β0=D.Normal(0, "σβ0"),
                                         σ=D.TruncatedCauchy(0, 1)),                      # Hierarchical_Example.jl:26-33
                  loglike=D.HierGaussianLikelihood())                                     # Hierarchical_Example.jl:36-44
blocks = [[True, True, np.zeros(n_subj, bool), True], [False, False, np.ones(n_subj, bool), False]]   # :88-92
de = D.DE(sample_prior=sample_prior, bounds=((-np.inf, np.inf), (0.0, np.inf), (-np.inf, np.inf), (0.0, np.inf)),
          sample=D.resample, burnin=5000, n_initial=(n_subj + 1) * 4, Np=6, n_groups=2, θsnooker=0.1,
          blocking_on=lambda de: True, blocks=blocks)
chains = D.sample(model, de, D.MCMCThreads(), 10000, progress=True)
d = chains.describe()
print({k: round(d[k]["mean"], 3) for k in ("μβ0", "σβ0", "σ")}, "(generating values 1.0, 1.0, 0.5)")
