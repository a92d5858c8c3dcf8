#!/usr/bin/env python3
"""Examples/Gaussian_Example.jl on the MI355X path: Normal(mu, sigma) data, mu ~ N(0,1), sigma ~ Cauchy+(0,1)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import demc_amd as D  # noqa: E402

rng = np.random.default_rng(50514)
data = rng.normal(0.0, 1.0, 50)


def sample_prior():
    return [rng.normal(0, 1), abs(rng.standard_cauchy())]


model = D.DEModel(sample_prior=sample_prior, names=("μ", "σ"), data=data,
                  prior_loglike=D.Priors(μ=D.Normal(0, 1), σ=D.TruncatedCauchy(0, 1)),   # Gaussian_Example.jl:11-16
                  loglike=D.GaussianLikelihood())                                          # Gaussian_Example.jl:26-28
de = D.DE(sample_prior=sample_prior, bounds=((-np.inf, np.inf), (0.0, np.inf)), burnin=1000, Np=6)
chains = D.sample(model, de, D.MCMCThreads(), 2000, progress=True)
for name, s in chains.describe().items():
    print(f"{name}: mean {s['mean']:.3f}  std {s['std']:.3f}  rhat {s['rhat']:.3f}")
