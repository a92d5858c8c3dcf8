#!/usr/bin/env python3
"""Examples/Run_LBA.jl: Linear Ballistic Accumulator, nu[2], A, k, tau."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import demc_amd as D  # noqa: E402

rng = np.random.default_rng(88484)


def rand_lba(nu, A, k, tau, n):
    """simulate LBA trials (drift ~ N(nu, 1) redrawn until one is positive, start ~ U(0, A), threshold A + k)"""
    choice, rt = np.empty(n), np.empty(n)
    for i in range(n):
        while True:
            v = rng.normal(nu, 1.0)
            if (v > 0).any():
                break
        t = (A + k - rng.uniform(0, A, len(nu))) / np.where(v > 0, v, np.nan)
        choice[i], rt[i] = np.nanargmin(t) + 1, np.nanmin(t) + tau
    return choice, rt


choice, rt = rand_lba(np.array([3.0, 2.0]), 0.8, 0.2, 0.3, 100)
min_rt = rt.min()


def sample_prior():
    return D.as_union([rng.normal(1, 5, 2), rng.normal(0.8, 0.2), rng.normal(0.2, 0.1), rng.uniform(0, min_rt)])


model = D.DEModel(sample_prior=sample_prior, names=("ν", "A", "k", "τ"), data=(choice, rt),
                  prior_loglike=D.Priors(ν=D.Normal(1, 5), A=D.Normal(0.8, 0.2), k=D.Normal(0.2, 0.1), τ=D.Uniform(0, min_rt)),
                  loglike=D.LBALikelihood())                                               # Run_LBA.jl:10-37
de = D.DE(sample_prior=sample_prior, bounds=((0.0, np.inf), (0.0, np.inf), (0.0, np.inf), (0.0, min_rt)), burnin=1500,
          n_groups=3, Np=15)
chain = D.sample(model, de, D.MCMCThreads(), 3000, progress=True)
for name, s in chain.describe().items():
    print(f"{name}: mean {s['mean']:.3f}  std {s['std']:.3f}  rhat {s['rhat']:.3f}")
