#!/usr/bin/env python3
"""Examples/Optimize_Example.jl / test/optimization_tests.jl: greedy DE (minimize! + evaluate_fun!) on rastrigin."""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import demc_amd as D  # noqa: E402

rng = np.random.default_rng(514)


def sample_prior():
    return [rng.uniform(-5, 5, 2)]


model = D.DEModel(sample_prior=sample_prior, loglike=D.RastriginObjective(), data=None, names=("x",))
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    de = D.DE(sample_prior=sample_prior, bounds=((-5.0, 5.0),), Np=12, n_groups=1, update_particle=D.minimize,
              evaluate_fitness=D.evaluate_fun)
particles = D.optimize(model, de, D.HIPBackend(schedule="synchronous"), 10000)
print(D.get_optimal(de, model, particles))
