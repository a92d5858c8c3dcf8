"""MI355X-native DE-MCMC hot path behind the API of itsdfish/DifferentialEvolutionMCMC.jl.

Exports follow src/DifferentialEvolutionMCMC.jl:15-18.  Compute happens only in libdemc_hip.so (hand-written
gfx950 kernels behind the C-ABI of include/demc.h); importing this package without the built library works, but
any sampling call raises -- there is no CPU fallback.
"""
from . import _ffi, families
from ._ffi import DemcError, HipEngine, MultiEngine
from .chains import Chains
from .families import (Beta, BinomialLikelihood, Cauchy, Exponential, Flat, Gamma, GaussianLikelihood, LogNormal, HierBinomialLikelihood,
                       HierGaussianLikelihood, LBALikelihood, LNRLikelihood, MvNormalFullLikelihood,
                       MvNormalIsoLikelihood, Normal, Priors, RastriginObjective, SourceLikelihood, TruncatedCauchy,
                       Uniform)
from .sampler import get_optimal, optimize, sample
from .structs import (DE, DEModel, HIPBackend, MCMCThreads, Particle, as_union, compute_posterior, evaluate_fun,
                      fixed_gamma, maximize, mh_update, minimize, project, random_gamma, resample, sample_current,
                      variable_gamma)

DEMCMC = __name__

__all__ = ["DE", "Particle", "DEModel", "sample", "MCMCThreads", "HIPBackend", "fixed_gamma", "variable_gamma",
           "random_gamma", "evaluate_fun", "compute_posterior", "optimize", "get_optimal", "resample", "as_union",
           "mh_update", "maximize", "minimize", "Chains", "Priors", "HipEngine", "MultiEngine", "DemcError"]
