"""Host-side mirror of src/structs.jl: Particle, DE, DEModel -- same names, keywords, defaults and error
behaviour, so that scripts written against the reference read the same here.  The function-valued hooks of DE
(structs.jl:71-77) are the reference's own function objects re-exported as tags; they are translated to the
enums of include/demc.h, and anything else is rejected with DEMC_EUNSUPPORTED (no CPU fallback)."""
import warnings

import numpy as np

from . import _ffi
from .families import Likelihood, Priors


class _Hook:
    def __init__(self, name, code):
        self.name, self.code = name, code

    def __repr__(self):
        return self.name


# de.generate_proposal (crossover.jl:154-226)
random_gamma = _Hook("random_gamma", 0)
fixed_gamma = _Hook("fixed_gamma", 1)
variable_gamma = _Hook("variable_gamma", 2)
# de.sample (crossover.jl:113-140); `sample` itself is the driver, so the current-population sampler is sample_current
sample_current = _Hook("sample", 0)
resample = _Hook("resample", 1)
# de.update_particle! (utilities.jl:201-226)
mh_update = _Hook("mh_update!", 0)
maximize = _Hook("maximize!", 1)
minimize = _Hook("minimize!", 2)
# de.evaluate_fitness! (utilities.jl:92-120)
compute_posterior = _Hook("compute_posterior!", 0)
evaluate_fun = _Hook("evaluate_fun!", 1)

SCHEDULES = {"synchronous": 1, "two_colour": 2}
LOGLIKE_MODES = {"streaming": 0, "suffstat": 1, "direct": 2}


class MCMCThreads:
    """AbstractMCMC.MCMCThreads tag (main.jl:62-71).  Accepted for source compatibility: on this backend the
    groups are always updated in parallel (that is what the kernels are), so it selects the same path."""


class HIPBackend:
    """Explicit tag for the MI355X path: sample(model, de, HIPBackend(), n_iter)."""

    def __init__(self, schedule="two_colour", loglike_mode="streaming", device_id=0, seed=None):
        if schedule not in SCHEDULES:
            raise _ffi.DemcError(_ffi.EUNSUPPORTED, f"schedule {schedule!r}: use 'synchronous' or 'two_colour' "
                                 "(the sequential in-place sweep is the CPU reference schedule)")
        if loglike_mode not in LOGLIKE_MODES:
            raise _ffi.DemcError(_ffi.EINVAL, f"loglike_mode {loglike_mode!r}")
        self.schedule, self.loglike_mode, self.device_id, self.seed = schedule, loglike_mode, device_id, seed


class Particle:
    """structs.jl:202-223.  Theta may hold scalars and arrays (nested parameters)."""

    def __init__(self, Θ=None, accept=None, weight=0.0, lp=None, id=0, Theta=None):
        th = Θ if Θ is not None else (Theta if Theta is not None else [0.0])
        if np.isscalar(th):
            th = [th]
        self.Θ = list(th)
        self.accept = [] if accept is None else accept
        self.weight = weight
        self.lp = [] if lp is None else lp
        self.id = id

    Theta = property(lambda self: self.Θ)

    # Particle algebra (utilities.jl:269-357): element-wise over the top-level entries
    def _bin(self, other, op):
        if isinstance(other, Particle):
            return Particle(Θ=[op(np.asarray(a, dtype=float), np.asarray(b, dtype=float)) for a, b in zip(self.Θ, other.Θ)])
        return Particle(Θ=[op(np.asarray(a, dtype=float), other) for a in self.Θ])

    def __add__(self, o):
        return self._bin(o, lambda a, b: a + b)

    __radd__ = __add__

    def __sub__(self, o):
        return self._bin(o, lambda a, b: a - b)

    def __mul__(self, o):
        return self._bin(o, lambda a, b: a * b)

    __rmul__ = __mul__

    def flat(self):
        return np.concatenate([np.asarray(t, dtype=np.float64).ravel() for t in self.Θ])


def project(p1, p2):
    """utilities.jl:239-246: p2 * (<p1,p2>/<p2,p2>), dots over all nested scalars."""
    a, b = p1.flat(), p2.flat()
    return p2 * (float(a @ b) / float(b @ b))


class DE:
    """DE(; n_groups=4, Np, burnin=1000, discard_burnin=true, α=.1, β=.1, ϵ=.001, σ=.05, κ=1.0, θsnooker=0.0,
    bounds, n_initial=0, generate_proposal=random_gamma, update_particle! = mh_update!, evaluate_fitness! =
    compute_posterior!, sample=sample, blocking_on = x->false, blocks=[false], sample_prior)  (structs.jl:80-131).
    ASCII aliases: alpha, beta, epsilon, sigma, kappa, theta_snooker, update_particle, evaluate_fitness."""

    def __init__(self, *, Np, bounds, sample_prior, n_groups=4, priors=None, burnin=1000, discard_burnin=True,
                 α=None, β=None, ϵ=None, σ=None, κ=None, θsnooker=None, alpha=0.1, beta=0.1, epsilon=0.001,
                 sigma=0.05, kappa=1.0, theta_snooker=0.0, n_initial=0, generate_proposal=random_gamma,
                 update_particle=mh_update, evaluate_fitness=compute_posterior, sample=sample_current,
                 blocking_on=None, blocks=None):
        self.n_groups, self.Np = int(n_groups), int(Np)
        self.burnin, self.discard_burnin = int(burnin), bool(discard_burnin)
        self.α = float(alpha if α is None else α)
        self.β = float(beta if β is None else β)
        self.ϵ = float(epsilon if ϵ is None else ϵ)
        self.σ = float(sigma if σ is None else σ)
        self.κ = float(kappa if κ is None else κ)
        self.θsnooker = float(theta_snooker if θsnooker is None else θsnooker)
        if self.n_groups == 1 and self.α > 0:  # structs.jl:102-105
            self.α = 0.0
            warnings.warn("migration probability α > 0 but n_groups == 1. Changing α = 0.0")
        self.bounds = bounds
        self.n_initial = int(n_initial)
        self.iter = 1
        self.generate_proposal = generate_proposal
        self.update_particle = update_particle
        self.evaluate_fitness = evaluate_fitness
        self.sample = sample
        self.blocking_on = blocking_on if blocking_on is not None else (lambda de: False)
        self.blocks = blocks if blocks is not None else [False]
        self.sample_prior = sample_prior
        self.samples = np.empty((0, 0, 0))
        for hook, allowed, what in ((generate_proposal, (random_gamma, fixed_gamma, variable_gamma), "generate_proposal"),
                                    (update_particle, (mh_update, maximize, minimize), "update_particle!"),
                                    (evaluate_fitness, (compute_posterior, evaluate_fun), "evaluate_fitness!"),
                                    (sample, (sample_current, resample), "sample")):
            if hook not in allowed:
                raise _ffi.DemcError(_ffi.EUNSUPPORTED,
                                     f"{what}={hook!r}: custom hooks cannot run on the device; registered: {allowed}")

    alpha = property(lambda s: s.α)
    beta = property(lambda s: s.β)
    epsilon = property(lambda s: s.ϵ)
    sigma = property(lambda s: s.σ)
    kappa = property(lambda s: s.κ)
    theta_snooker = property(lambda s: s.θsnooker)


class DEModel:
    """DEModel(args...; prior_loglike=nothing, loglike, names, sample_prior, data, kwargs...) (structs.jl:176-189).
    `loglike` is a registered Likelihood, `prior_loglike` a Priors table (families.py)."""

    def __init__(self, *args, prior_loglike=None, loglike, names, sample_prior, data=None, **kwargs):
        if not isinstance(loglike, Likelihood):
            raise _ffi.DemcError(_ffi.EUNSUPPORTED,
                                 "loglike must be a registered Likelihood (families.py): arbitrary closures cannot run "
                                 "on the device and there is no CPU fallback")
        if prior_loglike is not None and not isinstance(prior_loglike, Priors):
            raise _ffi.DemcError(_ffi.EUNSUPPORTED, "prior_loglike must be a Priors table (families.py)")
        self.prior_loglike = prior_loglike
        self.loglike = loglike
        self.sample_prior = sample_prior
        self.names = tuple(str(n) for n in names)
        self.data = data
        self.args, self.kwargs = args, kwargs


def as_union(p):
    """utilities.jl:182-187: in Julia this narrows the element type of a nested parameter vector; a Python list
    already holds mixed scalars/arrays, so it is the identity."""
    return list(p)
