"""Host driver mirroring src/main.jl (sample/_sample/sample_init/bundle_samples) and src/optimize.jl on top of the
C-ABI engine.  The per-iteration work -- step!/pstep! (main.jl:84-107) -- happens inside libdemc_hip.so."""
import numpy as np

from . import _ffi
from .chains import Chains
from .families import PRIOR_NORMAL_REF, Priors, SourceLikelihood
from .structs import (DE, HIPBackend, LOGLIKE_MODES, MCMCThreads, SCHEDULES, DEModel, Particle, maximize)


def _shapes(theta):
    return [np.shape(t) for t in theta]


def _flatten(theta):
    return np.concatenate([np.asarray(t, dtype=np.float64).ravel() for t in theta])


def get_names(model, shapes):
    """utilities.jl:131-149 (Julia's CartesianIndices order: first index fastest, 1-based)."""
    names = []
    for k, shp in zip(model.names, shapes):
        if len(shp) == 0:
            names.append(str(k))
        else:
            for lin in range(int(np.prod(shp))):
                idx = np.unravel_index(lin, shp, order="F")
                names.append(f"{k}[{','.join(str(i + 1) for i in idx)}]")
    return names + ["acceptance", "lp"]


def _flat_layout(model, de, theta0):
    """Flatten nested Theta to D scalars: bounds per top-level parameter, zip-truncated
    (utilities.jl:73-78); blocks as nested Bool arrays (structs.jl:45) -> byte masks."""
    shapes = _shapes(theta0)
    sizes = [int(np.prod(s)) if len(s) else 1 for s in shapes]
    D = int(sum(sizes))
    offs = np.concatenate([[0], np.cumsum(sizes)])
    lo = np.full(D, -np.inf)
    hi = np.full(D, np.inf)
    for i, b in enumerate(de.bounds):
        if i >= len(sizes):
            break
        lo[offs[i]:offs[i + 1]] = float(b[0])
        hi[offs[i]:offs[i + 1]] = float(b[1])
    kind = np.zeros(D, np.int32)
    a = np.zeros(D)
    b_ = np.ones(D)
    ref = np.zeros(D, np.int32)
    pri = model.prior_loglike
    if isinstance(pri, Priors):
        for i, nm in enumerate(model.names):
            p = pri.by_name.get(nm)
            if p is None:
                continue
            sl = slice(offs[i], offs[i + 1])
            kind[sl], a[sl], b_[sl] = p.kind, p.a, p.b
            if p.kind == PRIOR_NORMAL_REF:
                j = model.names.index(p.ref)
                if sizes[j] != 1:
                    raise _ffi.DemcError(_ffi.EINVAL, "hierarchical scale must be a scalar parameter")
                ref[sl] = offs[j]
    # de.blocks -> byte masks.  Whether blocking is ON is asked per iteration (de.blocking_on(de), main.jl:137,162),
    # see _blocking_schedule(); the masks are built whenever de.blocks holds real blocks.
    masks = None
    if _has_blocks(de.blocks):
        rows = []
        for blk in de.blocks:
            m = np.zeros(D, np.uint8)
            for i, e in enumerate(blk):
                e = np.asarray(e, dtype=bool)
                m[offs[i]:offs[i + 1]] = e.ravel() if e.ndim else e
            rows.append(m)
        masks = np.stack(rows)
    return dict(shapes=shapes, sizes=sizes, offs=offs, D=D, lo=lo, hi=hi, kind=kind, a=a, b=b_, ref=ref, masks=masks)


def _has_blocks(blocks):
    """DE's default is blocks = [false] (structs.jl:99): a placeholder, not a block list"""
    try:
        return len(blocks) > 0 and not isinstance(blocks[0], (bool, np.bool_))
    except TypeError:
        return False


def _blocking_schedule(de, n_iter):
    """de.blocking_on(de) is a user hook evaluated on every iteration with de.iter set (main.jl:34,137,162).  It runs on
    the host; consecutive iterations with the same answer become one engine call.  -> list of (first_iter, count, on)"""
    saved = de.iter
    flags = []
    for it in range(1, n_iter + 1):
        de.iter = it + de.n_initial
        flags.append(bool(de.blocking_on(de)))
    de.iter = saved
    runs = []
    for i, f in enumerate(flags):
        if runs and runs[-1][2] == f:
            runs[-1][1] += 1
        else:
            runs.append([i + 1, 1, f])
    return runs


def engine_config(de, lay, n_iter, backend, n_groups_local=None, group_offset=0, store_history=True, n_initial=None):
    n_initial = de.n_initial if n_initial is None else n_initial
    seed = backend.seed if backend.seed is not None else int(np.random.randint(0, 2**62))
    return dict(n_groups=de.n_groups if n_groups_local is None else n_groups_local, Np=de.Np, D=lay["D"], n_blocks=0,
                burnin=de.burnin, n_initial=n_initial, n_rows=n_iter + n_initial, alpha=de.α, beta=de.β, eps=de.ϵ,
                sigma=de.σ, kappa=de.κ, theta_snooker=de.θsnooker, proposal_kind=de.generate_proposal.code,
                partner_kind=de.sample.code, update_kind=de.update_particle.code,
                fitness_kind=de.evaluate_fitness.code, schedule=SCHEDULES[backend.schedule],
                store_history=1 if store_history else 0, group_offset=group_offset, n_groups_total=de.n_groups,
                seed=seed, device_id=backend.device_id, loglike_mode=LOGLIKE_MODES[backend.loglike_mode])


def configure_engine(eng, model, lay):
    data, dims, hyper = model.loglike.pack(model.data, lay["shapes"])
    if isinstance(model.loglike, SourceLikelihood) and model.loglike.row:
        eng.set_model_source_row(model.loglike.source, data, dims, hyper, has_prior=model.loglike.has_prior)
    elif isinstance(model.loglike, SourceLikelihood):
        eng.set_model_source(model.loglike.source, data, dims, hyper)
    else:
        eng.set_model(model.loglike.family, data, dims, hyper)
    eng.set_priors(lay["kind"], lay["a"], lay["b"], lay["ref"])
    eng.set_bounds(lay["lo"], lay["hi"])


def sample_init(model, de, n_iter, lay, P):
    """main.jl:263-271 + utilities.jl:13-41: history rows 1:n_initial are independent prior draws per particle;
    Theta starts at samples[1,:,id] when n_initial > 0, else at a fresh prior draw.  Ids run 1..P group-major
    in the reference; here 0..P-1."""
    D = lay["D"]
    init_rows = np.empty((de.n_initial, P, D))
    for p in range(P):
        for i in range(de.n_initial):
            init_rows[i, p] = _flatten(model.sample_prior())
    if de.n_initial > 0:
        theta = init_rows[0].copy()
    else:
        theta = np.stack([_flatten(model.sample_prior()) for _ in range(P)])
    return theta, init_rows


def rekey_by_id(th, acc, lp, idh, id0=0, P_total=None):
    """History comes back keyed by slot plus the id that occupied the slot; the reference keys by particle id
    (samples[iter, :, p.id], utilities.jl:170-180; accept/lp live on the Particle object)."""
    n, P, D = th.shape
    P_total = P if P_total is None else P_total
    oth = np.zeros((n, P_total, D))
    oacc = np.zeros((n, P_total), np.uint8)
    olp = np.zeros((n, P_total))
    rows = np.arange(n)[:, None]
    cols = idh - id0
    oth[rows, cols] = th
    oacc[rows, cols] = acc
    olp[rows, cols] = lp
    return oth, oacc, olp


def bundle_samples(model, de, lay, full, n_iter):
    """main.jl:222-250, including its row selection: rows offset+1 .. offset+Ns of the history with
    offset = burnin (or 0), which ignores the n_initial offset.  Deviation (q2): accept/lp are
    paired with Theta by particle id rather than by final slot.  `full` is [rows][D+2][P] by particle id."""
    Ns = n_iter - de.burnin if de.discard_burnin else n_iter
    offset = de.burnin if de.discard_burnin else 0
    names = get_names(model, lay["shapes"])
    return Chains(full[offset:offset + Ns], names, parameters=names[:-2])


def _parse(args):
    backend = None
    rest = []
    for a in args:
        if isinstance(a, (MCMCThreads, HIPBackend)):
            backend = a
        else:
            rest.append(a)
    if len(rest) != 1:
        raise TypeError("sample(model, de, [MCMCThreads()|HIPBackend()], n_iter)")
    if not isinstance(backend, HIPBackend):
        backend = HIPBackend()
    return backend, int(rest[0])


def _run(model, de, n_iter, backend, progress, engine_factory):
    if not isinstance(model, DEModel) or not isinstance(de, DE):
        raise TypeError("sample(model::DEModel, de::DE, ...)")
    theta0 = model.sample_prior()
    lay = _flat_layout(model, de, theta0)
    P = de.n_groups * de.Np
    cfg = engine_config(de, lay, n_iter, backend)
    eng = (engine_factory or _ffi.HipEngine)(**cfg)
    try:
        configure_engine(eng, model, lay)
        theta, init_rows = sample_init(model, de, n_iter, lay, P)
        if de.n_initial > 0:
            eng.set_history_rows(0, init_rows)
        eng.set_state(theta)  # weights: evaluate_fitness! on device (utilities.jl:19)
        # for iter in 1:n_iter: de.iter = iter + n_initial; groups = stepfun(model, de, groups)  (main.jl:33-38)
        no_blocks = np.zeros((0, lay["D"]), np.uint8)
        done = 0
        for first, count, on in _blocking_schedule(de, n_iter):
            if on and lay["masks"] is None:
                raise _ffi.DemcError(_ffi.EINVAL, "blocking_on(de) is true but de.blocks holds no blocks")
            eng.set_blocks(lay["masks"] if on else no_blocks)
            chunk = max(1, n_iter // 20) if progress else count
            it = first
            while it < first + count:
                n = min(chunk, first + count - it)
                eng.step(it + de.n_initial, n)
                it += n
                done += n
                if progress:
                    print(f"\rDE-MCMC {done}/{n_iter}", end="", flush=True)
        if progress:
            print()
        de.iter = n_iter + de.n_initial
        n_rows = n_iter + de.n_initial
        if hasattr(eng, "export_chains"):  # device-side re-key + layout (demc_export_chains)
            full = eng.export_chains(0, n_rows)  # [n_rows][D+2][P] by particle id
        else:  # engines without it (the CPU oracle injected by tests): re-key on the host
            th, acc, lp = rekey_by_id(*eng.get_history(0, n_rows))
            full = np.concatenate([np.transpose(th, (0, 2, 1)), acc[:, None, :].astype(np.float64), lp[:, None, :]], axis=1)
        state = eng.get_state()
    finally:
        eng.close()
    return lay, full, state


def sample(model, de, *args, progress=False, engine_factory=None, **kwargs):
    """sample(model, de, n_iter) / sample(model, de, MCMCThreads(), n_iter) (main.jl:19-20, 62-71) and
    sample(model, de, HIPBackend(...), n_iter).  Every form runs on the MI355X; there is no CPU path."""
    backend, n_iter = _parse(args)
    lay, full, _ = _run(model, de, n_iter, backend, progress, engine_factory)
    de.samples = full[:, :lay["D"], :]  # the reference's de.samples: (rows, parameters, particle id)
    return bundle_samples(model, de, lay, full, n_iter)


def optimize(model, de, *args, progress=False, engine_factory=None, **kwargs):
    """optimize(model, de, n_iter) (optimize.jl:17-38): same loop, returns the particles.  n_initial is not
    added to de.iter there (optimize.jl:32, quirk q7); history partners are therefore not supported in this mode."""
    backend, n_iter = _parse(args)
    if de.n_initial != 0:
        raise _ffi.DemcError(_ffi.EUNSUPPORTED, "optimize with n_initial > 0")
    lay, _, (theta, weight, ids) = _run(model, de, n_iter, backend, progress, engine_factory)
    out = []
    for s in range(theta.shape[0]):
        th = [theta[s, lay["offs"][i]:lay["offs"][i + 1]].reshape(shp) if len(shp) else float(theta[s, lay["offs"][i]])
              for i, shp in enumerate(lay["shapes"])]
        out.append(Particle(Θ=th, weight=float(weight[s]), id=int(ids[s])))
    return out


def get_optimal(de, model, particles):
    """utilities.jl:260-266"""
    best = particles[0]
    for p in particles:
        if (p.weight > best.weight) if de.update_particle is maximize else (p.weight < best.weight):
            best = p
    return dict(zip(model.names, best.Θ)), best.weight
