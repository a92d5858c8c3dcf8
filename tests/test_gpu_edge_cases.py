"""Edge cases of the hot path, HIP vs oracle (same harness as test_gpu_parity): ragged shapes (odd D, odd Np, N not a
multiple of the 16-observation MFMA tile, d not a multiple of the 4-dim k-step, D beyond one sub-group pass), degenerate
sampler settings, and the domain's "nulls": out-of-bounds particles (-Inf weights), NaN proposals (snooker drawing
itself as Pz in the reference schedules, crossover.jl:241), groups whose weights are all -Inf."""
import numpy as np
import pytest

from conftest import make_problem, setup_engine
from test_gpu_parity import teacher_forced

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,d", [(5, 1), (17, 3), (100, 4), (333, 7), (64, 33), (50, 64), (40, 65), (90, 100), (70, 130), (33, 200)])
def test_mvn_full_ragged_shapes(demc, orc, N, d):
    prob = make_problem("mvn_full", np.random.default_rng(100 + d), N=N, d=d)
    teacher_forced(demc, orc, prob, n_iter=4, n_groups=3, Np=9, schedule=2, burnin=2, check_hist=False)


@pytest.mark.parametrize("family", ["mvn_full", "mvn_iso"])
@pytest.mark.parametrize("offset", [1e3, 1e6])
def test_mvn_data_far_from_the_origin(demc, orc, family, offset):
    """the expanded quadratic form sum x'Ax - 2 y.x + N mu'A mu would lose all digits for data with a large mean; the
    library centres data and proposals, so the log-posterior keeps full precision whatever the offset"""
    rng = np.random.default_rng(31)
    prob = make_problem(family, rng, N=400, d=6)
    prob["data"] = prob["data"] + offset
    d = 6
    eng = demc.HipEngine(n_groups=2, Np=8, D=prob["D"], schedule=1)
    o = orc.Oracle(n_groups=2, Np=8, D=prob["D"], schedule=1)
    setup_engine(eng, prob)
    setup_engine(o, prob)
    th = prob["init"](16)
    th[:, :d] += offset + rng.normal(0, 0.05, (16, d))  # near the data, where the posterior mass is
    for mode in (0, 1):
        e2 = demc.HipEngine(n_groups=2, Np=8, D=prob["D"], schedule=1, loglike_mode=mode)
        setup_engine(e2, prob)
        # unbounded flat-ish comparison: priors are the same on both sides, so compare full log-posteriors
        np.testing.assert_allclose(e2.logpost(th), o.logpost(th), rtol=1e-10)
        e2.close()
    eng.close()


@pytest.mark.parametrize("Np", [4, 5, 7, 33, 600])
def test_odd_and_large_group_sizes(demc, orc, Np):
    prob = make_problem("gaussian", np.random.default_rng(7))
    teacher_forced(demc, orc, prob, n_iter=5, n_groups=2, Np=Np, schedule=2, burnin=3, alpha=0.5, check_hist=Np < 100)


def test_dimension_beyond_one_subgroup_pass(demc, orc):
    """D = 302 > 128: every lane owns several {2k,2k+1} pairs; hierarchical prior refers to theta[1]"""
    prob = make_problem("hier_binomial", np.random.default_rng(8), S=300)
    teacher_forced(demc, orc, prob, n_iter=4, n_groups=2, Np=6, schedule=2, burnin=2, theta_snooker=0.3, kappa=0.8,
                   check_hist=False)


def test_single_group_forces_alpha_zero(demc, orc):
    prob = make_problem("mvn_iso", np.random.default_rng(9), d=3)
    teacher_forced(demc, orc, prob, n_iter=8, n_groups=1, Np=8, schedule=1, burnin=4, alpha=0.9)


def test_two_groups_always_swap(demc, orc):
    prob = make_problem("binomial", np.random.default_rng(10))
    teacher_forced(demc, orc, prob, n_iter=10, n_groups=2, Np=6, schedule=2, burnin=5, alpha=1.0)


@pytest.mark.parametrize("kw", [dict(kappa=0.0), dict(eps=0.0), dict(beta=1.0), dict(beta=0.0, alpha=0.0),
                                dict(theta_snooker=1.0), dict(sigma=0.0, beta=1.0)])
def test_degenerate_sampler_settings(demc, orc, kw):
    """kappa = 0: every scalar is reset (proposal == current, always accepted); beta = 1: mutation only;
    theta_snooker = 1: snooker only"""
    prob = make_problem("mvn_iso", np.random.default_rng(11), d=4)
    teacher_forced(demc, orc, prob, n_iter=6, n_groups=3, Np=8, schedule=2, burnin=3, **kw)


def test_snooker_drawing_itself_gives_nan_proposal_and_is_rejected(demc, orc):
    """reference schedules draw Pz from the whole group incl. Pt (crossover.jl:241): Pd = 0 -> NaN -> out of bounds ->
    weight -Inf -> rejected.  Np = 4 and theta_snooker = 1 makes it frequent."""
    prob = make_problem("gaussian", np.random.default_rng(12))
    rng = np.random.default_rng(99)
    eng = demc.HipEngine(n_groups=8, Np=4, D=2, n_rows=6, schedule=1, theta_snooker=1.0, seed=5, trace=1)
    o = orc.Oracle(n_groups=8, Np=4, D=2, n_rows=6, schedule=1, theta_snooker=1.0, seed=5)
    setup_engine(eng, prob)
    setup_engine(o, prob)
    eng.set_state(prob["init"](32))
    n_nan = 0
    for it in range(1, 7):
        th, w, ids = eng.get_state()
        o.set_state(th, w, ids)
        eng.step(it, 1)
        o.step(it, 1)
        tg, to = eng.get_trace(), o.get_trace()
        assert np.array_equal(tg["idx"], to["idx"])
        nan_rows = np.isnan(to["proposal"]).any(1)
        assert np.array_equal(np.isnan(tg["proposal"]).any(1), nan_rows)
        assert np.all(tg["w_prop"][nan_rows] == -np.inf) and not tg["accepted"][nan_rows].any()
        assert np.array_equal(tg["accepted"], to["accepted"])
        n_nan += int(nan_rows.sum())
    assert n_nan > 10
    eng.close()


def test_out_of_bounds_particles_and_all_minus_inf_group(demc, orc):
    """particles that start out of bounds have weight -Inf (utilities.jl:92-99); a group whose weights are all -Inf makes
    the select_base softmax degenerate (uniform fallback) and the migration pick fall back to argmin"""
    prob = make_problem("gaussian", np.random.default_rng(13))
    G, Np = 4, 6
    th0 = prob["init"](G * Np)
    th0[:Np, 1] = -1.0          # whole group 0 out of bounds (sigma < 0)
    th0[Np + 1, 1] = -2.0       # one particle of group 1
    cfg = dict(n_groups=G, Np=Np, D=2, n_rows=8, schedule=2, burnin=8, alpha=1.0, seed=3)
    eng, o = demc.HipEngine(trace=1, **cfg), orc.Oracle(**cfg)
    setup_engine(eng, prob)
    setup_engine(o, prob)
    eng.set_state(th0)
    _, w, _ = eng.get_state()
    assert np.all(w[:Np] == -np.inf) and w[Np + 1] == -np.inf and np.isfinite(w[Np:]).sum() == 3 * Np - 1
    for it in range(1, 9):
        th, w, ids = eng.get_state()
        o.set_state(th, w, ids)
        eng.step(it, 1)
        o.step(it, 1)
        tg, to = eng.get_trace(), o.get_trace()
        assert np.array_equal(tg["idx"], to["idx"])
        np.testing.assert_allclose(tg["proposal"], to["proposal"], rtol=1e-11, atol=1e-13)
        assert np.array_equal(tg["accepted"], to["accepted"])
        sg, so = eng.get_state(), o.get_state()
        assert np.array_equal(sg[2], so[2])
        assert np.array_equal(np.isfinite(sg[1]), np.isfinite(so[1]))
    # -Inf -> finite is always accepted, so the dead group comes back to life
    assert np.isfinite(eng.get_state()[1]).all()
    eng.close()


def test_zero_iterations_and_history_bounds(demc):
    prob = make_problem("gaussian", np.random.default_rng(14))
    eng = demc.HipEngine(n_groups=2, Np=4, D=2, n_rows=3, schedule=1)
    setup_engine(eng, prob)
    eng.set_state(prob["init"](8))
    eng.step(1, 0)                                    # empty run is a no-op
    with pytest.raises(demc.DemcError):
        eng.step(0, 1)                                # de.iter is 1-based
    with pytest.raises(demc.DemcError):
        eng.get_history(0, 4)                         # beyond n_rows
    eng.step(1, 5)                                    # iterations beyond n_rows still run; their rows are not stored
    th, acc, lp, idh = eng.get_history(0, 3)
    assert th.shape == (3, 8, 2) and np.isfinite(lp).all()
    eng.close()


def test_cfg4_shape_parity(demc, orc):
    """BASELINE cfg4's shape at full width: hierarchical Binomial with S = 1e4 subjects (D = 10 002), the two blocks
    [hyper ; subjects] of Examples/Hierarchical_Example.jl:88-92, snooker on -- a few groups instead of 128"""
    prob = make_problem("hier_binomial", np.random.default_rng(44), S=10000)
    D_ = prob["D"]
    m0 = np.zeros(D_, np.uint8)
    m0[:2] = 1
    teacher_forced(demc, orc, prob, n_iter=3, n_groups=2, Np=6, schedule=2, burnin=2, theta_snooker=0.2, alpha=1.0,
                   masks=np.stack([m0, 1 - m0]), exact_de=False)


@pytest.mark.parametrize("S", [2500, 5000])
def test_whole_workgroup_per_particle_both_widths(demc, orc, S):
    """very long rows: one particle per workgroup, 256 threads for 2048 <= D < 4096 and 512 from D = 4096 (the cfg4 test above
    runs the 512 form with blocks); hierarchical Binomial so that the subject terms are summed inside the proposal kernel"""
    prob = make_problem("hier_binomial", np.random.default_rng(46), S=S)
    teacher_forced(demc, orc, prob, n_iter=3, n_groups=2, Np=8, schedule=2, burnin=2, alpha=1.0, exact_de=False)


def test_cfg5_shape_parity(demc, orc):
    """BASELINE cfg5's shape with fewer trials/groups: LBA, 3 accumulators (6 parameters), snooker on"""
    prob = make_problem("lba", np.random.default_rng(45), N=2000, na=3)
    teacher_forced(demc, orc, prob, n_iter=6, n_groups=4, Np=16, schedule=2, burnin=3, theta_snooker=0.1)


def _fuzz_cases(n, seed=20261003):
    rng = np.random.default_rng(seed)
    fams = ["gaussian", "binomial", "mvn_iso", "mvn_full", "hier_binomial", "hier_gaussian", "lba", "lnr"]
    out = []
    for i in range(n):
        fam = fams[i % len(fams)]
        schedule = int(rng.integers(1, 3))
        snooker = float(rng.choice([0.0, 0.0, 0.3, 1.0]))
        Np = int(rng.integers(6, 40))
        cfg = dict(n_groups=int(rng.integers(1, 7)), Np=Np, schedule=schedule, burnin=int(rng.integers(0, 4)),
                   theta_snooker=snooker, kappa=float(rng.choice([1.0, 1.0, 0.6])), beta=float(rng.choice([0.1, 0.5])),
                   alpha=float(rng.choice([0.1, 1.0])), proposal_kind=int(rng.integers(0, 3)), seed=int(rng.integers(1, 2**31)),
                   loglike_mode=int(rng.integers(0, 2)))
        kw = {}
        if fam in ("mvn_iso", "mvn_full"):
            kw = dict(N=int(rng.integers(3, 400)), d=int(rng.integers(1, 40)))
        elif fam in ("gaussian", "binomial", "lba", "lnr"):
            kw = dict(N=int(rng.integers(1, 300)))
        elif fam == "hier_binomial":
            kw = dict(S=int(rng.integers(2, 200)))
        elif fam == "hier_gaussian":
            kw = dict(S=int(rng.integers(2, 40)), n=int(rng.integers(1, 9)))
        out.append(pytest.param(fam, kw, cfg, id=f"{i}-{fam}-s{schedule}"))
    return out


@pytest.mark.parametrize("fam,kw,cfg", _fuzz_cases(int(__import__("os").environ.get("DEMC_FUZZ_CASES", "32")), int(__import__("os").environ.get("DEMC_FUZZ_SEED", "20261003"))))
def test_randomised_configurations(demc, orc, fam, kw, cfg):
    """seeded sweep over families x shapes x sampler settings x schedules x likelihood modes"""
    prob = make_problem(fam, np.random.default_rng(cfg["seed"]), **kw)
    teacher_forced(demc, orc, prob, n_iter=4, exact_de=False, **cfg)


def test_repeat_runs_agree_bit_for_bit(demc):
    """no atomics, fixed reduction trees, every host<->device hand-over ordered: a configuration run twice on fresh handles
    gives the same bits in state, history and chain export, in all three fuse modes (tools/determinism_sweep.py is the
    long form of this test)"""
    import hashlib
    for case in _fuzz_cases(16, seed=11):
        fam, kw, cfg = case.values
        prob = make_problem(fam, np.random.default_rng(cfg["seed"]), **kw)
        th0 = prob["init"](cfg["n_groups"] * cfg["Np"])
        for fuse in (0, 2, 1):
            sigs = set()
            for _ in range(2):
                e = demc.HipEngine(D=prob["D"], n_rows=12, trace=0, fuse=fuse, **cfg)
                setup_engine(e, prob)
                e.set_state(th0)
                e.step(1, 12)
                parts = list(e.get_state()) + list(e.get_history(0, 12)) + [e.export_chains(0, 12)]
                e.close()
                sigs.add(hashlib.md5(b"".join(np.ascontiguousarray(p).tobytes() for p in parts)).hexdigest())
            assert len(sigs) == 1, (case.id, fuse)


def test_longrow_repeat_runs_agree_bit_for_bit(demc):
    """the long-row kernel (span loops, edge rounds one scalar per lane, either workgroup size) twice on fresh handles:
    the same bits in state, history and chain export"""
    import hashlib
    for fam, kw, extra in (("hier_binomial", dict(S=2600), dict(theta_snooker=0.3, beta=0.3)),
                           ("hier_binomial", dict(S=2600), dict(geometry_groups=512)),
                           ("hier_gaussian", dict(S=1500, n=3), dict(kappa=0.8))):
        prob = make_problem(fam, np.random.default_rng(5), **kw)
        D = prob["D"]
        m0 = np.zeros(D, np.uint8)
        m0[:2] = 1
        th0 = prob["init"](3 * 10)
        sigs = set()
        for _ in range(2):
            e = demc.HipEngine(D=D, n_groups=3, Np=10, n_rows=10, trace=0, burnin=4, seed=77, n_blocks=2, **extra)
            setup_engine(e, prob)
            e.set_blocks(np.stack([m0, 1 - m0]))
            e.set_state(th0)
            e.step(1, 10)
            parts = list(e.get_state()) + list(e.get_history(0, 10)) + [e.export_chains(0, 10)]
            e.close()
            sigs.add(hashlib.md5(b"".join(np.ascontiguousarray(q).tobytes() for q in parts)).hexdigest())
        assert len(sigs) == 1, (fam, extra)



@pytest.mark.parametrize("extra", [dict(kappa=0.999), dict(theta_snooker=0.3, beta=0.3, kappa=0.999), dict(kappa=0.8)])
def test_persistent_longrow_repeat_runs_agree_bit_for_bit(demc, extra):
    """the PERSISTENT form of the long-row kernel (round 4): 96 x 32 particles of 2 102 scalars are 1 536 moving particles per
    colour phase for 512 resident workgroups -- three each, every row move but the last one of a workgroup issued from inside
    the next particle's span loops (LDS copy reused across particles, parity-buffered partial sums, LDS-only barriers).  Four
    runs on fresh handles must agree bit for bit in state, history and chain export: a hazard between consecutive particles of
    a workgroup would show as a run that differs.  Snooker / mutation / recombination variants take the other span forms and
    the general body (no deferral).  (kappa < 1: with kappa = 1 this population is served by the row-streaming kernel,
    k_frozen_sweep -- one particle per workgroup, nothing carried between particles; its repeat runs are the next test.)"""
    import hashlib
    from demc_amd import workloads as W
    w = W.cfg4(S=2100, G=96, Np=32)
    th0 = w["init"](96 * 32, np.random.default_rng(9))
    sigs = set()
    for _ in range(4):
        cfg = dict(n_groups=96, Np=32, D=w["D"], n_rows=6, schedule=2, seed=123, burnin=3, trace=0, alpha=0.4)
        cfg.update(extra)
        e = demc.HipEngine(**cfg)
        W.configure(e, w)
        e.set_state(th0)
        e.step(1, 6)
        assert e.last_kernels() == "k_longrow<256>"
        parts = list(e.get_state()) + list(e.get_history(0, 6)) + [e.export_chains(0, 6)]
        e.close()
        sigs.add(hashlib.md5(b"".join(np.ascontiguousarray(q).tobytes() for q in parts)).hexdigest())
    assert len(sigs) == 1, extra


@pytest.mark.parametrize("extra", [dict(), dict(theta_snooker=0.3, beta=0.3), dict(schedule=1, partner_kind=1, n_initial=3, theta_snooker=0.1, burnin=100)])
def test_row_streaming_kernel_repeat_runs_agree_bit_for_bit(demc, extra):
    """k_frozen_sweep in both instances (hyper-parameter block, subject block) on the same population: four runs on fresh handles
    agree bit for bit -- partners from the population and from the history (inside burn-in: the base row from the sweep-start
    snapshot, whole for the subject sweep, the block's columns for the hyper-parameter sweep)."""
    import hashlib
    from demc_amd import workloads as W
    w = W.cfg4(S=2100, G=96, Np=32)
    rng = np.random.default_rng(9)
    n_init = extra.get("n_initial", 0)
    rows0 = np.stack([w["init"](96 * 32, rng) for _ in range(n_init)]) if n_init else None
    th0 = w["init"](96 * 32, rng)
    sigs = set()
    for _ in range(4):
        cfg = dict(n_groups=96, Np=32, D=w["D"], n_rows=n_init + 5, schedule=2, seed=123, burnin=3, trace=0, alpha=0.4)
        cfg.update(extra)
        e = demc.HipEngine(**cfg)
        W.configure(e, w)
        if n_init:
            e.set_history_rows(0, rows0)
        e.set_state(th0)
        e.step(1 + n_init, 5)
        assert e.last_kernels() == "k_frozen_sweep<256,big>"
        parts = list(e.get_state()) + list(e.get_history(n_init, n_init + 5)) + [e.export_chains(n_init, n_init + 5)]
        e.close()
        sigs.add(hashlib.md5(b"".join(np.ascontiguousarray(q).tobytes() for q in parts)).hexdigest())
    assert len(sigs) == 1, extra


def test_lnr_log_survival_table_against_the_oracle_far_into_both_tails(demc, orc):
    """the LNR's log Phi(-z) comes from a polynomial table on [-8.5, 38.5] and from Mills' ratio beyond: drifts that put the
    losing accumulators at z from -30 (survival 1) to +60 (survival 1e-784: log-survival -1800) against the oracle's libm form"""
    from demc_amd import families as F
    rng = np.random.default_rng(5)
    N, na = 40, 3
    choice = rng.integers(1, na + 1, N).astype(float)
    rt = rng.uniform(0.5, 1.5, N)
    data = np.concatenate([choice, rt])
    D = na + 1
    th = np.zeros((16, D))
    th[:, :na] = np.linspace(-62.0, 30.0, 16)[:, None] + rng.normal(0, 0.3, (16, na))  # z = (log t - nu)/sigma
    th[:, na] = 0.1
    outs = []
    for mk in (demc.HipEngine, orc.Oracle):
        e = mk(n_groups=2, Np=8, D=D, schedule=1)
        e.set_model(F.FAM_LNR, data, [N, na], [1.0])
        e.set_priors([0] * D, [0.0] * D, [1.0] * D)
        e.set_bounds([-np.inf] * na + [0.0], [np.inf] * na + [0.4])
        outs.append(e.logpost(th))
        e.close()
    assert np.isfinite(outs[1]).all() and outs[1].min() < -5e4
    np.testing.assert_allclose(outs[0], outs[1], rtol=1e-9)


@pytest.mark.parametrize("G,Np,d,burnin,kernel", [(256, 256, 32, 1000, "k_res_mvn<512,false,32,2>"), (256, 256, 32, 0, "k_res_mvn<512,false,32,1>"),
                                                  (128, 64, 8, 1000, "k_res_mvn<256,false,8,2>")])
def test_de_mc_z_lean_body_repeat_runs_agree_bit_for_bit(demc, G, Np, d, burnin, kernel):
    """k_res_mvn<..., HIST> at BASELINE's full cfg3 population (256 workgroups, every CU busy): three runs of 30 iterations on
    fresh handles must agree bit for bit in state and history.  Inside burn-in a particle's base row is a row of the group's current
    population that another wave of the workgroup writes in the same launch; the barrier in front of the first store is what
    makes the outcome independent of which wave is ahead (a race there showed as a handful of differing decisions per run)."""
    import hashlib
    from demc_amd import workloads as W
    w = W.cfg3(N=2000, d=d, G=G, Np=Np)
    n_init, n_it = 8, 30
    rng = np.random.default_rng(3)
    P = G * Np
    rows0 = np.stack([w["init"](P, rng) for _ in range(n_init)])
    th0 = w["init"](P, rng)
    sigs = set()
    for _ in range(3):
        e = demc.HipEngine(n_groups=G, Np=Np, D=w["D"], n_rows=n_init + n_it, schedule=1, seed=99, burnin=burnin, trace=0,
                           loglike_mode=1, partner_kind=1, n_initial=n_init, **w["engine"])
        W.configure(e, w)
        e.set_history_rows(0, rows0)
        e.set_state(th0)
        e.step(1 + n_init, n_it)
        assert e.last_kernels() == kernel
        parts = list(e.get_state()) + list(e.get_history(n_init, n_init + n_it))
        e.close()
        sigs.add(hashlib.md5(b"".join(np.ascontiguousarray(q).tobytes() for q in parts)).hexdigest())
    assert len(sigs) == 1
