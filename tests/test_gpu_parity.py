"""GPU parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): RNG / index bookkeeping bit-exact; log-posteriors within 1e-6 relative (we
assert 1e-9); DE-branch proposals bit-exact (pure +,-,* in a fixed order); snooker / mutation proposals 1e-11
(dot-product reduction order and libm vs device sin/cos/log differ in the last ulp)."""
import numpy as np
import pytest

from conftest import make_problem, setup_engine

pytestmark = pytest.mark.gpu

LOGPOST_RTOL = 1e-9


def _rtol(prob):
    """1e-9 everywhere except the LBA.  Its survival factor 1 - F(t), F(t) = 1 + (t/A)[n1 Phi(n1) - n2 Phi(n2) + phi(n1)
    - phi(n2)], is formed by cancellation (as in SequentialSamplingModels and in the oracle): for a trial far in the
    tail (1 - F ~ 1e-11, the density just above the 1e-10 floor) the last-bit differences between libm's erfc and the
    device's Phi table are amplified to ~1e-4 in that trial's log-density -- for ANY two implementations of the
    formula.  Typical agreement is 1e-12; the seeded 400-case sweep needs 1e-5 (one such trial in ~230 cases)."""
    from demc_amd import families as F
    return 1e-5 if prob["fam"] == F.FAM_LBA else LOGPOST_RTOL


def _pair(demc, orc, prob, **cfg):
    base = dict(D=prob["D"], seed=cfg.pop("seed", 1234), trace=1)  # (these tests read the per-slot trace: they opt in)
    base.update(cfg)
    eng = demc.HipEngine(**base)
    o = orc.Oracle(**{k: v for k, v in base.items() if k in orc.CFG_KEYS})
    setup_engine(eng, prob)
    setup_engine(o, prob)
    return eng, o


def teacher_forced(demc, orc, prob, n_iter, n_initial=0, masks=None, check_hist=True, exact_de=True, **cfg):
    rng = np.random.default_rng(99)
    ng, Np = cfg["n_groups"], cfg["Np"]
    P = ng * Np
    cfg.setdefault("n_rows", n_iter + n_initial)
    cfg["n_initial"] = n_initial
    eng, o = _pair(demc, orc, prob, **cfg)
    if masks is not None:
        eng.set_blocks(masks)
        o.set_blocks(masks)
    theta0 = prob["init"](P)
    if n_initial:
        rows = np.stack([prob["init"](P) for _ in range(n_initial)])
        eng.set_history_rows(0, rows)
        o.set_history_rows(0, rows)
        theta0 = rows[0].copy()
    eng.set_state(theta0)
    th, w, ids = eng.get_state()
    w_o = o.logpost(th)
    fin = np.isfinite(w_o)
    assert np.array_equal(np.isfinite(w), fin)
    np.testing.assert_allclose(w[fin], w_o[fin], rtol=_rtol(prob))
    n_mis = 0
    for it in range(1 + n_initial, 1 + n_initial + n_iter):
        th, w, ids = eng.get_state()
        o.set_state(th, w, ids)  # teacher forcing: both engines step from the device state
        eng.step(it, 1)
        o.step(it, 1)
        tg, to = eng.get_trace(), o.get_trace()
        # RNG / index bookkeeping: bit-exact
        assert np.array_equal(tg["idx"], to["idx"]), f"iter {it}: partner/branch indices differ"
        kind = to["idx"][:, 0]
        de = kind == 0
        # DE-branch proposals are bit-exact whenever their inputs are: always in the synchronous schedule; in
        # two_colour the second phase reads rows the first phase may have accepted from a snooker / mutation
        # proposal (1-ulp differences), so there bit-exactness is asserted for the first phase only.
        exact = de.copy() if exact_de else np.zeros(P, bool)
        if cfg["schedule"] == 2:
            first = (np.arange(P) % Np) < Np // 2
            if not (kind[first] == 0).all():
                exact &= first
        if cfg["schedule"] == 0 and not (kind == 0).all():
            # sequential: particle k reads rows that 0..k-1 may have accepted from a snooker / mutation proposal
            exact[:] = False
        assert np.array_equal(tg["proposal"][exact], to["proposal"][exact]), f"iter {it}: DE proposals not bit-exact"
        np.testing.assert_allclose(tg["proposal"], to["proposal"], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(tg["log_adj"], to["log_adj"], rtol=1e-9, atol=1e-9)
        fin = np.isfinite(to["w_prop"])
        assert np.array_equal(np.isfinite(tg["w_prop"]), fin)
        np.testing.assert_allclose(tg["w_prop"][fin], to["w_prop"][fin], rtol=_rtol(prob))
        mism = tg["accepted"] != to["accepted"]
        n_mis += int(mism.sum())
        th_g, w_g, id_g = eng.get_state()
        th_o, w_o2, id_o = o.get_state()
        assert np.array_equal(id_g, id_o)
        if not mism.any():
            assert np.array_equal(th_g[exact], th_o[exact])
            np.testing.assert_allclose(th_g, th_o, rtol=1e-11, atol=1e-13)
    # a knife-edge accept flip needs |u - ratio| ~ 1e-12: with these sizes none is expected
    assert n_mis == 0, f"{n_mis} accept decisions differ"
    if check_hist:
        hg = eng.get_history(0, n_iter + n_initial)
        ho = o.get_history(0, n_iter + n_initial)
        assert np.array_equal(hg[3], ho[3])  # ids per slot
        assert np.array_equal(hg[1][n_initial:], ho[1][n_initial:])  # accept flags
        np.testing.assert_allclose(hg[0], ho[0], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(hg[2][n_initial:], ho[2][n_initial:], rtol=_rtol(prob))
    eng.close()
    o.close()


FAMILIES = ["gaussian", "binomial", "mvn_iso", "mvn_full", "hier_binomial", "hier_gaussian", "lba", "lnr"]


@pytest.mark.parametrize("family", FAMILIES)
def test_logpost_matches_oracle(demc, orc, family):
    rng = np.random.default_rng(7)
    prob = make_problem(family, rng)
    eng, o = _pair(demc, orc, prob, n_groups=3, Np=8, schedule=1)
    th = prob["init"](24)
    th[1, 0] = th[1, 0]  # in-bounds rows
    lg, lo = eng.logpost(th), o.logpost(th)
    fin = np.isfinite(lo)
    assert fin.sum() >= 20
    assert np.array_equal(np.isfinite(lg), fin)
    np.testing.assert_allclose(lg[fin], lo[fin], rtol=_rtol(prob))
    # out-of-bounds rows are -Inf and never evaluated (utilities.jl:92-99)
    if np.isfinite(prob["lo"]).any():
        j = int(np.argmax(np.isfinite(prob["lo"])))
        th[0, j] = prob["lo"][j] - 1.0
        assert eng.logpost(th)[0] == -np.inf
    eng.close()


@pytest.mark.parametrize("family", FAMILIES)
@pytest.mark.parametrize("schedule", [0, 1, 2])
def test_step_parity_default_sampler(demc, orc, family, schedule):
    """defaults of DE(): random_gamma, alpha = beta = 0.1, kappa = 1, no snooker; burn-in covers the base term"""
    prob = make_problem(family, np.random.default_rng(11))
    teacher_forced(demc, orc, prob, n_iter=12, n_groups=4, Np=8, schedule=schedule, burnin=6)


@pytest.mark.parametrize("schedule", [0, 1, 2])
@pytest.mark.parametrize("proposal_kind", [0, 1, 2])
def test_step_parity_snooker_recombination(demc, orc, schedule, proposal_kind):
    prob = make_problem("mvn_iso", np.random.default_rng(12), d=6)
    teacher_forced(demc, orc, prob, n_iter=10, n_groups=3, Np=10, schedule=schedule, burnin=4, theta_snooker=0.4,
                   kappa=0.7, beta=0.2, proposal_kind=proposal_kind)


def test_step_parity_blocks(demc, orc):
    """block_update! (main.jl:174-179) + reset! (crossover.jl:336-352): one sweep per block"""
    prob = make_problem("hier_gaussian", np.random.default_rng(13), S=6, n=5)
    D = prob["D"]
    m0 = np.zeros(D, np.uint8)
    m0[[0, 1, D - 1]] = 1
    masks = np.stack([m0, 1 - m0])
    # the second block's sweep starts from rows the first sweep accepted (possibly 1-ulp-different snooker rows),
    # so proposals are compared to 1e-11 here; indices, masks and accept decisions stay exact
    teacher_forced(demc, orc, prob, n_iter=8, n_groups=3, Np=8, schedule=2, burnin=3, theta_snooker=0.2, masks=masks,
                   exact_de=False)
    teacher_forced(demc, orc, prob, n_iter=8, n_groups=3, Np=8, schedule=1, burnin=3, beta=0.0, masks=masks[:1])


@pytest.mark.parametrize("schedule", [1, 2])
def test_step_parity_history_partners(demc, orc, schedule):
    """sample = resample (DE-MC_Z, crossover.jl:113-124) with snooker, as test/multivariate_normal_tests.jl:50-59"""
    prob = make_problem("mvn_iso", np.random.default_rng(14), d=4)
    # partners come from the engine's own history, which is not teacher-forced and may hold rows accepted from
    # 1-ulp-different snooker / mutation proposals: proposals are compared to 1e-11, indices stay exact
    teacher_forced(demc, orc, prob, n_iter=10, n_initial=6, n_groups=2, Np=4, schedule=schedule, burnin=5,
                   theta_snooker=0.3, partner_kind=1, alpha=0.0, exact_de=False)


@pytest.mark.parametrize("schedule", [0, 2])
def test_migration_parity(demc, orc, schedule):
    """alpha = 1: every iteration migrates; ids travel with the particles (migration.jl:84-91)"""
    prob = make_problem("gaussian", np.random.default_rng(15))
    teacher_forced(demc, orc, prob, n_iter=15, n_groups=6, Np=6, schedule=schedule, burnin=5, alpha=1.0)


def test_reference_schedule_blocks_and_suffstat(demc, orc):
    """the reference's own schedule (sequential in-place sweep, crossover.jl:13-15) on the device: with block updates
    (one sweep per block, main.jl:174-179) and in SUFFSTAT mode (the whole update in the fused K1 tail, one particle of
    every group per launch)"""
    prob = make_problem("hier_gaussian", np.random.default_rng(13), S=6, n=5)
    D = prob["D"]
    m0 = np.zeros(D, np.uint8)
    m0[[0, 1, D - 1]] = 1
    teacher_forced(demc, orc, prob, n_iter=6, n_groups=3, Np=8, schedule=0, burnin=3, theta_snooker=0.2,
                   masks=np.stack([m0, 1 - m0]), exact_de=False)
    prob = make_problem("mvn_full", np.random.default_rng(21), N=300, d=7)
    teacher_forced(demc, orc, prob, n_iter=8, n_groups=4, Np=12, schedule=0, burnin=4, theta_snooker=0.2,
                   loglike_mode=1, alpha=0.5)


def test_optimize_modes(demc, orc):
    """greedy DE: evaluate_fun! + maximize!/minimize! (utilities.jl:113-120, 212-226)"""
    prob = make_problem("gaussian", np.random.default_rng(16))
    teacher_forced(demc, orc, prob, n_iter=10, n_groups=3, Np=8, schedule=1, burnin=0, update_kind=1, fitness_kind=1,
                   check_hist=False)


def test_large_tile_mvn_full(demc, orc):
    """D = 32 full-Sigma (the headline shape, reduced N): exercises KS = 8 MFMA k-steps, several particle tiles
    and observation chunks"""
    prob = make_problem("mvn_full", np.random.default_rng(17), N=3000, d=32)
    teacher_forced(demc, orc, prob, n_iter=3, n_groups=5, Np=70, schedule=2, burnin=1, check_hist=False)


def test_suffstat_mode_matches_streaming(demc, orc):
    prob = make_problem("mvn_full", np.random.default_rng(18), N=500, d=8)
    th = prob["init"](64)
    vals = []
    for mode in (0, 1):
        eng = demc.HipEngine(n_groups=4, Np=16, D=8, schedule=1, loglike_mode=mode)
        setup_engine(eng, prob)
        vals.append(eng.logpost(th))
        eng.close()
    np.testing.assert_allclose(vals[0], vals[1], rtol=1e-10)


@pytest.mark.parametrize("family", ["mvn_full", "mvn_iso"])
@pytest.mark.parametrize("schedule", [1, 2])
def test_step_parity_suffstat_fused(demc, orc, family, schedule):
    """SUFFSTAT likelihood: with two_colour the whole update runs in the fused K1 tail (one launch per colour phase);
    the oracle still visits every observation"""
    prob = make_problem(family, np.random.default_rng(21), N=300, d=7)
    teacher_forced(demc, orc, prob, n_iter=10, n_groups=4, Np=12, schedule=schedule, burnin=5, theta_snooker=0.2,
                   loglike_mode=1, alpha=0.5)


def test_fused_tail_with_several_workgroups_per_group(demc, orc):
    """Np = 200 -> two workgroups per group in the fused kernel: they write active rows / weights of the same group
    concurrently, which is safe because everything a moving particle READS (partners, base row, base weights) lives
    in the other colour.  Burn-in covers the base term."""
    prob = make_problem("mvn_full", np.random.default_rng(23), N=300, d=8)
    teacher_forced(demc, orc, prob, n_iter=6, n_groups=3, Np=200, schedule=2, burnin=6, loglike_mode=1, alpha=0.5,
                   check_hist=False)


def _run_fuse_mode(demc, prob, th0, fuse, G, Np, D, n_iter, blocks=None, **kw):
    e = demc.HipEngine(n_groups=G, Np=Np, D=D, n_rows=n_iter, schedule=2, seed=3, fuse=fuse, n_blocks=0 if blocks is None else len(blocks),
                       **kw)
    setup_engine(e, prob)
    if blocks is not None:
        e.set_blocks(blocks)
    e.set_state(th0)
    e.step(1, n_iter)
    out = e.get_history(0, n_iter) + e.get_state()
    e.close()
    return out


def test_fused_equals_unfused(demc):
    """the fused tail and the resident form are implementation details: fuse = 0 (resident K1: one launch per run of
    iterations between migrations), 2 (one fused launch per colour phase) and 1 (K1 -> K3) give the same bits"""
    prob = make_problem("mvn_full", np.random.default_rng(22), N=400, d=10)
    th0 = prob["init"](6 * 20)
    hs = [_run_fuse_mode(demc, prob, th0, fuse, 6, 20, 10, 25, loglike_mode=1, theta_snooker=0.1, alpha=0.3, burnin=10)
          for fuse in (0, 2, 1)]
    for other in hs[1:]:
        for a, b in zip(hs[0], other):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("family,G,Np,d,N,kw", [
    ("mvn_full", 32, 64, 8, 10000, dict()),                              # BASELINE cfg2: 8 workgroups per group, X chunk in LDS
    ("mvn_full", 5, 20, 10, 400, dict(theta_snooker=0.2)),               # odd group count (plain block order), snooker
    ("mvn_iso", 8, 33, 5, 700, dict(kappa=0.7)),                         # iso family (sigma is a parameter), odd D, odd halves
    ("mvn_full", 3, 300, 16, 333, dict()),                               # 150 moving particles: several particle-tile groups
    ("mvn_full", 16, 16, 33, 2000, dict(beta=0.5)),                      # d > 32: VALU preparation, 16 k-steps, mutation
    ("mvn_full", 200, 8, 3, 50, dict()),                                 # more groups than fit twice: one workgroup per group
])
def test_streaming_resident_form_equals_the_launch_chain(demc, family, G, Np, d, N, kw):
    """STREAMING likelihood of a small population: fuse = 0 keeps the group resident AND streams the observations inside the
    same kernel (several workgroups per group, hand-over of the per-chunk cross terms); fuse = 1 is the K1 -> K2 -> K3 chain
    per colour phase.  Same samples, acceptances and ids across migrations and burn-in; log-densities to rounding (the
    observation sums are split differently)."""
    prob = make_problem(family, np.random.default_rng(81), N=N, d=d)
    D = prob["D"]
    th0 = prob["init"](G * Np)
    a, b = (_run_fuse_mode(demc, prob, th0, fuse, G, Np, D, 24, alpha=0.3, burnin=10, loglike_mode=0, **kw) for fuse in (0, 1))
    for i, (x, y) in enumerate(zip(a, b)):
        if i in (2, 5):
            np.testing.assert_allclose(x, y, rtol=1e-10)
        elif i in (0, 4) and kw.get("theta_snooker", 0.0) > 0.0:
            np.testing.assert_allclose(x, y, rtol=0, atol=1e-11)
        else:
            assert np.array_equal(x, y), f"array {i}"


@pytest.mark.parametrize("family,S,kw", [
    ("hier_binomial", 2500, dict()),                                      # D = 2502, default sampler inside burn-in (base term)
    ("hier_binomial", 5001, dict(theta_snooker=0.3, kappa=0.8)),          # odd D (no 16-byte row accesses), snooker, recombination
    ("hier_gaussian", 2100, dict(beta=0.5)),                              # several observations per subject, mutation
    ("hier_binomial", 4100, dict(blocks=True, theta_snooker=0.2)),        # two block sweeps [hyper ; subject], snooker
])
def test_longrow_kernel_equals_generic_fused_kernel(demc, family, S, kw):
    """rows of thousands of scalars run in the one-pass long-row kernel (fuse = 0); fuse = 2 keeps k_propose's fused form
    with a workgroup per particle.  Same proposals, priors and decisions; the likelihood sums run in a different order."""
    kw = dict(kw)
    prob = make_problem(family, np.random.default_rng(91), S=S, n=4)
    D = prob["D"]
    blocks = None
    if kw.pop("blocks", False):
        m0 = np.zeros(D, np.uint8)
        m0[:2] = 1
        blocks = np.stack([m0, 1 - m0])
    G, Np = 3, 10
    th0 = prob["init"](G * Np)
    a, b = (_run_fuse_mode(demc, prob, th0, fuse, G, Np, D, 12, blocks=blocks, alpha=0.3, burnin=6, **kw) for fuse in (0, 2))
    for i, (x, y) in enumerate(zip(a, b)):
        if i in (2, 5):
            np.testing.assert_allclose(x, y, rtol=1e-10)
        elif i in (0, 4) and kw.get("theta_snooker", 0.0) > 0.0:
            np.testing.assert_allclose(x, y, rtol=0, atol=1e-11)
        else:
            assert np.array_equal(x, y), f"array {i}"


def test_longrow_kernel_against_the_oracle(demc, orc):
    """teacher-forced against the oracle: sequential (reference) and two_colour schedules, history partners, blocks"""
    prob = make_problem("hier_binomial", np.random.default_rng(92), S=2200)
    D = prob["D"]
    m0 = np.zeros(D, np.uint8)
    m0[:2] = 1
    teacher_forced(demc, orc, prob, n_iter=5, n_groups=2, Np=8, schedule=2, burnin=3, theta_snooker=0.25, kappa=0.9,
                   masks=np.stack([m0, 1 - m0]), exact_de=False)
    teacher_forced(demc, orc, prob, n_iter=4, n_groups=2, Np=6, schedule=0, burnin=2, theta_snooker=0.25, exact_de=False)
    teacher_forced(demc, orc, prob, n_iter=4, n_initial=3, n_groups=2, Np=6, schedule=2, burnin=2, partner_kind=1, alpha=0.0,
                   exact_de=False)


def test_longrow_span_loops_against_the_oracle(demc, orc):
    """the long-row kernel's forms, each teacher-forced against the oracle (the trace is on: proposals are compared scalar by
    scalar): span loops per uniform region -- crossover with and without the base row (burn-in 2 of 4 iterations), snooker,
    mutation (beta = 0.4), frozen rows of a block sweep -- with the rounds around the hyper-parameters and at the ragged end
    done one scalar per lane; two 256-thread workgroups per CU (geometry_groups makes the moving particles outnumber twice
    the CUs); a block mask with more runs than the kernarg table holds, a long segment with a Cauchy prior and an odd row length (no
    span loops: the general body / one scalar per lane throughout)."""
    prob = make_problem("hier_binomial", np.random.default_rng(93), S=2600)
    D = prob["D"]
    m0 = np.zeros(D, np.uint8)
    m0[:2] = 1
    blocks = np.stack([m0, 1 - m0])
    for geo in (0, 512):
        teacher_forced(demc, orc, prob, n_iter=4, n_groups=2, Np=8, schedule=2, burnin=2, theta_snooker=0.3, beta=0.4,
                       masks=blocks, exact_de=False, geometry_groups=geo)
    teacher_forced(demc, orc, prob, n_iter=3, n_groups=2, Np=6, schedule=2, burnin=1, beta=0.3, exact_de=False)  # no blocks
    stripes = (np.arange(D) // 37 % 2).astype(np.uint8)  # 70 runs
    teacher_forced(demc, orc, prob, n_iter=3, n_groups=2, Np=6, schedule=2, burnin=1, theta_snooker=0.2,
                   masks=np.stack([stripes, 1 - stripes]), exact_de=False)
    cauchy = dict(prob)
    cauchy["pk"] = [1, 2] + [5] * 1300 + [9] * 1300  # PR_CAUCHY on the second half of the subjects
    teacher_forced(demc, orc, cauchy, n_iter=3, n_groups=2, Np=6, schedule=2, burnin=1, theta_snooker=0.2, masks=blocks,
                   exact_de=False)
    odd = make_problem("hier_binomial", np.random.default_rng(95), S=2601)  # D = 2603: unaligned rows, the general body
    mo = np.zeros(odd["D"], np.uint8)
    mo[:2] = 1
    teacher_forced(demc, orc, odd, n_iter=3, n_groups=2, Np=6, schedule=2, burnin=1, theta_snooker=0.2, beta=0.3,
                   masks=np.stack([mo, 1 - mo]), exact_de=False)
    hg = make_problem("hier_gaussian", np.random.default_rng(94), S=1500, n=3)
    mg = np.zeros(hg["D"], np.uint8)
    mg[:2] = 1
    mg[-1] = 1
    teacher_forced(demc, orc, hg, n_iter=3, n_groups=2, Np=6, schedule=2, burnin=1, theta_snooker=0.2, beta=0.2,
                   masks=np.stack([mg, 1 - mg]), exact_de=False)


def test_streaming_resident_form_with_blocks_and_trace(demc, orc):
    """the streaming-resident kernel under block updates, and teacher-forced against the oracle with the trace on"""
    prob = make_problem("mvn_full", np.random.default_rng(82), N=500, d=6)
    blocks = np.array([[1, 1, 0, 0, 0, 0], [0, 0, 1, 1, 1, 1]], np.uint8)
    th0 = prob["init"](4 * 24)
    a, b = (_run_fuse_mode(demc, prob, th0, fuse, 4, 24, 6, 20, blocks=blocks, alpha=0.2, burnin=8, loglike_mode=0) for fuse in (0, 1))
    for i, (x, y) in enumerate(zip(a, b)):
        if i in (2, 5):
            np.testing.assert_allclose(x, y, rtol=1e-10)
        else:
            assert np.array_equal(x, y), f"array {i}"
    teacher_forced(demc, orc, prob, n_iter=10, n_groups=4, Np=12, schedule=2, burnin=5, theta_snooker=0.2, alpha=0.5)


@pytest.mark.parametrize("family,Np,d,kw", [
    ("mvn_full", 256, 32, dict(loglike_mode=1)),                    # 512-thread workgroups, one pass per phase, MFMA prep
    ("mvn_full", 301, 16, dict(loglike_mode=1, theta_snooker=0.3)),  # odd Np: halves of 150 / 151, two passes
    ("mvn_iso", 64, 9, dict(loglike_mode=1, kappa=0.7)),            # odd D (no LDS-DMA), recombination
    ("gaussian", 10, 2, dict()),                                    # in-kernel observation loop, one lane per particle
    ("binomial", 33, 1, dict(beta=0.5)),                            # mutation in half of the groups
])
def test_resident_form_equals_one_launch_per_phase(demc, family, Np, d, kw):
    """fuse = 0 runs every iteration between two migrations inside ONE launch with the group resident in LDS; fuse = 2
    launches once per colour phase.  Same samples, acceptances and ids, across migrations (alpha = 0.3) and burn-in."""
    prob = make_problem(family, np.random.default_rng(61), N=200, d=d)
    D = prob["D"]
    th0 = prob["init"](5 * Np)
    a, b = (_run_fuse_mode(demc, prob, th0, fuse, 5, Np, D, 30, alpha=0.3, burnin=12, **kw) for fuse in (0, 2))
    # (theta_hist, accept_hist, lp_hist, id_hist, theta, weight, id).  The two forms may split a particle over a different
    # number of lanes, i.e. form its sums in a different order: log-densities agree to rounding, and so do snooker
    # proposals (their projection is such a sum, utilities.jl:239-246); acceptances, ids and crossover samples exactly.
    for i, (x, y) in enumerate(zip(a, b)):
        if i in (2, 5):
            np.testing.assert_allclose(x, y, rtol=1e-11)
        elif i in (0, 4) and kw.get("theta_snooker", 0.0) > 0.0:
            np.testing.assert_allclose(x, y, rtol=0, atol=1e-11)
        else:
            assert np.array_equal(x, y)


@pytest.mark.parametrize("fuse", [0, 2, 1])
def test_plain_instance_equals_general_instance(demc, fuse):
    """The default sampler (random_gamma, no snooker, kappa = 1, no blocks, Metropolis, no trace) runs in a K1 instance
    with every other branch compiled out; trace = 1 selects the general instance.  Same bits."""
    prob = make_problem("mvn_full", np.random.default_rng(63), N=300, d=12)
    th0 = prob["init"](8 * 40)
    a, b = (_run_fuse_mode(demc, prob, th0, fuse, 8, 40, 12, 25, alpha=0.3, burnin=10, loglike_mode=1, trace=tr) for tr in (0, 1))
    for i, (x, y) in enumerate(zip(a, b)):
        if i in (2, 5) and fuse == 0:
            # fuse = 0: the default sampler on MvNormal-full runs in its own lean resident kernel (demc_resmvn.hpp), whose lanes
            # own different scalars: the prior sums run in another order (log-densities to rounding, everything else exact)
            np.testing.assert_allclose(x, y, rtol=1e-12)
        else:
            assert np.array_equal(x, y), f"array {i}"


@pytest.mark.parametrize("G,Np,d,N,mode,kw", [
    (256, 256, 32, 2000, 1, dict()),                      # BASELINE cfg3 shape in SUFFSTAT mode: 512 threads, 128 particles per phase
    (32, 64, 8, 10000, 0, dict()),                        # BASELINE cfg2: STREAMING, 8 workgroups per group
    (5, 9, 7, 300, 1, dict(beta=0.5)),                    # odd everything, mutation in half of the sweeps
    (6, 30, 31, 400, 0, dict()),                          # D = 31: ragged last noise block, STREAMING
    (3, 20, 3, 100, 1, dict(lo=-0.5)),                    # D < 4: one lane per particle does all the work; a bound that bites
    (300, 8, 16, 200, 1, dict()),                         # more groups than CUs
])
def test_lean_resident_kernel_equals_the_general_kernel(demc, G, Np, d, N, mode, kw):
    """The default sampler on MvNormal-full (D = d <= 32, one pass per colour phase) has a lean resident kernel; trace = 1 selects
    the general k_propose instances.  Same proposals, acceptances, ids and samples across burn-in (base term), mutation sweeps
    and migrations; log-densities to rounding (prior sums in another lane order; STREAMING: same chunking)."""
    kw = dict(kw)
    prob = make_problem("mvn_full", np.random.default_rng(95), N=N, d=d)
    lo = kw.pop("lo", None)
    if lo is not None:
        prob = dict(prob, lo=[lo] * d)
    th0 = prob["init"](G * Np)
    if lo is not None:
        th0 = np.abs(th0)
    a, b = (_run_fuse_mode(demc, prob, th0, 0, G, Np, d, 16, alpha=0.3, burnin=8, loglike_mode=mode, trace=tr, **kw) for tr in (0, 1))
    for i, (x, y) in enumerate(zip(a, b)):
        if i in (2, 5):
            np.testing.assert_allclose(x, y, rtol=1e-11)
        else:
            assert np.array_equal(x, y), f"array {i}"


def test_resident_form_over_a_grid_of_shapes(demc):
    """boundary shapes of the resident kernel: smallest and odd group sizes, the 256 / 512-thread switch (moving half x lanes
    per particle around 256), odd and tiny D, one and several passes per phase -- each against the per-phase form"""
    rng = np.random.default_rng(64)
    checked = 0
    for Np in (4, 5, 7, 8, 9, 16, 17, 31, 33, 63, 64, 65, 127, 129):
        for d in (1, 2, 3, 5, 8, 9, 16, 17, 32, 33):
            prob = make_problem("mvn_full", rng, N=40, d=d)
            th0 = prob["init"](3 * Np)
            a, b = (_run_fuse_mode(demc, prob, th0, fuse, 3, Np, d, 6, alpha=0.4, burnin=3, loglike_mode=1) for fuse in (0, 2))
            for i, (x, y) in enumerate(zip(a, b)):
                if i in (2, 5):
                    np.testing.assert_allclose(x, y, rtol=1e-11, err_msg=f"Np={Np} d={d} array {i}")
                else:
                    assert np.array_equal(x, y), f"Np={Np} d={d} array {i}"
            checked += 1
    assert checked == 140


def test_resident_form_with_block_updates(demc):
    """block_update! (main.jl:174-179): several masked sweeps per iteration, all inside the resident launch"""
    prob = make_problem("mvn_full", np.random.default_rng(62), N=150, d=6)
    blocks = np.array([[1, 1, 0, 0, 0, 0], [0, 0, 1, 1, 1, 1]], np.uint8)
    th0 = prob["init"](4 * 24)
    a, b = (_run_fuse_mode(demc, prob, th0, fuse, 4, 24, 6, 20, blocks=blocks, alpha=0.2, burnin=8, loglike_mode=1) for fuse in (0, 2))
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_rejects_unsupported(demc):
    with pytest.raises(demc.DemcError):
        demc.HipEngine(n_groups=2, Np=8, D=2, schedule=3)  # unknown schedule
    with pytest.raises(demc.DemcError):
        demc.HipEngine(n_groups=2, Np=2, D=2, schedule=1)  # Np >= 3
    eng = demc.HipEngine(n_groups=2, Np=8, D=2, schedule=1)
    with pytest.raises(demc.DemcError):
        eng.set_model(99, np.zeros(3), [3])
    with pytest.raises(demc.DemcError):
        eng.step(1, 1)  # no model
    eng.close()


def test_lba_lnr_against_scipy_goldens(demc):
    """the device's table-driven Phi / phi (LBA) and log Phi(-z) (LNR) against the scipy goldens directly"""
    import os
    from demc_amd import families as F
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "logpdf_golden.npz"))
    for fam, key, extra in ((F.FAM_LBA, "lba", 3), (F.FAM_LNR, "lnr", 1)):
        c, rt, th = G[f"{key}_choice"], G[f"{key}_rt"], G[f"{key}_theta"]
        D = th.shape[1]
        eng = demc.HipEngine(n_groups=1, Np=max(4, th.shape[0]), D=D, schedule=1)
        eng.set_model(fam, np.concatenate([c, rt]), [c.size, D - extra], [1.0] if fam == F.FAM_LNR else None)
        eng.set_priors([F.PRIOR_FLAT] * D, [0.0] * D, [1.0] * D)
        np.testing.assert_allclose(eng.logpost(th), G[f"{key}_ll"], rtol=1e-9)
        eng.close()


@pytest.mark.parametrize("na,N", [(2, 1500), (3, 1100), (4, 1500), (6, 700), (2, 300)])
def test_lba_wave_kernel_every_accumulator_count(demc, orc, na, N):
    """k_lba_wave has compiled-in instances for two and three accumulators and a general one: the log-posterior of random
    proposals against the oracle for 2 / 3 / 4 / 6 accumulators, with whole batches + a ragged end (N = 1 500, 1 100), a ragged
    end only (700, 300) -- the trials arrive UNSORTED and demc_set_model sorts its own copy (the sum is compared, 1e-9)"""
    prob = make_problem("lba", np.random.default_rng(200 + na), N=N, na=na)
    eng, o = _pair(demc, orc, prob, n_groups=2, Np=16, schedule=1)
    th = prob["init"](32)
    lg, lo_ = eng.logpost(th), o.logpost(th)
    assert np.isfinite(lo_).sum() >= 24 and np.array_equal(np.isfinite(lg), np.isfinite(lo_))
    fin = np.isfinite(lo_)
    np.testing.assert_allclose(lg[fin], lo_[fin], rtol=1e-9)
    eng.close()


@pytest.mark.parametrize("kind,a,b", [(6, 2.5, 1.5), (7, 0.0, 0.7), (8, 0.3, 0.8), (9, -1.0, 2.0), (4, 2.0, 3.0), (3, 0.1, 5.0)])
def test_prior_kinds_match_oracle_and_goldens(demc, orc, kind, a, b):
    """every registered prior (Gamma, Exponential, LogNormal, Cauchy, Beta, Uniform) on the sigma slot of the Gaussian
    model: device log-posterior vs oracle, including points outside the prior's support (-Inf)"""
    prob = make_problem("gaussian", np.random.default_rng(71))
    prob = dict(prob, pk=[1, kind], pa=[0.0, a], pb=[10.0, b], lo=[-np.inf, -np.inf], hi=[np.inf, np.inf])
    eng, o = _pair(demc, orc, prob, n_groups=2, Np=8, schedule=1)
    th = prob["init"](16)
    th[:4, 1] = [0.05, 0.5, 0.999, 4.0]
    th[4, 1] = -0.3   # outside the support of the positive-only priors; sigma < 0 also makes the likelihood NaN/odd
    lg, lo_ = eng.logpost(th), o.logpost(th)
    fin = np.isfinite(lo_)
    assert fin.sum() >= 3
    assert np.array_equal(np.isfinite(lg), fin)
    np.testing.assert_allclose(lg[fin], lo_[fin], rtol=1e-10)
    eng.close()
