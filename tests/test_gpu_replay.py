"""Replay buffer (demc_set_replay) on the MI355X, through the C-ABI:
  * the reference's known-answer tests (test/utility_tests.jl) driven through demc_step with caller-supplied draws --
    the same functions tests/test_oracle_replay.py runs on the CPU oracle;
  * with RANDOM caller-supplied draws the HIP path and the oracle make the same choices bit for bit (what a Julia host
    feeding its own RNG stream would rely on), in the reference's sequential schedule and the two device schedules."""
import numpy as np
import pytest

import replay_kats as K
from conftest import make_problem, setup_engine

pytestmark = pytest.mark.gpu


@pytest.fixture()
def make(demc):
    def _make(**cfg):
        return demc.HipEngine(**cfg)
    return _make


def _migrate(e, it):
    e.migration_pack_dev(it, None)    # NULL: the handle's own staging rows (single shard)
    e.migration_apply_dev(it, None)


@pytest.mark.parametrize("schedule", [0, 1, 2])
@pytest.mark.parametrize("kat", K.ALL_STEP_KATS, ids=lambda f: f.__name__)
def test_reference_kats_through_the_device_step(make, kat, schedule):
    kat(make, schedule)


@pytest.mark.parametrize("fuse", [0, 1, 2])
def test_kats_do_not_depend_on_the_fuse_mode(demc, fuse):
    def mk(**cfg):
        return demc.HipEngine(fuse=fuse, **cfg)
    for kat in (K.projection_through_snooker, K.particle_algebra_through_crossover, K.base_term_through_crossover):
        kat(mk, 2)


@pytest.mark.parametrize("schedule", [0, 1, 2])
def test_migration_kat_on_the_device(make, schedule):
    K.migration_circular_shift(make, _migrate, schedule)   # k_mig_pack + k_mig_apply alone
    K.migration_circular_shift(make, None, schedule)       # through demc_step: replayed alpha coin, no-op update


def _random_replay(rng, G, Np, D, schedule):
    P = G * Np
    half = Np // 2
    part = rng.uniform(0, 1, (P, 5))
    partner = np.empty((P, 3), np.int64)
    for s in range(P):
        pl = s % Np
        if schedule == 2:  # two_colour: the resting half
            pool = np.arange(half, Np) if pl < half else np.arange(0, half)
        else:
            pool = np.setdiff1d(np.arange(Np), [pl])
        partner[s] = rng.choice(pool, 3, replace=False)
    return dict(u_group=rng.uniform(0, 1, G), u_part=part, partner=partner, u_noise=rng.uniform(0, 1, (P, D)),
                z_noise=rng.normal(0, 1, (P, D)), u_recomb=rng.uniform(0, 1, (P, D)))


@pytest.mark.parametrize("schedule", [0, 1, 2])
@pytest.mark.parametrize("family", ["mvn_iso", "gaussian", "lnr"])
def test_random_replayed_draws_give_identical_choices(demc, orc, family, schedule):
    """every draw of every sweep supplied by the caller: branch / partner indices and accept decisions identical, crossover
    proposals bit-exact, mutation proposals bit-exact too (the normals are given, no device log/sincos involved)"""
    rng = np.random.default_rng(100 + schedule)
    prob = make_problem(family, rng, d=5)
    G, Np, D = 3, 8, prob["D"]
    cfg = dict(n_groups=G, Np=Np, D=D, n_rows=8, schedule=schedule, burnin=4, theta_snooker=0.3, kappa=0.8, beta=0.3,
               alpha=0.0, seed=21)
    eng = demc.HipEngine(trace=1, **cfg)
    o = orc.Oracle(**{k: v for k, v in cfg.items() if k in orc.CFG_KEYS})
    setup_engine(eng, prob)
    setup_engine(o, prob)
    eng.set_state(prob["init"](G * Np))
    for it in range(1, 9):
        rp = _random_replay(rng, G, Np, D, schedule)
        eng.set_replay(**rp)
        o.set_replay(**rp)
        th, w, ids = eng.get_state()
        o.set_state(th, w, ids)
        eng.step(it, 1)
        o.step(it, 1)
        tg, to = eng.get_trace(), o.get_trace()
        assert np.array_equal(tg["idx"], to["idx"])
        kind = to["idx"][:, 0]
        # what the caller asked for is what ran
        mut = np.repeat(rp["u_group"] <= 0.3, Np)
        assert np.array_equal(kind == 2, mut)
        snk = (~mut) & (rp["u_part"][:, 0] <= 0.3)
        assert np.array_equal(kind == 1, snk)
        assert np.array_equal(to["idx"][snk, 1:], rp["partner"][snk])
        assert np.array_equal(to["idx"][kind == 0, 1:3], rp["partner"][kind == 0, :2])
        exact = kind != 1
        if schedule != 1 and (kind == 1).any():
            exact[:] = False   # later particles read rows accepted from snooker proposals (1-ulp projection sums)
        assert np.array_equal(tg["proposal"][exact], to["proposal"][exact])
        np.testing.assert_allclose(tg["proposal"], to["proposal"], rtol=1e-11, atol=1e-13)
        fin = np.isfinite(to["w_prop"])
        assert np.array_equal(np.isfinite(tg["w_prop"]), fin)
        np.testing.assert_allclose(tg["w_prop"][fin], to["w_prop"][fin], rtol=1e-9)
        assert np.array_equal(tg["accepted"], to["accepted"])
    eng.set_replay()
    o.set_replay()
    eng.close()
    o.close()


def test_replay_is_validated(demc):
    e = demc.HipEngine(n_groups=2, Np=6, D=2, schedule=1)
    bad = np.full((12, 3), -1, np.int64)
    bad[3, 1] = 6
    with pytest.raises(demc.DemcError):
        e.set_replay(partner=bad)
    with pytest.raises(demc.DemcError):
        e.set_replay(mig_groups=[0, 0])
    with pytest.raises(demc.DemcError):
        e.set_replay(mig_groups=[0, 2])
    e.close()


def test_plus_inf_weights_in_migration_match_the_oracle(demc, orc):
    """select_particle with +Inf weights (evaluate_fun!/minimize! leave +Inf on out-of-bounds particles): exp(-Inf) = 0 is an
    ordinary weight, only -Inf / NaN / all-+Inf fall back to findmin (migration.jl:64-70)"""
    G, Np, D = 4, 5, 2
    rng = np.random.default_rng(3)
    th = rng.normal(0, 1, (G * Np, D))
    w = rng.normal(-5, 2, G * Np)
    w[0] = np.inf                      # group 0: one +Inf
    w[5:10] = np.inf                   # group 1: all +Inf
    w[12] = -np.inf                    # group 2: a -Inf
    cfg = dict(n_groups=G, Np=Np, D=D, n_rows=1, alpha=1.0, seed=8, schedule=2)
    eng = demc.HipEngine(**cfg)
    o = orc.Oracle(**{k: v for k, v in cfg.items() if k in orc.CFG_KEYS})
    for e in (eng, o):
        e.set_model(K.FAM_MVN_FULL, np.zeros((3, 2)), [3, 2], np.eye(2))
    for it in range(1, 40):
        eng.set_state(th, w, np.arange(G * Np))
        o.set_state(th, w, np.arange(G * Np))
        _migrate(eng, it)
        rows = o.migration_pack(it)
        o.migration_apply(it, rows)
        assert int(rows[0, 0]) != 0 and int(rows[1, 0]) == 0 and int(rows[2, 0]) == 2
        for a, b in zip(eng.get_state(), o.get_state()):
            assert np.array_equal(a, b)
    eng.close()
    o.close()
