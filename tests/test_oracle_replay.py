"""Replay buffer on the CPU oracle: the reference's known-answer tests (test/utility_tests.jl) driven
through orc_step with caller-supplied draws -- i.e. through the code the oracle's sampler actually runs (propose(),
orc_migration_apply), in the reference's own schedule (sequential) and in the two device schedules."""
import numpy as np
import pytest

import replay_kats as K


@pytest.fixture()
def make(orc):
    def _make(**cfg):
        return orc.Oracle(**{k: v for k, v in cfg.items() if k in orc.CFG_KEYS})
    return _make


def _migrate(e, it):
    rows = e.migration_pack(it)
    e.migration_apply(it, rows)


@pytest.mark.parametrize("schedule", [0, 1, 2])
@pytest.mark.parametrize("kat", K.ALL_STEP_KATS, ids=lambda f: f.__name__)
def test_reference_kats_through_the_oracle_step(make, kat, schedule):
    kat(make, schedule)


@pytest.mark.parametrize("schedule", [0, 1, 2])
def test_migration_kat_through_the_oracle(make, schedule):
    K.migration_circular_shift(make, _migrate, schedule)   # migration! alone (orc_migration_pack / _apply)
    K.migration_circular_shift(make, None, schedule)       # through orc_step: replayed alpha coin, no-op update


def test_replayed_alpha_coin_and_clear(make):
    e = make(n_groups=3, Np=4, D=2, n_rows=3, alpha=0.5, seed=9, schedule=2)
    e.set_model(K.FAM_MVN_FULL, np.zeros((3, 2)), [3, 2], np.eye(2))
    e.set_priors([1, 1], [0, 0], [1, 1])
    e.set_bounds([-np.inf] * 2, [np.inf] * 2)
    e.set_state(np.random.default_rng(0).normal(0, 1, (12, 2)))
    ids0 = e.get_state()[2].copy()
    e.set_replay(u_step=[0.9])            # coin does not fire: ids stay where they are
    e.step(1, 1)
    assert np.array_equal(e.get_state()[2], ids0)
    e.set_replay(u_step=[0.1], mig_groups=[0, 2], mig_particle=[1, -1, 3])   # fires: groups 0 and 2 swap their picks
    e.step(2, 1)
    ids1 = e.get_state()[2]
    assert ids1[0 * 4 + 1] == ids0[2 * 4 + 3] and ids1[2 * 4 + 3] == ids0[0 * 4 + 1]
    e.set_replay()                        # back to Philox
    e.step(3, 1)
    e.close()


def test_plus_inf_weight_is_probability_zero_not_the_fallback(make):
    """migration.jl:64-70: exp(-Inf) = 0 is an ordinary weight; only a NaN in exp.(-w)/sum (w = -Inf, NaN, or every
    weight +Inf) triggers findmin.  (evaluate_fun!/minimize! leave +Inf on out-of-bounds particles.)"""
    e = make(n_groups=2, Np=4, D=2, n_rows=1, alpha=1.0, seed=3, schedule=2)
    e.set_model(K.FAM_MVN_FULL, np.zeros((3, 2)), [3, 2], np.eye(2))
    th = np.random.default_rng(1).normal(0, 1, (8, 2))
    w = np.array([np.inf, -3.0, -3.0, -3.0, np.inf, np.inf, np.inf, np.inf])
    e.set_state(th, w, np.arange(8))
    picks = set()
    for it in range(1, 60):
        rows = e.migration_pack(it)
        picks.add(int(rows[0, 0]))
        assert int(rows[1, 0]) == 0          # all +Inf: findmin -> first index
    assert picks == {1, 2, 3}                # the +Inf particle has probability 0; the finite ones are all reachable
    w2 = np.array([-np.inf, -3.0, -3.0, -5.0, 0, 0, 0, 0.0])
    e.set_state(th, w2, np.arange(8))
    assert int(e.migration_pack(1)[0, 0]) == 0   # -Inf -> NaN in the reference -> findmin = index of -Inf
    e.close()
