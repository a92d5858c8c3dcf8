"""The reference's own deterministic unit tests (test/utility_tests.jl) restated against the CPU oracle: these are
the known-answer vectors that pin the oracle's particle algebra, projection, block reset and migration shift."""
import ctypes as C

import numpy as np

dp = C.POINTER(C.c_double)
bp = C.POINTER(C.c_uint8)


def _p(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _axpby(L, x, y, a, b):
    x, y = _p(x), _p(y)
    out = np.empty_like(x)
    L.orc_axpby(x.ctypes.data_as(dp), y.ctypes.data_as(dp), a, b, x.size, out.ctypes.data_as(dp))
    return out


def test_philox_known_answers(orc):
    """Random123 kat_vectors for philox4x32-10"""
    assert orc.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert orc.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert orc.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    L = orc.lib()
    assert L.orc_u53(0, 0) == 0.0
    assert L.orc_u53(0xffffffff, 0xffffffff) == 1.0 - 2.0 ** -53


def test_projection(orc):
    """test/utility_tests.jl:76-92: project([-1,4] onto [2,7]) == [52/53, 182/53] (flat and nested Theta)"""
    x1, x2, out = _p([-1.0, 4.0]), _p([2.0, 7.0]), np.empty(2)
    orc.lib().orc_project(x1.ctypes.data_as(dp), x2.ctypes.data_as(dp), 2, out.ctypes.data_as(dp))
    np.testing.assert_allclose(out, [52 / 53, 182 / 53], rtol=1e-15)
    np.testing.assert_allclose(out, (x1 @ x2) / (x2 @ x2) * x2, rtol=1e-15)


def test_reset_vector_and_matrix_masks(orc):
    """test/utility_tests.jl:46-68 with nested Theta flattened column-major (Julia order)"""
    L = orc.lib()
    p1 = _p([0.7, 0.5, 0.1, 0.4, 0.6])
    p2 = _p([0.9, 0.8, 0.5, 0.7, 0.8])
    m = np.array([1, 0, 0, 0, 1], np.uint8)
    L.orc_reset(p1.ctypes.data_as(dp), p2.ctypes.data_as(dp), m.ctypes.data_as(bp), 5)
    assert p1[0] != p2[0] and p1[1] == p2[1] and p1[2] == p2[2] and p1[3] == p2[3] and p1[4] != p2[4]
    # matrix parameter [0.7 0.5; 0.1 0.3] -> column-major (0.7, 0.1, 0.5, 0.3); mask [true false; false true]
    p1 = _p([0.7, 0.1, 0.5, 0.3, 0.4, 0.6])
    p2 = _p([0.9, 0.5, 0.8, 0.2, 0.7, 0.8])
    m = np.array([1, 0, 0, 1, 0, 1], np.uint8)
    L.orc_reset(p1.ctypes.data_as(dp), p2.ctypes.data_as(dp), m.ctypes.data_as(bp), 6)
    assert p1[0] != p2[0] and p1[1] == p2[1] and p1[2] == p2[2] and p1[3] != p2[3] and p1[4] == p2[4] and p1[5] != p2[5]


def test_particle_operations(orc):
    """test/utility_tests.jl:165-198"""
    L = orc.lib()
    one = np.ones(2)
    np.testing.assert_allclose(_axpby(L, [1, 2], one, 1, 2), [3, 4])            # p + 2
    np.testing.assert_allclose(_axpby(L, [1, 2], one, 4, 0), [4, 8])            # p * 4
    np.testing.assert_allclose(_axpby(L, [1, 2], [1, 2], 1, 1), [2, 4])         # p1 + p2
    np.testing.assert_allclose(_axpby(L, [1, 2], [1, 2], 3, 3), [6, 12])        # 3 * (p1 + p2)
    d = _axpby(L, [1, 2], [-2, 3], 3, -3)
    np.testing.assert_allclose(d, [9, -3])                                     # 3 * (p1 - p2)
    np.testing.assert_allclose(_axpby(L, d, [-2, 3], 1, 1), [7, 0])             # 3 * (p1 - p2) + p3


def test_uniform_noise_is_bounded_and_nonzero(orc):
    """p + Uniform(-.1,.1): within 0.2 and different from p (test/utility_tests.jl:193-198), through the sampler:
    a crossover proposal with gamma*(Pm-Pn) = 0 is Pt + b"""
    o = orc.Oracle(n_groups=1, Np=4, D=2, eps=0.1, beta=0.0, alpha=0.0, burnin=0, n_rows=1, schedule=1, seed=3)
    o.set_model(0, np.zeros(5), [5])
    th = np.tile([1.0, 2.0], (4, 1))  # identical particles -> difference vector is exactly zero
    o.set_state(th)
    o.step(1, 1)
    prop = o.get_trace()["proposal"]
    assert np.all(np.abs(prop - th) <= 0.1) and np.all(prop != th)


def test_shift_particles_is_a_circular_shift(orc):
    """test/utility_tests.jl:149-154: selected group i receives the particle of selected group i-1"""
    rng = np.random.default_rng(0)
    c = rng.normal(size=(5, 3))
    x = c.copy()
    orc.lib().orc_shift_particles(x.ctypes.data_as(dp), 5, 3)
    np.testing.assert_array_equal(x, np.roll(c, 1, axis=0))


def test_migration_moves_whole_particles(orc):
    """migration.jl:84-91 through the sampler: theta, weight and id travel together; weights stay consistent"""
    rng = np.random.default_rng(1)
    o = orc.Oracle(n_groups=5, Np=4, D=2, alpha=1.0, n_rows=4, schedule=1, seed=11)
    o.set_model(0, rng.normal(size=30), [30])
    o.set_priors([1, 2], [0, 0], [10, 1])
    o.set_bounds([-np.inf, 0], [np.inf, np.inf])
    th0 = np.stack([rng.normal(size=20), rng.uniform(0.5, 2, 20)], 1)
    o.set_state(th0)
    _, w0, id0 = o.get_state()
    assert o.migration_due(1)
    sel = o.migration_plan(1)
    assert 2 <= sel.size <= 5 and len(set(sel.tolist())) == sel.size
    rows = o.migration_pack(1)
    o.migration_apply(1, rows)
    th1, w1, id1 = o.get_state()
    assert sorted(id1.tolist()) == list(range(20))  # a permutation: nothing lost, nothing duplicated
    for i, gd in enumerate(sel):
        gs = sel[(i - 1) % sel.size]
        sd, ss = gd * 4 + int(rows[gd, 0]), gs * 4 + int(rows[gs, 0])
        np.testing.assert_array_equal(th1[sd], th0[ss])
        assert w1[sd] == w0[ss] and id1[sd] == id0[ss]
    untouched = np.setdiff1d(np.arange(20), [g * 4 + int(rows[g, 0]) for g in sel])
    np.testing.assert_array_equal(th1[untouched], th0[untouched])


def test_adjust_loglike_forms_agree_and_faithful_form_overflows(orc):
    """crossover.jl:268-273: the stable form equals the written one where the latter is finite"""
    rng = np.random.default_rng(2)
    L = orc.lib()
    for D in (2, 5, 31):
        a, b, z = (np.ascontiguousarray(rng.normal(size=D)) for _ in range(3))
        args = (a.ctypes.data_as(dp), b.ctypes.data_as(dp), z.ctypes.data_as(dp), D)
        np.testing.assert_allclose(L.orc_adjust_loglike(*args, 0), L.orc_adjust_loglike(*args, 1), rtol=1e-10, atol=1e-12)
    D = 10002
    a, b, z = (np.ascontiguousarray(rng.normal(size=D) * 30) for _ in range(3))
    args = (a.ctypes.data_as(dp), b.ctypes.data_as(dp), z.ctypes.data_as(dp), D)
    assert np.isfinite(L.orc_adjust_loglike(*args, 0)) and not np.isfinite(L.orc_adjust_loglike(*args, 1))


def test_reference_softmax_quirks_are_reproduced(orc):
    """select_base / select_particle as written (crossover.jl:282-289, migration.jl:64-70): un-stabilised exp
    under/overflows at scale; the quirk-faithful helpers document what the reference then does"""
    L = orc.lib()
    w = np.ascontiguousarray([-1.0, -2.0, -0.5])
    assert L.orc_select_particle_ref(w.ctypes.data_as(dp), 3, 0.0) == 0
    big = np.ascontiguousarray([-4.5e6, -4.5e6 - 3, -4.5e6 + 2])
    assert L.orc_select_particle_ref(big.ctypes.data_as(dp), 3, 0.99) == 1  # exp(-w) = Inf -> NaN -> findmin(w)
    assert L.orc_select_base_ref(w.ctypes.data_as(dp), 3, 0.999999) == 2
