"""The reference's deterministic unit tests (test/utility_tests.jl) driven THROUGH the sampler step with replayed draws.

Each function takes `make(**cfg)` -> an engine (the CPU oracle, or the HIP engine: same method names) and runs one
step in which every random choice that matters is supplied by the test (set_replay), so that the proposal a particle
receives is the reference's known answer.  tests/test_oracle_replay.py runs them on the oracle (CPU),
tests/test_gpu_replay.py on the MI355X through the C-ABI.
"""
import numpy as np

FAM_MVN_FULL = 2
NAN = float("nan")


def _engine(make, rows, **cfg):
    """one group holding `rows` ([Np][2]); flat prior, unbounded, a 2-D unit-covariance Gaussian likelihood"""
    rows = np.asarray(rows, dtype=np.float64)
    Np = rows.shape[0]
    base = dict(n_groups=1, Np=Np, D=2, n_rows=2, burnin=0, eps=0.0, alpha=0.0, beta=0.0, seed=77, trace=1)
    base.update(cfg)
    e = make(**base)
    e.set_model(FAM_MVN_FULL, np.array([[0.1, -0.2], [0.3, 0.4], [-0.5, 0.2]]), [3, 2], np.eye(2))
    e.set_priors([0, 0], [0.0, 0.0], [1.0, 1.0])
    e.set_bounds([-np.inf] * 2, [np.inf] * 2)
    e.set_state(rows)
    return e


def projection_through_snooker(make, schedule):
    """test/utility_tests.jl:76-92: project([-1,4] onto [2,7]) = [52/53, 182/53].  snooker_update! (crossover.jl:239-257)
    for particle 0 with Pz = row 3, Pm = row 4, Pn = row 5 (rows of the other half, legal in every schedule):
    Pd = Pt - Pz = [2,7], Pr1 = project(Pm, Pd), Pr2 = project([0,0], Pd) = 0, eps = 0
    =>  (proposal - Pt)/gamma = [52/53, 182/53]."""
    rows = [[3.0, 9.0], [0.5, 0.5], [0.25, -0.5], [1.0, 2.0], [-1.0, 4.0], [0.0, 0.0]]
    e = _engine(make, rows, schedule=schedule, theta_snooker=1.0)
    u_g = 0.75
    part = np.full((len(rows), 5), NAN)
    part[:, 0] = 0.0   # snooker coin fires (crossover.jl:31)
    part[:, 2] = u_g   # gamma = rand(Uniform(1.2, 2.2)) (crossover.jl:249)
    partner = np.full((len(rows), 3), -1, np.int64)
    partner[0] = [3, 4, 5]
    e.set_replay(u_part=part, partner=partner, u_group=[1.0])
    e.step(1, 1)
    tr = e.get_trace()
    assert tr["idx"][0].tolist() == [1, 3, 4, 5]
    gamma = 1.2 + (2.2 - 1.2) * u_g
    got = (tr["proposal"][0] - np.array(rows[0])) / gamma
    np.testing.assert_allclose(got, [52 / 53, 182 / 53], rtol=4e-15)
    e.close()
    return got


def particle_algebra_through_crossover(make, schedule):
    """test/utility_tests.jl:165-198: 3*(p1 - p2) = [9,-3] and 3*(p1 - p2) + p3 = [7,0] with p1 = [1,2], p2 = p3 = [-2,3].
    random_gamma (crossover.jl:154-172) past burn-in with eps = 0: proposal = Pt + gamma_1*(Pm - Pn); the replayed
    gamma uniform 5.0 gives gamma_1 = 0.5 + 0.5*5 = 3 exactly (test mode does not clamp uniforms)."""
    rows = [[0.0, 0.0], [-2.0, 3.0], [4.0, 4.0], [1.0, 2.0], [-2.0, 3.0], [5.0, 5.0]]
    e = _engine(make, rows, schedule=schedule)
    part = np.full((len(rows), 5), NAN)
    part[:, 0] = 1.0   # no snooker
    part[:, 2] = 5.0   # gamma_1 = 3
    partner = np.full((len(rows), 3), -1, np.int64)
    partner[0] = [3, 4, -1]  # Pt = [0,0]:  3*(p1 - p2)
    partner[1] = [3, 4, -1]  # Pt = p3:     3*(p1 - p2) + p3
    e.set_replay(u_part=part, partner=partner, u_group=[1.0])
    e.step(1, 1)
    tr = e.get_trace()
    assert tr["idx"][0].tolist() == [0, 3, 4, -1] and tr["idx"][1].tolist() == [0, 3, 4, -1]
    assert tr["proposal"][0].tolist() == [9.0, -3.0]
    assert tr["proposal"][1].tolist() == [7.0, 0.0]
    e.close()


def base_term_through_crossover(make, schedule):
    """crossover.jl:164-168 inside burn-in: ((Pt + g1*(Pm-Pn)) + g2*(Pb-Pt)) + b with replayed Pb, g1 = 1, g2 = 0.5:
    Pt = [2,2], Pm - Pn = [3,-1], Pb = [4,0]  =>  [2,2] + [3,-1] + 0.5*[2,-2] = [6,0]"""
    rows = [[2.0, 2.0], [7.0, 7.0], [8.0, 8.0], [1.0, 2.0], [-2.0, 3.0], [4.0, 0.0]]
    e = _engine(make, rows, schedule=schedule, burnin=10)
    part = np.full((len(rows), 5), NAN)
    part[:, 0] = 1.0
    part[:, 2] = 1.0   # g1 = 0.5 + 0.5*1 = 1
    part[:, 3] = 0.0   # g2 = 0.5
    partner = np.full((len(rows), 3), -1, np.int64)
    partner[0] = [3, 4, 5]
    e.set_replay(u_part=part, partner=partner, u_group=[1.0])
    e.step(1, 1)
    tr = e.get_trace()
    assert tr["idx"][0].tolist() == [0, 3, 4, 5]
    assert tr["proposal"][0].tolist() == [6.0, 0.0]
    e.close()


def uniform_noise_through_crossover(make, schedule):
    """test/utility_tests.jl:193-198: p + Uniform(-0.1, 0.1) stays within 0.1 of p and differs from it.  Identical
    particles make gamma*(Pm - Pn) exactly zero, so proposal - Pt = b = -eps + 2 eps u for the replayed u."""
    rows = np.tile([1.0, 2.0], (6, 1))
    e = _engine(make, rows, schedule=schedule, eps=0.1)
    u = np.linspace(0.05, 0.95, 12).reshape(6, 2)
    part = np.full((6, 5), NAN)
    part[:, 0] = 1.0
    # Pm and Pn replayed as the SAME row: the difference stays exactly zero even after earlier particles of the
    # sweep have moved (sequential schedule, second colour phase)
    partner = np.array([[5, 5, -1]] * 3 + [[0, 0, -1]] * 3, np.int64)
    e.set_replay(u_part=part, u_noise=u, u_group=[1.0], partner=partner)
    e.step(1, 1)
    prop = e.get_trace()["proposal"]
    b = -0.1 + (0.1 - (-0.1)) * u
    # (a particle's own row is untouched until its own turn, so Pt is the initial row in every schedule)
    assert np.array_equal(prop, rows + b)
    assert np.all(np.abs(prop - rows) <= 0.1) and np.all(prop != rows)
    e.close()


def reset_mask_through_block_sweep(make, schedule):
    """test/utility_tests.jl:46-68: reset! keeps the previous value where the block mask is false.  One block
    [true, false]: the proposal moves in scalar 0 only."""
    rows = [[0.0, 0.0], [-2.0, 3.0], [4.0, 4.0], [1.0, 2.0], [-2.0, 3.0], [5.0, 5.0]]
    e = _engine(make, rows, schedule=schedule, n_blocks=1)
    e.set_blocks(np.array([[1, 0]], np.uint8))
    part = np.full((6, 5), NAN)
    part[:, 0] = 1.0
    part[:, 2] = 5.0
    partner = np.full((6, 3), -1, np.int64)
    partner[0] = [3, 4, -1]
    e.set_replay(u_part=part, partner=partner, u_group=[1.0])
    e.step(1, 1)
    tr = e.get_trace()
    assert tr["proposal"][0].tolist() == [9.0, 0.0]   # scalar 1 reset to the previous value
    e.close()


def mutation_with_replayed_normals(make, schedule):
    """mutation! (mutation.jl:13-25): proposal = theta + Normal(0, sigma) per scalar, the normals supplied"""
    rows = np.arange(12.0).reshape(6, 2)
    e = _engine(make, rows, schedule=schedule, beta=1.0, sigma=0.5)
    z = np.linspace(-2, 2, 12).reshape(6, 2)
    e.set_replay(z_noise=z, u_group=[0.0])
    e.step(1, 1)
    tr = e.get_trace()
    assert (tr["idx"][:, 0] == 2).all()
    assert np.array_equal(tr["proposal"], rows + 0.5 * z)
    e.close()


def accept_uniform_decides(make, schedule):
    """accept (utilities.jl:55-58): rand() <= min(1, exp(w' - w)).  A replayed uniform of 0 accepts every finite
    proposal; a uniform above 1 rejects every proposal that is not an improvement."""
    rng = np.random.default_rng(5)
    rows = rng.normal(0, 1, (6, 2))
    for u_acc in (0.0, 1.5):
        e = _engine(make, rows, schedule=schedule, eps=0.01)
        part = np.full((6, 5), NAN)
        part[:, 0] = 1.0
        part[:, 4] = u_acc
        e.set_replay(u_part=part, u_group=[1.0])
        _, w0, _ = e.get_state()
        e.step(1, 1)
        tr = e.get_trace()
        if u_acc == 0.0:
            assert tr["accepted"].all()
        else:  # a particle's own weight is untouched until its own turn, in every schedule
            assert np.array_equal(tr["accepted"].astype(bool), tr["w_prop"] >= w0)
        e.close()


def migration_circular_shift(make, migrate, schedule):
    """test/utility_tests.jl:118-160: after shift_particles! the i-th group of the sub-group holds, at its picked slot,
    the particle the (i-1)-th group had picked (circshift(., 1)), and -- the sub-group aliasing `groups` -- the
    change is visible in the population (ridx = [2,4] there: 0-based groups 1 and 3 lead the sub-group here).
    select_groups' sub-group and select_particles' picks are replayed.  `migrate(engine, iter)` runs migration! alone;
    migrate = None goes through the whole step!: the replayed alpha coin fires (main.jl:85) and the update that
    follows is made a no-op (eps = 0 and Pm = Pn replayed as one row: every proposal equals its particle)."""
    G, Np, D = 5, 4, 2
    rng = np.random.default_rng(6)
    th0 = rng.normal(0, 1, (G * Np, D))
    e = make(n_groups=G, Np=Np, D=D, n_rows=2, burnin=0, alpha=0.5, eps=0.0, seed=5, schedule=schedule, trace=1)
    e.set_model(FAM_MVN_FULL, np.array([[0.1, -0.2], [0.3, 0.4], [-0.5, 0.2]]), [3, 2], np.eye(2))
    e.set_priors([1, 1], [0.0, 0.0], [1.0, 1.0])
    e.set_bounds([-np.inf] * 2, [np.inf] * 2)
    e.set_state(th0)
    _, w0, id0 = e.get_state()
    sub = [1, 3, 0]                 # ordered sub-group (global group indices)
    p_idx = {1: 2, 3: 0, 0: 3}      # select_particles' pick inside each selected group
    pick = np.full(G, -1, np.int64)
    for g, j in p_idx.items():
        pick[g] = j
    if migrate is not None:
        e.set_replay(mig_groups=sub, mig_particle=pick)
        migrate(e, 1)
    else:
        part = np.full((G * Np, 5), NAN)
        part[:, 0] = 1.0            # no snooker
        partner = np.tile(np.array([[3, 3, -1], [3, 3, -1], [0, 0, -1], [0, 0, -1]], np.int64), (G, 1))
        e.set_replay(u_step=[0.0], mig_groups=sub, mig_particle=pick, u_group=np.ones(G), u_part=part, partner=partner)
        e.step(1, 1)
    th1, w1, id1 = e.get_state()
    for i, g in enumerate(sub):
        gp = sub[(i - 1) % len(sub)]
        dst, src = g * Np + p_idx[g], gp * Np + p_idx[gp]
        assert np.array_equal(th1[dst], th0[src]) and id1[dst] == id0[src]
        np.testing.assert_allclose(w1[dst], w0[src], rtol=1e-12)
    moved = [g * Np + p_idx[g] for g in sub]
    rest = np.setdiff1d(np.arange(G * Np), moved)
    assert np.array_equal(th1[rest], th0[rest]) and np.array_equal(id1[rest], id0[rest])
    assert sorted(id1.tolist()) == sorted(id0.tolist())
    e.close()


ALL_STEP_KATS = [projection_through_snooker, particle_algebra_through_crossover, base_term_through_crossover,
                 uniform_noise_through_crossover, reset_mask_through_block_sweep, mutation_with_replayed_normals,
                 accept_uniform_decides]
