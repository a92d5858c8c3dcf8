"""The kernel instances that SHIP -- the ones `sample()`, bench.py and the Julia shim run (trace = 0) -- against the CPU
oracle DIRECTLY, free-running: both engines start from the same rows, evaluate their own weights and run the iterations on
their own (no teacher forcing, no trace); afterwards the whole history is compared -- the id in every slot and every accept
flag bit for bit, theta bit for bit where the run holds crossover moves only (the proposal is a fixed +,-,* sequence,
crossover.jl:154-172) and to 1e-10 where mutation's device log/sincos enter (mutation.jl:13-25), log-posteriors to 1e-9
(north-star bar 1e-6; utilities.jl:201-210).  Every test asserts, through demc_last_kernels, WHICH instance it compared.

The teacher-forced tests of test_gpu_parity.py read the per-slot trace and therefore run the general k_propose instance;
these are the production twins: PLAIN K1 + k_cross_mfma + K3, k_res_mvn in its three shapes, the long-row span loops, the
thread-per-proposal LBA kernel."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def free_run(demc, orc, w, n_it, kernels, G, Np, theta_exact, lp_rtol=1e-9, theta_rtol=1e-10, exact_kernels=None, **cfg):
    from demc_amd import workloads as W
    P = G * Np
    base = dict(n_groups=G, Np=Np, D=w["D"], n_rows=n_it, schedule=2, seed=4242, burnin=n_it // 2, trace=0)
    base.update(w["engine"])
    base.update(cfg)
    eng = demc.HipEngine(**base)
    o = orc.Oracle(n_threads=8, **{k: v for k, v in base.items() if k in orc.CFG_KEYS})
    for e in (eng, o):
        W.configure(e, w)
    rng = np.random.default_rng(5)
    n_init = int(base.get("n_initial", 0))
    if n_init:  # initialize_samples (utilities.jl:35-39): rows 1:n_initial of the history are independent prior draws
        rows0 = np.stack([w["init"](P, rng) for _ in range(n_init)])
        eng.set_history_rows(0, rows0)
        o.set_history_rows(0, rows0)
    th0 = w["init"](P, rng)
    eng.set_state(th0)   # each engine evaluates its own starting weights
    o.set_state(th0)
    np.testing.assert_allclose(eng.get_state()[1], o.get_state()[1], rtol=lp_rtol)
    n_run = n_it - n_init
    eng.step(1 + n_init, n_run)
    o.step(1 + n_init, n_run)
    ran = eng.last_kernels()
    for name in kernels:
        assert name in ran, f"expected {name}, the engine ran {ran}"
    if exact_kernels is not None:
        assert ran == exact_kernels, f"expected exactly {exact_kernels}, the engine ran {ran}"
    hg, ho = eng.get_history(n_init, n_it), o.get_history(n_init, n_it)
    assert np.array_equal(hg[3], ho[3]), "particle ids per slot differ (migration bookkeeping)"
    n_flip = int((hg[1] != ho[1]).sum())
    assert n_flip == 0, f"{n_flip} accept decisions differ"
    assert hg[1].mean() > 0.02, "nothing was accepted: the comparison would be vacuous"
    if theta_exact:
        assert np.array_equal(hg[0], ho[0]), "theta history is not bit-exact"
    else:
        np.testing.assert_allclose(hg[0], ho[0], rtol=theta_rtol, atol=1e-13)
    np.testing.assert_allclose(hg[2], ho[2], rtol=lp_rtol)
    sg, so = eng.get_state(), o.get_state()
    assert np.array_equal(sg[2], so[2])
    np.testing.assert_allclose(sg[1], so[1], rtol=lp_rtol)
    eng.close()
    o.close()
    return ran


@pytest.mark.parametrize("beta", [0.0, 0.1])
def test_cfg2_shape_streaming_lean_kernel(demc, orc, beta):
    """BASELINE cfg2 as it is benchmarked: 32 x 64, D = 8, N = 1e4, STREAMING -> k_res_mvn<256,true,8> (the observation
    stream inside the resident kernel, 8 workgroups per group handing their cross terms over)"""
    from demc_amd import workloads as W
    w = W.cfg2()
    free_run(demc, orc, w, 10, ["k_res_mvn<256,true,8>"], w["G"], w["Np"], theta_exact=beta == 0.0, beta=beta, loglike_mode=0)


@pytest.mark.parametrize("N", [2050, 2048, 1030])
def test_cfg2_tile_loop_tails(demc, orc, N):
    """the observation stage of k_res_mvn<256,true,8> is one asm statement (cross_loop_lds_2x2: two-tile trips, then a tail of
    one or two tiles): observation counts that hand a wave 5 / 2 / 3 / 1 tiles (N = 2050: 129 tiles, the last one ragged, 17
    per workgroup and 10 for the last), 4 everywhere (2048), and 1 / 1 / 0 / 0 in the last workgroup (1030: 65 tiles) --
    every entry and exit of the statement, against the oracle"""
    from demc_amd import workloads as W
    w = W.cfg2(N=N)
    free_run(demc, orc, w, 6, ["k_res_mvn<256,true,8>"], w["G"], w["Np"], theta_exact=True, beta=0.0, loglike_mode=0)


@pytest.mark.parametrize("Np,kernel", [(200, "k_res_mvn<512,false,8>"), (64, "k_res_mvn<256,false,8>"), (30, "k_res_mvn<256,false,8>")])
def test_d8_vector_pipe_product_in_the_suffstat_instances(demc, orc, Np, kernel):
    """D = 8 instances form Sigma^-1 (theta' - xbar) on the vector pipe inside the quad (no MFMA, no LDS transposition): the
    512-thread form (100 moving particles), the 256-thread form, and a group whose halves (15) leave quads without a particle"""
    from demc_amd import workloads as W
    w = W.cfg2(N=1500, G=6, Np=Np)
    free_run(demc, orc, w, 8, [kernel], 6, Np, theta_exact=True, beta=0.0, loglike_mode=1)


@pytest.mark.parametrize("beta", [0.0, 0.1])
def test_cfg3_geometry_suffstat_lean_kernel(demc, orc, beta):
    """cfg3's group shape (Np = 256, D = 32) in SUFFSTAT mode -> k_res_mvn<512,false,32>, on 8 groups and N = 2000 so that
    the oracle (which visits every observation) finishes in seconds"""
    from demc_amd import workloads as W
    w = W.cfg3(N=2000, G=8)
    free_run(demc, orc, w, 10, ["k_res_mvn<512,false,32>"], 8, 256, theta_exact=beta == 0.0, beta=beta, loglike_mode=1)


@pytest.mark.parametrize("beta", [0.0, 0.1])
def test_cfg3_geometry_streaming_chain(demc, orc, beta):
    """the STREAMING chain (the headline of rounds 1-5; a labelled row since round 6), three kernels: PLAIN K1 (LDS tile, MvNormal preparation on the matrix cores) -> k_cross_mfma<8,4> ->
    k_accept_store, with cfg3's lane geometry (geometry_groups = 256 -> 4 lanes per particle) on 16 groups; fuse = 2 keeps
    the per-phase chain that the full-size population takes by itself"""
    from demc_amd import workloads as W
    w = W.cfg3(N=2000, G=16)
    free_run(demc, orc, w, 8, ["k_propose<256,true,TAIL_PREP_MFMA,false,true>", "k_cross_mfma<8,4>", "k_accept_store"], 16, 256,
             theta_exact=beta == 0.0, beta=beta, loglike_mode=0, fuse=2, geometry_groups=256)


@pytest.mark.parametrize("mode,burnin,kernels", [
    (1, 0, "k_res_mvn<512,false,32,1>"),                                             # SUFFSTAT past burn-in: ONE kernel, the lean body with history partners
    (0, 0, "k_propose<256,false,TAIL_PREP_MFMA,false,true> + k_cross_mfma<8,4> + k_accept_store"),   # STREAMING: the chain, lean K1
    (1, 100, "k_res_mvn<512,false,32,2>"),                                          # burn-in: the base particle comes from the group's current
])                                                                                  # population -- the workgroup's own group: still one kernel
def test_de_mc_z_history_partners_lean_instance(demc, orc, mode, burnin, kernels):
    """DE-MC_Z -- `sample = resample` (crossover.jl:113-124; the reference's own parallel-safe schedule): partners are
    cells of the history of all particles (rows 1:iter-1, n_initial prior rows at the start, utilities.jl:29-41), the synchronous
    schedule.  Past burn-in (random_gamma reads no base particle: crossover.jl:164) nothing a particle reads is written in the
    launch, so the whole update runs in ONE kernel; the default sampler around it takes the lean instance of the no-tile form
    (round 3 ran the ~20 k-instruction general instance and a separate accept kernel).  cfg3's group shape, free-running
    against the oracle; theta bit for bit (beta = 0).  The burn-in case with all 128 quads of a workgroup busy is the one that
    caught a cross-wave race on the base rows (a wave storing its particle's accepted row before another had read it as a base row):
    keep it at this shape."""
    from demc_amd import workloads as W
    w = W.cfg3(N=2000, G=8)
    free_run(demc, orc, w, 8 + 12, [], 8, 256, theta_exact=True, exact_kernels=kernels, beta=0.0, loglike_mode=mode, schedule=1,
             partner_kind=1, n_initial=8, burnin=burnin, geometry_groups=256)


@pytest.mark.parametrize("burnin", [0, 100])
@pytest.mark.parametrize("d,Np,kernel", [(8, 64, "k_res_mvn<256,false,8,%d>"), (12, 40, "k_res_mvn<256,false,0,%d>"),
                                         (7, 30, "k_res_mvn<256,false,0,%d>"), (32, 130, "k_res_mvn<512,false,32,%d>")])
def test_de_mc_z_lean_body_in_its_other_shapes(demc, orc, d, Np, kernel, burnin):
    """k_res_mvn<..., HIST> (DE-MC_Z past burn-in: both halves of a group in one launch per iteration, partner rows = history
    cells read from HBM, the first half's stores held back behind the second half's loads) in the instances the cfg3-shaped test
    above does not reach: D = 8 (two scalars per lane, the product on the vector pipe), a general row length with a ragged second
    block (D = 12), an odd one (D = 7: no 16-byte accesses), and a group whose halves leave quads without a particle (Np = 130 on
    512 threads); past burn-in and inside it (select_base over the whole group's weights of the iteration's start, the base row
    read from the current population before any of the group's writes).  Free-running against the oracle, theta bit for bit."""
    from demc_amd import workloads as W
    w = W.cfg3(N=1500, d=d, G=6, Np=Np)
    free_run(demc, orc, w, 6 + 10, [], 6, Np, theta_exact=True, exact_kernels=kernel % (2 if burnin else 1), beta=0.0, loglike_mode=1,
             schedule=1, partner_kind=1, n_initial=6, burnin=burnin)


@pytest.mark.parametrize("Np,burnin,kernel", [(31, 0, "k_res_mvn<256,false,32,1>"), (31, 100, "k_res_mvn<256,false,32,2>"),
                                              (256, 100, "k_res_mvn<512,false,32,2>")])
def test_de_mc_z_lean_body_with_mutation_sweeps_and_odd_groups(demc, orc, Np, burnin, kernel):
    """the same kernel through mutation sweeps (beta = 0.3: a group in three takes pt + Normal(0, sigma) instead of the crossover,
    mutation.jl:13-25 -- no partner cells, no base row, but the same held-back stores and the same barrier) and with an odd group
    (Np = 31: halves of 15 and 16).  Mutation's device log / sincospi: theta to 1e-10."""
    from demc_amd import workloads as W
    w = W.cfg3(N=1500, d=32, G=6, Np=Np)
    free_run(demc, orc, w, 6 + 12, [], 6, Np, theta_exact=False, exact_kernels=kernel, beta=0.3, loglike_mode=1, schedule=1,
             partner_kind=1, n_initial=6, burnin=burnin)


@pytest.mark.parametrize("d,Np,burnin,snooker,kernel", [(32, 256, 0, 0.1, "k_res_mvn<512,false,32,3>"), (32, 256, 100, 0.5, "k_res_mvn<512,false,32,3>"),
                                                        (8, 64, 100, 0.3, "k_res_mvn<256,false,8,3>"), (12, 30, 0, 0.3, "k_res_mvn<256,false,0,3>")])
def test_de_mc_z_with_snooker_as_the_reference_runs_it(demc, orc, d, Np, burnin, snooker, kernel):
    """test/multivariate_normal_tests.jl:50-59 and Examples/Hierarchical_Example.jl:103-114 run DE-MC_Z with theta_snooker = 0.1:
    history cells for the snooker's three particles too (crossover.jl:241-243 through de.sample).  On the default sampler's family
    that is instance 3 of the lean body (round 4's first form: the general kernel's LEAN-2 instance at 2.2x the cycles per
    particle): snooker and crossover particles side by side in a wave, in and past burn-in, the third cell in the slot of the
    base row, adjust_loglike's norms carried into the decision.  The projections are reduced in another order than the oracle's:
    theta to 1e-10; every accept decision equal."""
    from demc_amd import workloads as W
    w = W.cfg3(N=2000, d=d, G=8, Np=Np)
    free_run(demc, orc, w, 8 + 12, [], 8, Np, theta_exact=False, exact_kernels=kernel, beta=0.0, loglike_mode=1, schedule=1,
             partner_kind=1, n_initial=8, burnin=burnin, theta_snooker=snooker)


@pytest.mark.parametrize("Np,burnin,snooker,beta,kernel", [
    (64, 0, 0.1, 0.0, "k_res_mvn<256,false,31,3,iso>"),  # the reference's own setting: DE-MC_Z + snooker 0.1 (:50-59), past burn-in
    (64, 100, 0.1, 0.0, "k_res_mvn<256,false,31,3,iso>"),  # ... inside burn-in: base particles for the crossover particles of a wave
    (200, 100, 0.5, 0.3, "k_res_mvn<512,false,31,3,iso>"),  # 512 threads, half the particles snooker, mutation sweeps
    (64, 0, 0.0, 0.0, "k_res_mvn<256,false,31,1,iso>"),   # no snooker: instance 1 past burn-in
    (30, 100, 0.0, 0.0, "k_res_mvn<256,false,31,2,iso>"),  # ... instance 2 inside it; halves of 15 leave quads without a particle
])
def test_mvn30_de_mc_z_takes_the_lean_iso_instance(demc, orc, Np, burnin, snooker, beta, kernel):
    """test/multivariate_normal_tests.jl:6-59: MvNormal(mu, sigma^2 I) with sigma a parameter (theta = (mu[30], sigma), D = 31:
    an odd row, two prior segments -- Normal on mu, Cauchy+ on sigma -- and a lower bound on sigma), run by the reference as DE-MC_Z
    with theta_snooker = 0.1.  Round 4 served it with the general kernel's LEAN-2 instance; round 5 gave the lean DE-MC_Z body an
    isotropic form (k_res_mvn<..., iso>: |mu - xbar|^2 on the vector pipe inside the quad, no matrix stage).  The workload of
    bench.py's mvn30 rows on 6 groups, free-running against the oracle: every accept decision and id equal, lp to 1e-9,
    theta to 1e-10 (snooker projections and mutation's device log are reduced / evaluated in another order)."""
    from demc_amd import workloads as W
    w = W.mvn30(G=6, Np=Np)
    free_run(demc, orc, w, 8 + 12, [], 6, Np, theta_exact=(snooker == 0.0 and beta == 0.0), exact_kernels=kernel, beta=beta, loglike_mode=1,
             schedule=1, partner_kind=1, n_initial=8, burnin=burnin, theta_snooker=snooker)
    if Np == 64 and burnin == 100:  # another row length (12 means + sigma: the instance without a compiled-in D)
        w = W.mvn30(d=12, G=6, Np=Np)
        free_run(demc, orc, w, 8 + 12, [], 6, Np, theta_exact=False, exact_kernels="k_res_mvn<256,false,0,3,iso>", beta=0.0, loglike_mode=1,
                 schedule=1, partner_kind=1, n_initial=8, burnin=burnin, theta_snooker=snooker)


@pytest.mark.parametrize("burnin,kernels", [
    (0, "k_longrow<512>"),    # past burn-in nothing a particle reads is written in the launch: ONE k_longrow launch per block sweep
    (100, "k_longrow<512>"),  # inside it random_gamma reads a base particle of the current population: from a snapshot of the sweep's start
    (4, "k_longrow<512>"),    # ... and a run that leaves burn-in on the way
])
def test_hierarchical_example_configuration_de_mc_z_snooker_blocks(demc, orc, burnin, kernels):
    """Examples/Hierarchical_Example.jl:88-114 -- the reference's own hierarchical run: `sample = resample` (DE-MC_Z), theta_snooker =
    0.1, block updates [hyper ; subject] on every iteration -- on cfg4's family with long rows (S = 2100 subjects: a workgroup per
    particle), as bench.py's cfg4_whole_history_partners_snooker_blocks rows run it: long partner rows gathered from the
    history, snooker updates that read three of them and need them in the hyper-parameter sweep too (adjust_loglike's norms run
    over every scalar).  Free-running against the oracle; which kernel serves it is asserted: the long-row kernel, past burn-in
    and -- round 5 -- inside it too (the base particle of the current population, which another workgroup of the synchronous
    launch may be writing, is read from a device-to-device snapshot of the sweep's start; round 4 fell back to the per-phase
    chain K1 -> k_hier_loglike -> K3 there, 2.3x the time)."""
    from demc_amd import workloads as W
    w = W.cfg4(S=2100, G=4, Np=8)
    free_run(demc, orc, w, 4 + 8, [], 4, 8, theta_exact=False, exact_kernels=kernels, beta=0.0, schedule=1, partner_kind=1, n_initial=4,
             burnin=burnin, theta_snooker=0.1, lp_rtol=1e-8)


@pytest.mark.parametrize("which", ["hier_small", "gaussian", "mvn_general_row"])
def test_de_mc_z_inside_burn_in_is_one_launch_for_the_general_kernel_too(demc, orc, which):
    """DE-MC_Z inside burn-in on the families the GENERAL kernel serves (short hierarchical rows with the reference's blocks and
    snooker, Examples/Gaussian_Example.jl's model, an MvNormal row the lean body has no instance for): random_gamma's base
    particle is read from the snapshot of the sweep's start (KParams::base_theta), so the synchronous sweep is ONE fused launch
    -- no K3 -- as it is past burn-in; every decision equal to the oracle's."""
    from demc_amd import workloads as W
    extra = dict(theta_snooker=0.1)
    if which == "hier_small":
        w, G, Np = W.cfg4(S=40, G=6, Np=16), 6, 16
    elif which == "gaussian":
        w, G, Np = W.cfg1(), 4, 10
        extra = {}
    else:
        w, G, Np = W.cfg3(N=500, d=40, G=4, Np=24), 4, 24  # (D = 40 > 32: no lean instance)
        extra = dict(loglike_mode=1)
    ran = free_run(demc, orc, w, 4 + 10, [], G, Np, theta_exact=False, beta=0.0, schedule=1, partner_kind=1, n_initial=4, burnin=100,
                   lp_rtol=1e-8, **extra)
    assert "k_propose<" in ran and "k_accept_store" not in ran, ran


def _de_mc_z_cases(n, seed=20261004):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        cfg = dict(d=int(rng.integers(2, 33)), Np=int(rng.integers(6, 70)), G=int(rng.integers(1, 7)), burnin=int(rng.choice([0, 4, 100])),
                   theta_snooker=float(rng.choice([0.0, 0.0, 0.3])), beta=float(rng.choice([0.0, 0.3])), loglike_mode=int(rng.choice([1, 1, 0])),
                   n_initial=int(rng.integers(1, 6)), seed=int(rng.integers(1, 2**31)))
        out.append(pytest.param(cfg, id=f"{i}-d{cfg['d']}-Np{cfg['Np']}-b{cfg['burnin']}-s{cfg['theta_snooker']:g}-m{cfg['loglike_mode']}"))
    return out


@pytest.mark.parametrize("cfg", _de_mc_z_cases(16))
def test_de_mc_z_randomised_free_runs(demc, orc, cfg):
    """DE-MC_Z over random shapes and sampler settings, free-running against the oracle: row lengths 2..32 (the lean body's general,
    D = 8 and D = 32 instances and their ragged blocks), groups of 6..69 particles, runs that start inside burn-in and leave it
    (instance 2 -> 1), snooker updates (instance 3), mutation sweeps, SUFFSTAT and STREAMING (the general kernel's chain): every
    accept decision and every particle id equal, theta to 1e-10."""
    from demc_amd import workloads as W
    cfg = dict(cfg)
    d, Np, G, n_init = cfg.pop("d"), cfg.pop("Np"), cfg.pop("G"), cfg["n_initial"]
    w = W.cfg3(N=700, d=d, G=G, Np=Np)
    free_run(demc, orc, w, n_init + 10, [], G, Np, theta_exact=False, schedule=1, partner_kind=1, **cfg)


def _two_colour_cases(n, seed=20261005):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        cfg = dict(d=int(rng.integers(2, 33)), Np=int(rng.integers(6, 70)), G=int(rng.integers(1, 7)), burnin=int(rng.choice([0, 4, 100])),
                   beta=float(rng.choice([0.0, 0.3])), loglike_mode=int(rng.choice([1, 1, 0])), alpha=float(rng.choice([0.1, 0.5])),
                   seed=int(rng.integers(1, 2**31)))
        out.append(pytest.param(cfg, id=f"{i}-d{cfg['d']}-Np{cfg['Np']}-G{cfg['G']}-b{cfg['burnin']}-m{cfg['loglike_mode']}"))
    return out


@pytest.mark.parametrize("cfg", _two_colour_cases(12))
def test_default_sampler_randomised_free_runs(demc, orc, cfg):
    """the default sampler on MvNormal-full over random shapes, free-running against the oracle on the two_colour schedule: the lean
    resident kernel in its general, D = 8 and D = 32 instances (SUFFSTAT), the streaming-resident forms (STREAMING), groups of
    6..69 particles with odd halves, runs that leave burn-in inside a launch, mutation sweeps, migrations every second
    iteration: every accept decision and particle id equal, theta to 1e-10."""
    from demc_amd import workloads as W
    cfg = dict(cfg)
    d, Np, G = cfg.pop("d"), cfg.pop("Np"), cfg.pop("G")
    w = W.cfg3(N=700, d=d, G=G, Np=Np)
    free_run(demc, orc, w, 12, [], G, Np, theta_exact=False, **cfg)


@pytest.mark.parametrize("kernel,G,Np,extra", [("k_longrow<512>", 4, 8, {}), ("k_frozen_sweep<256,big>", 40, 32, {}), ("k_frozen_sweep<256,big>", 128, 32, {}),
                                               ("k_longrow<256>", 40, 32, dict(kappa=0.9)), ("k_longrow<256>", 128, 32, dict(kappa=0.9))])
def test_cfg4_shape_long_row_span_loops(demc, orc, kernel, G, Np, extra):
    """hierarchical Binomial with the two blocks [hyper; subject] of Examples/Hierarchical_Example.jl:88-92 at S = 2100
    (rows long enough for a workgroup per particle): k_longrow<512> (one particle per CU), and once the moving particles
    outnumber twice the CUs the row-streaming kernel (k_frozen_sweep: no LDS row, three or four workgroups per CU; the name
    checked is the LAST sweep's, the subject block's) -- or, where that kernel has no form (recombination, kappa < 1),
    k_longrow<256>, two workgroups per CU.  Mutation sweeps (beta = 0.1) included: theta to 1e-10.
    k_longrow is PERSISTENT (round 4): 40 x 32 gives 640 moving particles per colour phase to 512 resident workgroups (some
    take two), 128 x 32 -- BASELINE's whole cfg4 population -- four each: the row of one particle (accepted proposal, or the
    current row of a rejected one) is written from inside the span loops of the next, block by block; the history compared
    here is what those deferred stores wrote."""
    from demc_amd import workloads as W
    w = W.cfg4(S=2100, G=G, Np=Np)
    free_run(demc, orc, w, 6, [kernel], G, Np, theta_exact=False, **extra)


@pytest.mark.parametrize("kernel,extra", [("k_frozen_sweep<256,big>", dict(theta_snooker=0.3)), ("k_longrow<256>", dict(kappa=0.8)),
                                          ("k_frozen_sweep<256,big>", dict(beta=0.5)), ("k_longrow<256>", dict(theta_snooker=0.3, kappa=0.7, beta=0.3)),
                                          ("k_longrow<256>", dict(masks=None)), ("k_longrow<256>", dict(S=2101)),
                                          ("k_longrow<256>", dict(theta_snooker=0.3, kappa=0.999)), ("k_longrow<256>", dict(beta=0.5, kappa=0.999))])
def test_long_row_kernel_every_sweep_kind_trace_free(demc, orc, kernel, extra):
    """the long-row kernels WITHOUT the trace (the forms that ship) through snooker sweeps (moving and frozen spans), recombination
    (k_longrow's general per-pair body), mutation-heavy runs, an unblocked row and an odd row length (no span loops at all) --
    history against the oracle, free-running.  (kappa = 0.999 keeps k_longrow<256> on the sweep kinds the row-streaming kernel
    has taken over: it remains the kernel of every configuration with recombination.)"""
    from demc_amd import workloads as W
    extra = dict(extra)
    w = W.cfg4(S=extra.pop("S", 2100), G=40, Np=32)
    if "masks" in extra:
        w["masks"] = extra.pop("masks")
    free_run(demc, orc, w, 5, [kernel], 40, 32, theta_exact=False, **extra)


def _de_mc_z_family_cases(n, seed=20261007):
    """DE-MC_Z over the OTHER families and row shapes (round 5): MvNormal(mu, sigma^2 I) with sigma a parameter at random
    dimensions (the lean iso instances: D = 31 and the general row length, odd and even rows), short hierarchical rows with and
    without blocks, the Gaussian example -- inside burn-in (the sweep-start snapshot), leaving it, and past it."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        fam = str(rng.choice(["iso", "iso", "iso30", "hier", "gauss"]))
        c = dict(fam=fam, d=int(rng.integers(2, 31)), Np=int(rng.integers(6, 70)), G=int(rng.integers(1, 7)), S=int(rng.integers(8, 60)),
                 burnin=int(rng.choice([0, 4, 100])), theta_snooker=float(rng.choice([0.0, 0.1, 0.3])), beta=float(rng.choice([0.0, 0.3])),
                 n_initial=int(rng.integers(1, 6)), blocks=bool(rng.random() < 0.5), seed=int(rng.integers(1, 2**31)))
        out.append(pytest.param(c, id=f"{i}-{fam}-d{c['d']}-Np{c['Np']}-b{c['burnin']}-s{c['theta_snooker']:g}"))
    return out


def run_de_mc_z_family_case(demc, orc, c):
    from demc_amd import workloads as W
    c = dict(c)
    fam, d, Np, G, S, blocks, n_init = c.pop("fam"), c.pop("d"), c.pop("Np"), c.pop("G"), c.pop("S"), c.pop("blocks"), c["n_initial"]
    if fam in ("iso", "iso30"):
        w = W.mvn30(N=60, d=30 if fam == "iso30" else d, G=G, Np=Np)
        c.update(loglike_mode=1)
    elif fam == "hier":
        w = W.cfg4(S=S, G=G, Np=Np)
        if not blocks:
            w["masks"] = None
    else:
        w = dict(W.cfg1(), G=G, Np=Np)
    return free_run(demc, orc, w, n_init + 10, [], G, Np, theta_exact=False, schedule=1, partner_kind=1, lp_rtol=1e-8, **c)


@pytest.mark.parametrize("c", _de_mc_z_family_cases(10))
def test_de_mc_z_randomised_free_runs_over_the_other_families(demc, orc, c):
    """the generator above against the oracle: every accept decision and particle id equal, theta to 1e-10 (tests/free_run_sweep.py
    de_mc_z_families N runs the long form)"""
    run_de_mc_z_family_case(demc, orc, c)


_Z = dict(schedule=1, partner_kind=1, n_initial=4)  # DE-MC_Z: partners from the history, the synchronous schedule


@pytest.mark.parametrize("family,S,beta,burnin,extra", [
    ("hier_binomial", 2100, 0.0, 100, {}),   # inside burn-in: random_gamma's base particle (select_base over the resting colour)
    ("hier_binomial", 2100, 0.3, 0, {}),     # mutation sweeps: the whole row moves, an accepted one is formed again for its stores
    ("hier_binomial", 2101, 0.1, 3, {}),     # an odd row: the scalar-per-lane form of the frozen loop
    ("hier_gaussian", 2100, 0.1, 100, {}),   # three runs inside the block (mu, sd of the effects ... the observation sd at the end)
    ("hier_binomial", 2100, 0.1, 0, dict(theta_snooker=0.3)),          # snooker: projections over three whole partner rows
    ("hier_binomial", 2100, 0.1, 0, dict(_Z, theta_snooker=0.3)),      # DE-MC_Z past burn-in: partner cells of the history
    ("hier_binomial", 2101, 0.1, 100, dict(_Z, theta_snooker=0.3)),    # ... inside it: base row and weights from the sweep-start snapshot
    ("hier_gaussian", 2100, 0.1, 6, dict(_Z, theta_snooker=0.1)),      # ... leaving it on the way; Examples/Hierarchical_Example.jl's snooker
    ("hier_binomial", 2100, 0.0, 100, dict(_Z)),                       # DE-MC_Z without snooker
])
def test_frozen_row_sweep_kernel(demc, orc, family, S, beta, burnin, extra):
    """k_frozen_sweep (demc_frozen.hpp): a block sweep whose block holds only the hyper-parameters freezes the row -- the sweep is
    one pass over the particle's own row, no partner rows, no LDS row.  Here the ONLY block is the hyper-parameter block of
    Examples/Hierarchical_Example.jl:88-92, so it is also the iteration's last sweep and writes the history rows the comparison
    reads: free-running against the oracle, every decision and id equal, theta to 1e-10 (mutation's device log), the kernel by
    name.  40 x 32 particles: enough for the form to be taken (two workgroups' worth of particles per CU)."""
    from conftest import make_problem
    G, Np = 40, 32
    prob = make_problem(family, np.random.default_rng(97), S=S)
    m0 = np.zeros(prob["D"], np.uint8)
    m0[:2] = 1
    if family == "hier_gaussian":
        m0[-1] = 1
    rng = np.random.default_rng(98)
    w = dict(prob, G=G, Np=Np, masks=m0[None, :], engine={}, init=lambda P, rng_: prob["init"](P))
    ran = free_run(demc, orc, w, extra.get("n_initial", 0) + 6, [], G, Np, theta_exact=False, exact_kernels="k_frozen_sweep<256>", beta=beta,
                   burnin=burnin, lp_rtol=1e-8, **extra)
    assert ran == "k_frozen_sweep<256>"


@pytest.mark.parametrize("burnin", [0, 100, 6])
def test_hierarchical_example_configuration_with_enough_particles_for_the_frozen_sweep(demc, orc, burnin):
    """Examples/Hierarchical_Example.jl:88-114 again (DE-MC_Z, theta_snooker = 0.1, blocks [hyper ; subject]) with 40 x 16 particles:
    the hyper-parameter sweep is k_frozen_sweep (not the iteration's last sweep: no history row to write, the row it leaves is
    what the subject sweep reads), the subject sweep k_longrow -- two kernels alternating on one stream, against the oracle."""
    from demc_amd import workloads as W
    w = W.cfg4(S=2100, G=40, Np=16)
    ran = free_run(demc, orc, w, 4 + 6, [], 40, 16, theta_exact=False, beta=0.1, burnin=burnin, theta_snooker=0.1, lp_rtol=1e-8, **_Z)
    assert ran == "k_frozen_sweep<256,big>", ran


@pytest.mark.parametrize("order,beta,kernel", [("hyper_then_subject", 0.1, "k_frozen_sweep<256,big>"), ("subject_then_hyper", 0.1, "k_frozen_sweep<256>"),
                                               ("hyper_then_subject", 0.5, "k_frozen_sweep<256,big>")])
def test_row_streaming_kernel_at_the_benchmarked_row_length(demc, orc, order, beta, kernel):
    """k_frozen_sweep AT cfg4's ROW LENGTH (missing 2): S = 10 000 subjects, D = 10 002 -- the rows the `cfg4_whole*`
    bench rows stream -- on 40 x 32 particles (640 moving particles per colour phase: the form is taken), blocks [hyper ; subject]
    of Examples/Hierarchical_Example.jl:88-92, crossover and mutation sweeps (beta = 0.1 / 0.5), inside and past burn-in
    (burnin = 3 of 6 iterations), migrations on: free-running against the oracle, every accept decision and particle id equal,
    theta to 1e-10.  demc_last_kernels names the LAST sweep's instance, so the block order is run both ways: the subject
    block last names k_frozen_sweep<256,big> (every scalar proposed on the fly from own, partner and base rows, 40 rounds of 256
    scalars per row, the ragged end at scalar 10 002), the hyper-parameter block last names k_frozen_sweep<256> (the frozen pass
    over 10 000 subject terms, whose history row is then what the comparison reads)."""
    from demc_amd import workloads as W
    w = W.cfg4(S=10000, G=40, Np=32)
    assert w["D"] == 10002
    if order == "subject_then_hyper":
        w["masks"] = w["masks"][::-1].copy()
    ran = free_run(demc, orc, w, 6, [], 40, 32, theta_exact=False, lp_rtol=1e-8, beta=beta)
    assert ran == kernel, ran


@pytest.mark.parametrize("burnin", [100, 0, 5])
def test_hierarchical_example_configuration_at_the_benchmarked_row_length(demc, orc, burnin):
    """Examples/Hierarchical_Example.jl:88-114 as the reference runs it -- `sample = resample` (DE-MC_Z), theta_snooker = 0.1, blocks
    [hyper ; subject] -- at D = 10 002 on 40 x 32 particles (synchronous schedule: 1 280 moving particles), inside burn-in (base
    rows and select_base's weights from the sweep-start snapshot: block columns for the hyper-parameter sweep, the rows that sweep
    streamed for the subject sweep), past it, and leaving it on the way: both k_frozen_sweep instances alternate on one stream,
    partner rows are 80 KB history cells, a snooker particle projects over three of them -- against the oracle, free-running."""
    from demc_amd import workloads as W
    w = W.cfg4(S=10000, G=40, Np=32)
    ran = free_run(demc, orc, w, 4 + 6, [], 40, 32, theta_exact=False, beta=0.1, burnin=burnin, theta_snooker=0.1, lp_rtol=1e-8, **_Z)
    assert ran == "k_frozen_sweep<256,big>", ran


@pytest.mark.parametrize("extra", [dict(), dict(_Z, theta_snooker=0.1, burnin=2)])
def test_whole_cfg4_population_against_the_oracle(demc, orc, extra):
    """BASELINE's cfg4 at the WHOLE population's geometry -- 128 groups x 32 particles x 10 002 scalars, the `cfg4_whole` bench row
    (two_colour, partners from the population) and the reference's own configuration of it (DE-MC_Z + snooker + blocks, leaving
    burn-in inside the run) -- for a few iterations against the oracle (4 096 particles x 10 002 scalars x 2 sweeps: a second per
    iteration on the host's threads).  Until round 6 these rows were checked for finite weights only."""
    from demc_amd import workloads as W
    w = W.cfg4(S=10000, G=128, Np=32)
    n_init = extra.get("n_initial", 0)
    ran = free_run(demc, orc, w, n_init + 3, [], 128, 32, theta_exact=False, beta=0.1, lp_rtol=1e-8, **extra)
    assert ran == "k_frozen_sweep<256,big>", ran


def _long_row_cases(n, seed=20261006):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        c = dict(S=int(rng.integers(2047, 2600)), G=int(rng.integers(2, 44)), Np=int(rng.choice([6, 8, 10, 16, 32])),
                 theta_snooker=float(rng.choice([0.0, 0.0, 0.3])), beta=float(rng.choice([0.1, 0.5])), kappa=float(rng.choice([1.0, 1.0, 0.8])),
                 burnin=int(rng.choice([0, 3, 100])), seed=int(rng.integers(1, 2**31)), hist=bool(rng.random() < 0.3),
                 blocks=bool(rng.random() >= 0.25))
        out.append(pytest.param(c, id=f"{i}-S{c['S']}-G{c['G']}-Np{c['Np']}-h{int(c['hist'])}-b{int(c['blocks'])}"))
    return out


@pytest.mark.parametrize("c", _long_row_cases(8))
def test_long_row_randomised_free_runs(demc, orc, c):
    """hierarchical Binomial rows of 2 049 .. 2 601 scalars (odd and even, ragged last blocks) over random group counts and
    sizes -- one or two workgroups per CU, one to several particles per persistent workgroup --, with and without the two block
    sweeps, snooker, recombination, mutation-heavy runs, partners from the population or from the history: free-running against
    the oracle (a 60-case sweep of this generator was clean)."""
    from demc_amd import workloads as W
    c = dict(c)
    S, G, Np, hist, blocks = c.pop("S"), c.pop("G"), c.pop("Np"), c.pop("hist"), c.pop("blocks")
    w = W.cfg4(S=S, G=G, Np=Np)
    if not blocks:
        w["masks"] = None
    if hist:
        c.update(schedule=1, partner_kind=1, n_initial=3)
    free_run(demc, orc, w, (3 if hist else 0) + 5, [], G, Np, theta_exact=False, **c)


def _row_streaming_cases(n, seed=20261008):
    """populations large enough for k_frozen_sweep (two workgroups' worth of moving particles per CU) over random row lengths,
    block layouts (the reference's [hyper ; subject]; the hyper-parameters one block each and the subjects in two halves: short runs
    inside the block next to long ones, a run that starts on an odd scalar), snooker, mutation-heavy runs, partners from the
    population or from the history, inside / leaving / past burn-in"""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        hist = bool(rng.random() < 0.5)
        c = dict(S=int(rng.choice([2046, 2047, 2100, 2210, 2304])), G=int(rng.integers(33, 44)) if hist else int(rng.integers(65, 72)),
                 Np=int(rng.choice([16, 32])) if hist else 16,
                 # (DE-MC_Z from prior-drawn history rows accepts next to nothing without snooker moves in so short a run: the
                 # reference's own DE-MC_Z runs use theta_snooker = 0.1; the snooker-free form is test_frozen_row_sweep_kernel's)
                 theta_snooker=float(rng.choice([0.1, 0.3] if hist else [0.0, 0.1, 0.3])), beta=float(rng.choice([0.0, 0.1, 0.5])),
                 burnin=int(rng.choice([0, 5, 100])), seed=int(rng.integers(1, 2**31)), hist=hist, layout=int(rng.integers(0, 3)))
        out.append(pytest.param(c, id=f"{i}-S{c['S']}-G{c['G']}-Np{c['Np']}-h{int(hist)}-l{c['layout']}-s{c['theta_snooker']:g}-b{c['burnin']}"))
    return out


def run_row_streaming_case(demc, orc, c):
    from demc_amd import workloads as W
    c = dict(c)
    S, G, Np, hist, layout = c.pop("S"), c.pop("G"), c.pop("Np"), c.pop("hist"), c.pop("layout")
    w = W.cfg4(S=S, G=G, Np=Np)
    D = w["D"]
    if layout == 1:  # four blocks: mu | sd | the first subjects up to an odd border | the rest
        cut = 2 + (S // 2) | 1
        m = np.zeros((4, D), np.uint8)
        m[0, 0] = 1; m[1, 1] = 1; m[2, 2:cut] = 1; m[3, cut:] = 1
        w["masks"] = m
    elif layout == 2:  # the subject block first, then the hyper-parameters (the by-product snapshot has no sweep to serve)
        w["masks"] = w["masks"][::-1].copy()
    if hist:
        c.update(schedule=1, partner_kind=1, n_initial=3)
    return free_run(demc, orc, w, (3 if hist else 0) + 4, [], G, Np, theta_exact=False, lp_rtol=1e-8, **c)


@pytest.mark.parametrize("c", _row_streaming_cases(6))
def test_row_streaming_kernel_randomised_free_runs(demc, orc, c):
    """the generator above against the oracle: every accept decision and particle id equal, theta to 1e-10
    (tests/free_run_sweep.py row_streaming N runs the long form)"""
    ran = run_row_streaming_case(demc, orc, c)
    if c["S"] % 2 == 0 or c["layout"] == 2:  # (the name is the LAST sweep's; an odd row's subject sweeps stay with k_longrow)
        assert ran.startswith("k_frozen_sweep<256"), ran


@pytest.mark.parametrize("N", [500, 1500, 4096, 5000])
def test_cfg5_shape_lba_wave_per_proposal(demc, orc, N):
    """LBA, 3 accumulators, snooker 0.1 (Examples/Run_LBA.jl), simulated trials: K1 -> k_lba_wave (a wave per proposal, lanes
    across the trials, which demc_set_model sorted by (choice, decision time); Phi / phi tables in LDS) -> k_accept_store.
    N = 500: no whole batch of 512 trials, the ragged-end loop alone; 1 500: two whole batches and a ragged end; 4 096: whole
    batches only, several chunks; 5 000: chunks of unequal length.  LBA log-densities at 1e-5 (survival factors formed by
    cancellation, see test_gpu_parity._rtol), snooker projections to 1e-10."""
    from demc_amd import workloads as W
    w = W.cfg5(N=N, G=8, Np=16)
    free_run(demc, orc, w, 10, ["k_lba_wave", "k_accept_store"], 8, 16, theta_exact=False, lp_rtol=1e-5)


def test_sample_runs_the_production_instances(demc):
    """the public sample() no longer asks for the diagnostic trace: the default sampler on the Gaussian example runs the
    PLAIN resident instance"""
    import demc_amd as D
    from demc_amd import sampler as S
    seen = []
    real = D.HipEngine

    def factory(**cfg):
        e = real(**cfg)
        seen.append(e)
        return e
    rng = np.random.default_rng(3)
    data = rng.normal(0.0, 1.0, 50)
    sp = lambda: [rng.normal(0, 1), abs(rng.standard_cauchy()) + 0.1]
    model = D.DEModel(sample_prior=sp, prior_loglike=D.Priors(mu=D.Normal(0, 1), sigma=D.TruncatedCauchy(0, 1)),
                      loglike=D.GaussianLikelihood(), data=data, names=("mu", "sigma"))
    de = D.DE(sample_prior=sp, bounds=((-np.inf, np.inf), (0.0, np.inf)), burnin=200, Np=6)
    S.sample(model, de, D.HIPBackend(seed=3), 400, engine_factory=factory)
    assert seen and seen[0].cfg.trace == 0
    # (the engine is closed by sample(); its configuration is what this test is about)


@pytest.mark.parametrize("n_distinct,kernel", [(32, "k_propose<512,true,TAIL_PREP_MFMA,true,true>"), (8, "k_res_mvn<512,false,0>"),
                                               (1, "k_res_mvn<512,false,32>")])
def test_lean_kernel_plan_follows_the_prior_table(demc, orc, n_distinct, kernel):
    """priors and bounds arrive AFTER demc_set_model in every host (workloads.configure, DEMCHIP.jl, the C driver): the lean
    kernel's plan must be taken again then.  32 distinct per-dimension priors / bounds do not fit the run-length table
    (16 segments): the general resident instance has to run (round 2 launched the lean kernel on an empty segment table);
    8 distinct ones take the lean kernel's general-row instance; one segment the D = 32 instance.  All three == oracle."""
    from demc_amd import workloads as W
    w = W.cfg3(N=500, G=4)
    d = w["D"]
    seg = np.arange(d) // (d // n_distinct)
    w["pa"] = list(0.05 * seg)                 # Normal(0.05 k, 1 + 0.01 k) on the dimensions of segment k
    w["pb"] = list(1.0 + 0.01 * seg)
    w["lo"] = list(-50.0 - seg)
    w["hi"] = list(60.0 + seg)
    free_run(demc, orc, w, 8, [kernel], 4, 256, theta_exact=True, beta=0.0, loglike_mode=1)


# ---------------------------------------------------------------------------------------------------------------------
# DIRECT likelihood: the residual form sum_i |L^-1 (x_i - mu)|^2, term by term (what the reference's loglike implies,
# test/multivariate_normal_tests.jl:31-33; SURVEY 8d's 3ND count) -- the form that does not collapse after centring
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("family,d", [("mvn_full", 5), ("mvn_full", 8), ("mvn_full", 13), ("mvn_full", 32), ("mvn_full", 40),
                                      ("mvn_iso", 6), ("mvn_iso", 20)])
def test_direct_mode_log_posteriors_match_the_oracle(demc, orc, family, d):
    from conftest import make_problem, setup_engine
    prob = make_problem(family, np.random.default_rng(90 + d), N=700, d=d)
    th = prob["init"](64)
    o = orc.Oracle(n_groups=4, Np=16, D=prob["D"], schedule=1)
    setup_engine(o, prob)
    want = o.logpost(th)
    o.close()
    for mode in (2, 0, 1):
        e = demc.HipEngine(n_groups=4, Np=16, D=prob["D"], schedule=1, loglike_mode=mode)
        setup_engine(e, prob)
        np.testing.assert_allclose(e.logpost(th), want, rtol=1e-9, err_msg=f"loglike_mode {mode}")
        e.close()
    with pytest.raises(demc.DemcError):
        demc.HipEngine(n_groups=4, Np=16, D=2, schedule=1, loglike_mode=3)


@pytest.mark.parametrize("beta", [0.0, 0.1])
def test_cfg3_geometry_direct_chain(demc, orc, beta):
    """PLAIN K1 (whitened proposal m = L^-1 mu~ on the matrix cores) -> k_direct_mvn<32> -> k_accept_store, free-running
    against the oracle (whose MvNormal likelihood is the same whitened residual form, oracle/demc_oracle.c:490-505)"""
    from demc_amd import workloads as W
    w = W.cfg3(N=2000, G=16)
    free_run(demc, orc, w, 8, ["k_propose<256,true,TAIL_PREP_MFMA,false,true>", "k_direct_mvn<32>", "k_accept_store"], 16, 256,
             theta_exact=beta == 0.0, beta=beta, loglike_mode=2, geometry_groups=256)


@pytest.mark.parametrize("shape,beta", [("cfg2", 0.0), ("cfg2", 0.1), ("cfg2_ragged", 0.1), ("d32", 0.1), ("d32_small_groups", 0.0)])
def test_direct_likelihood_in_the_streaming_resident_lean_kernel(demc, orc, shape, beta):
    """DIRECT mode on a population too small to fill the chip with the K1 -> k_direct_mvn -> K3 chain (BASELINE cfg2: 32 x 64, D = 8,
    N = 1e4 -- six dependent launches an iteration): the streaming-resident lean kernel's DIRECT instance, the residual form
    sum_i |z_i - m|^2 term by term on the vector pipe out of an LDS copy of the workgroup's chunk of whitened rows, C workgroups
    per group handing their sums over.  Free-running against the oracle (whose MvNormal likelihood IS the whitened residual form,
    oracle/demc_oracle.c:490-505): cfg2 as benchmarked; an observation count that leaves the last chunk short and ragged; D = 32
    (the proposal's m through the matrix-core preparation) with 32 and with 9 / 10 moving particles (slices of 32 and of 16 lanes)."""
    from demc_amd import workloads as W
    if shape == "cfg2":
        w, kern = W.cfg2(), "k_res_mvn<512,true,8,direct>"
    elif shape == "cfg2_ragged":
        w, kern = W.cfg2(N=2050), "k_res_mvn<512,true,8,direct>"
    elif shape == "d32":
        w, kern = W.cfg3(N=3000, G=16, Np=64), "k_res_mvn<256,true,32,direct>"
    else:
        w, kern = W.cfg3(N=1500, G=12, Np=19), "k_res_mvn<256,true,32,direct>"
    free_run(demc, orc, w, 10, [], w["G"], w["Np"], theta_exact=beta == 0.0, exact_kernels=kern, beta=beta, loglike_mode=2, alpha=0.3)


def test_the_three_likelihood_modes_make_the_same_decisions(demc):
    """STREAMING (expanded form on the matrix cores), SUFFSTAT and DIRECT are three evaluation orders of one log-density:
    same proposals, same accept decisions, log-posteriors to rounding -- cfg2's shape, 40 iterations"""
    from demc_amd import workloads as W
    w = W.cfg2(N=4000)
    G, Np, n_it = w["G"], w["Np"], 40
    th0 = w["init"](G * Np, np.random.default_rng(6))
    outs, names = [], []
    for mode in (0, 1, 2):
        e = demc.HipEngine(n_groups=G, Np=Np, D=w["D"], n_rows=n_it, schedule=2, seed=77, burnin=20, loglike_mode=mode)
        W.configure(e, w)
        e.set_state(th0)
        e.step(1, n_it)
        outs.append(e.get_history(0, n_it))
        names.append(e.last_kernels())
        e.close()
    # (round 6: a population this small takes the DIRECT likelihood inside the streaming-resident lean kernel too)
    assert "k_res_mvn<256,true,8>" in names[0] and "k_res_mvn<256,false,8>" in names[1] and names[2] == "k_res_mvn<512,true,8,direct>", names
    for other in outs[1:]:
        assert np.array_equal(outs[0][1], other[1]) and np.array_equal(outs[0][3], other[3])
        assert np.array_equal(outs[0][0], other[0])
        np.testing.assert_allclose(outs[0][2], other[2], rtol=1e-10)


# ---------------------------------------------------------------------------------------------------------------------
# LEAN 2: the default sampler + snooker updates (theta_snooker > 0: test/multivariate_normal_tests.jl:58,
# Examples/Hierarchical_Example.jl:112, Examples/Run_LBA.jl) has kernel instances of its own -- replay, block masks,
# recombination, optimiser updates, the other proposal kinds and the trace compiled out, the snooker branches in
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", ["cfg1_resident", "cfg2_suffstat_resident", "cfg2_streaming_resident", "cfg3_chain", "cfg5_chain",
                                   "cfg1_blocks", "cfg2_blocks_snooker"])
def test_snooker_lean_instances_match_the_oracle(demc, orc, shape):
    """LEAN 2 also carries block updates (blocking_on / blocks: test/blocking_tests.jl:33-53): the last two shapes"""
    from demc_amd import workloads as W
    snk = 0.3
    if shape == "cfg1_blocks":        # test/blocking_tests.jl: blocks [[true,false],[false,true]], no snooker
        w, G, Np, kern, kw, snk = W.cfg1(), 8, 10, "k_propose<256,true,TAIL_OBS,true,2>", {}, 0.0
        w["masks"] = np.array([[1, 0], [0, 1]], np.uint8)
    elif shape == "cfg2_blocks_snooker":
        w, G, Np, kern, kw = W.cfg2(N=3000), 32, 64, "k_propose<256,true,TAIL_PREP_MFMA,true,2>", dict(loglike_mode=1)
        w["masks"] = np.array([[1] * 4 + [0] * 4, [0] * 4 + [1] * 4], np.uint8)
    elif shape == "cfg1_resident":    # Gaussian example, the whole update in the resident kernel's own observation loop
        w, G, Np, kern, kw = W.cfg1(), 8, 10, "k_propose<256,true,TAIL_OBS,true,2>", {}
    elif shape == "cfg2_suffstat_resident":
        w, G, Np, kern, kw = W.cfg2(N=3000), 32, 64, "k_propose<256,true,TAIL_PREP_MFMA,true,2>", dict(loglike_mode=1)
    elif shape == "cfg2_streaming_resident":
        w, G, Np, kern, kw = W.cfg2(N=3000), 32, 64, "k_propose<256,true,TAIL_PREP_MFMA,true,2,true>", dict(loglike_mode=0)
    elif shape == "cfg3_chain":
        w, G, Np, kern, kw = W.cfg3(N=1500, G=16), 16, 256, "k_propose<256,true,TAIL_PREP_MFMA,false,2>", dict(loglike_mode=0, fuse=2, geometry_groups=256)
    else:
        w, G, Np, kern, kw = W.cfg5(N=400, G=8, Np=16), 8, 16, "k_propose<256,true,TAIL_NONE,false,2>", {}
    ran = free_run(demc, orc, w, 8, [kern], G, Np, theta_exact=False, theta_snooker=snk,
                   lp_rtol=1e-5 if shape == "cfg5_chain" else 1e-9, **kw)
    assert "k_res_mvn" not in ran


# ---------------------------------------------------------------------------------------------------------------------
# k_res_obs (demc_resobs.hpp, round 6): the lean resident kernel of the DEFAULT sampler on the per-observation families -- what the
# reference's own gates run (test/gaussian_tests.jl:39-41, test/binomial_tests.jl, test/lognormal_race_tests.jl:40-42,
# Examples/Gaussian_Example.jl = BASELINE cfg1)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("family,G,Np,kw,beta", [
    ("cfg1", 4, 10, {}, 0.0),                       # BASELINE cfg1 as benchmarked: 4 x 10, D = 2, 50 observations; crossover only: theta bit for bit
    ("cfg1", 4, 10, {}, 0.1),                       # ... with mutation sweeps (device log / sincospi: 1e-10)
    ("gaussian", 4, 6, dict(N=50), 0.1),            # test/gaussian_tests.jl:39-41: Np = 6 (three movers a phase: most lanes of the pass idle)
    ("gaussian", 3, 40, dict(N=300), 0.1),          # 20 movers: two passes of 16 particles per colour phase, 19 observations a lane
    ("gaussian", 5, 9, dict(N=7), 0.3),             # odd group (4 / 5 movers), fewer observations than lanes
    ("binomial", 6, 12, dict(N=5), 0.1),            # test/binomial_tests.jl's model: D = 1, Beta prior, bounds [0, 1]
    ("binomial", 2, 64, dict(N=33), 0.5),           # 32 movers, mutation in half of the sweeps
    ("lnr", 4, 24, dict(N=100, na=3), 0.1),         # test/lognormal_race_tests.jl:40-42's shape: four groups of 24, the table in LDS
    ("lnr", 3, 8, dict(N=60, na=8), 0.1),           # eight accumulators: D = 9 -- three NOISE blocks, a Uniform prior on the last scalar
])
def test_lean_resident_kernel_of_the_per_observation_families(demc, orc, family, G, Np, kw, beta):
    """the default sampler on Gaussian / Binomial / LNR runs in k_res_obs<256> (sixteen lanes per particle, the whole update of a
    group resident over the iterations between two migrations): free-running against the oracle, migrations on, in and past burn-in
    -- every accept decision and particle id equal, theta bit for bit without mutation sweeps, log-posteriors to 1e-9 -- the
    kernel by name"""
    from demc_amd import workloads as W
    from conftest import make_problem
    if family == "cfg1":
        w = dict(W.cfg1(), G=G, Np=Np)
    else:
        prob = make_problem(family, np.random.default_rng(311), **kw)
        w = dict(prob, G=G, Np=Np, masks=None, engine={}, init=lambda P, rng_: prob["init"](P))
    ran = free_run(demc, orc, w, 24, [], G, Np, theta_exact=beta == 0.0, exact_kernels="k_res_obs<256>", beta=beta, alpha=0.3)
    assert ran == "k_res_obs<256>"


def test_lean_per_observation_kernel_equals_the_general_kernel(demc):
    """k_res_obs makes the proposals and decisions of the general kernel (trace = 1 selects k_propose): state, ids, accept flags and
    theta history bit for bit on Gaussian, Binomial and LNR across burn-in, mutation sweeps and migrations; log-densities to rounding
    (the sums over observations and prior terms run in another lane order)"""
    from conftest import make_problem, setup_engine
    for family, G, Np, kw in (("gaussian", 4, 10, dict(N=50)), ("binomial", 3, 20, dict(N=9)), ("lnr", 4, 24, dict(N=80, na=3))):
        prob = make_problem(family, np.random.default_rng(312), **kw)
        th0 = prob["init"](G * Np)
        outs, names = [], []
        for tr in (0, 1):
            e = demc.HipEngine(n_groups=G, Np=Np, D=prob["D"], n_rows=30, schedule=2, seed=11, alpha=0.3, burnin=12, trace=tr)
            setup_engine(e, prob)
            e.set_state(th0)
            e.step(1, 30)
            names.append(e.last_kernels())
            outs.append(e.get_history(0, 30) + e.get_state())
            e.close()
        assert names[0] == "k_res_obs<256>" and "k_propose" in names[1], names
        for i, (x, y) in enumerate(zip(*outs)):
            if x.dtype.kind == "f" and i in (2, 5):  # lp history, weights
                np.testing.assert_allclose(x, y, rtol=1e-11, err_msg=f"{family} array {i}")
            else:
                assert np.array_equal(x, y), f"{family} array {i}"


def _per_observation_cases(n, seed=20261009):
    """the default sampler on the per-observation families over random shapes: Gaussian / Binomial / LNR (2 .. 8 accumulators),
    groups of 4 .. 80 particles (one to three passes of sixteen per colour phase, odd halves), 3 .. 700 observations (fewer than
    lanes, ragged strides), mutation-heavy runs, in / leaving / past burn-in, migrations every other iteration"""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        fam = str(rng.choice(["gaussian", "gaussian", "binomial", "lnr", "lnr"]))
        c = dict(fam=fam, Np=int(rng.integers(4, 81)), G=int(rng.integers(1, 9)), N=int(rng.choice([3, 7, 16, 17, 50, 130, 700])),
                 na=int(rng.integers(2, 9)), beta=float(rng.choice([0.0, 0.1, 0.5])), burnin=int(rng.choice([0, 6, 100])),
                 alpha=float(rng.choice([0.1, 0.5])), seed=int(rng.integers(1, 2**31)))
        out.append(pytest.param(c, id=f"{i}-{fam}-Np{c['Np']}-G{c['G']}-N{c['N']}-na{c['na']}-b{c['burnin']}"))
    return out


def run_per_observation_case(demc, orc, c):
    from conftest import make_problem
    c = dict(c)
    fam, Np, G, N, na = c.pop("fam"), c.pop("Np"), c.pop("G"), c.pop("N"), c.pop("na")
    prob = make_problem(fam, np.random.default_rng(c["seed"] % 100000), **({"N": N, "na": na} if fam == "lnr" else {"N": N}))
    w = dict(prob, G=G, Np=Np, masks=None, engine={}, init=lambda P, rng_: prob["init"](P))
    if G == 1:
        c["alpha"] = 0.0  # (structs.jl:102-105: no migration with one group)
    return free_run(demc, orc, w, 14, [], G, Np, theta_exact=c["beta"] == 0.0, exact_kernels="k_res_obs<256>", **c)


@pytest.mark.parametrize("c", _per_observation_cases(8))
def test_per_observation_kernel_randomised_free_runs(demc, orc, c):
    """the generator above against the oracle: every accept decision and particle id equal, theta bit for bit without mutation
    sweeps (tests/free_run_sweep.py per_observation N runs the long form)"""
    run_per_observation_case(demc, orc, c)


def _direct_resident_cases(n, seed=20261010):
    """DIRECT mode on small populations (the streaming-resident lean kernel's DIRECT instances): D = 8 and 32, observation counts
    that leave ragged and empty last chunks, groups of 6 .. 128 particles (slices of 4 .. 16 proposal groups), mutation sweeps,
    in / leaving / past burn-in"""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        d = int(rng.choice([8, 8, 32]))
        c = dict(d=d, Np=int(rng.integers(6, 129)), G=int(rng.integers(2, 33 if d == 8 else 13)),
                 N=int(rng.choice([300, 1030, 2048, 2050, 4097] if d == 8 else [300, 1030, 2050])),
                 beta=float(rng.choice([0.0, 0.1, 0.5])), burnin=int(rng.choice([0, 4, 100])), alpha=float(rng.choice([0.1, 0.5])),
                 seed=int(rng.integers(1, 2**31)))
        out.append(pytest.param(c, id=f"{i}-d{d}-Np{c['Np']}-G{c['G']}-N{c['N']}-b{c['burnin']}"))
    return out


def run_direct_resident_case(demc, orc, c):
    from demc_amd import workloads as W
    c = dict(c)
    d, Np, G, N = c.pop("d"), c.pop("Np"), c.pop("G"), c.pop("N")
    w = W.cfg3(N=N, d=d, G=G, Np=Np)
    return free_run(demc, orc, w, 10, [], G, Np, theta_exact=c["beta"] == 0.0, loglike_mode=2, **c)


@pytest.mark.parametrize("c", _direct_resident_cases(8))
def test_direct_resident_kernel_randomised_free_runs(demc, orc, c):
    """the generator above against the oracle (whatever kernel the shape takes: the DIRECT streaming-resident instance where the
    observation chunk fits in LDS, else the K1 -> k_direct_mvn -> K3 chain): every accept decision and particle id equal"""
    ran = run_direct_resident_case(demc, orc, c)
    assert "direct" in ran, ran


def test_direct_resident_kernel_repeats_bit_for_bit(demc):
    """every output is a pure function of (inputs, seed): BASELINE cfg2 in DIRECT mode (eight workgroups per group handing their row
    partials over through epoch-tagged granules, 40 iterations, migrations on) three times on fresh handles -- state, weights, ids and
    the whole history bit for bit"""
    from demc_amd import workloads as W
    w = W.cfg2()
    G, Np, n_it = w["G"], w["Np"], 40
    th0 = w["init"](G * Np, np.random.default_rng(8))
    outs = []
    for _ in range(3):
        e = demc.HipEngine(n_groups=G, Np=Np, D=w["D"], n_rows=n_it, schedule=2, seed=99, burnin=20, alpha=0.3, loglike_mode=2)
        W.configure(e, w)
        e.set_state(th0)
        e.step(1, n_it)
        assert e.last_kernels() == "k_res_mvn<512,true,8,direct>"
        outs.append(e.get_history(0, n_it) + e.get_state())
        e.close()
    for other in outs[1:]:
        for x, y in zip(outs[0], other):
            assert np.array_equal(x, y)
