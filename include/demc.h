/*
 * demc.h -- C-ABI of libdemc_hip.so: the MI355X (gfx950) DE-MCMC hot path.
 *
 * This is the drop-in boundary for ONE path of itsdfish/DifferentialEvolutionMCMC.jl
 * (v0.7.10): the per-iteration particle update
 *     proposal (crossover / snooker / mutation) -> prior + log-likelihood -> Metropolis accept
 *     -> history store, plus the migration exchange,
 * i.e. what `groups = stepfun(model, de, groups)` does once per iteration in
 * src/main.jl:33-38 (step!/pstep!, src/main.jl:84-107).  A Julia maintainer binds these
 * symbols with @ccall; the Python host in
 * differentialevolutionmcmc.jl_amd/ binds them with ctypes.
 *
 * Conventions
 *   - every function returns int32 status (DEMC_OK == 0); no C++ exception crosses the ABI;
 *     demc_last_error(h) gives the message (valid until the next call on that handle);
 *   - "host" pointers are caller-allocated, caller-owned and only borrowed for the call;
 *     "dev" pointers are device (HBM) addresses on the handle's device, e.g. torch
 *     tensor.data_ptr(); the library never returns pointers to its own memory;
 *   - a handle is not thread-safe; one handle per GPU; calls enqueue on the handle's HIP
 *     stream and (unless stated) return after the stream has drained;
 *   - all arithmetic is IEEE double (the reference is Float64 throughout), ids int64,
 *     flags uint8;
 *   - particles are flattened: theta[P][D] row-major (particle-contiguous = Julia
 *     Matrix{Float64}(D, P)), P = n_groups*Np, slot s = g*Np + p  (structs.jl:202-208,
 *     main.jl:263-271).
 */
#ifndef DEMC_H
#define DEMC_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DEMC_VERSION 120 /* 0.1.2 */

enum {
    DEMC_OK = 0,
    DEMC_EINVAL = 1,       /* bad argument / inconsistent configuration */
    DEMC_EHIP = 2,         /* HIP runtime error (message has hipGetErrorString) */
    DEMC_ENOMEM = 3,
    DEMC_ERCCL = 4,        /* RCCL error on the handle's communicator (message has ncclGetErrorString) */
    DEMC_EUNSUPPORTED = 5  /* hook / model outside the registered family: no CPU fallback */
};

/* de.generate_proposal (structs.jl:71; crossover.jl:154-226) */
enum { DEMC_PROPOSAL_RANDOM_GAMMA = 0, DEMC_PROPOSAL_FIXED_GAMMA = 1, DEMC_PROPOSAL_VARIABLE_GAMMA = 2 };
/* de.sample (structs.jl:74): sample = current population (crossover.jl:138-140), resample = history (:113-121) */
enum { DEMC_PARTNER_CURRENT = 0, DEMC_PARTNER_HISTORY = 1 };
/* de.update_particle! (structs.jl:72): mh_update! / maximize! / minimize! (utilities.jl:201-226) */
enum { DEMC_UPDATE_MH = 0, DEMC_UPDATE_MAXIMIZE = 1, DEMC_UPDATE_MINIMIZE = 2 };
/* de.evaluate_fitness! (structs.jl:73): compute_posterior! / evaluate_fun! (utilities.jl:92-120) */
enum { DEMC_FITNESS_POSTERIOR = 0, DEMC_FITNESS_FUN = 1 };
/* Sweep schedule inside a group.  The reference sweeps particles one at a time in place
 * (crossover.jl:13-15), which a fused kernel cannot do; the device schedules are
 *   SYNCHRONOUS: all particles of a group propose from the sweep-start state;
 *   TWO_COLOUR : the first half of the group proposes with partners from the second half,
 *                then the halves swap (each half-update is a valid Metropolis-within-Gibbs step).
 *   SEQUENTIAL : the reference's own sweep -- particle k of every group moves after particles 0..k-1 of its group have
 *                been updated in place.  On the device this is Np dependent launches per sweep (all groups in
 *                parallel, one particle of each at a time, like one task per group in p_update!, main.jl:135-148):
 *                slow by construction; it exists so that, together with demc_set_replay, a run can follow the
 *                reference's schedule with the reference's random numbers. */
enum { DEMC_SCHED_SEQUENTIAL = 0, DEMC_SCHED_SYNCHRONOUS = 1, DEMC_SCHED_TWO_COLOUR = 2 };
/* How Gaussian-family likelihoods are evaluated.
 *   STREAMING : every proposal visits every observation, as model.loglike does
 *               (structs.jl:186; test/multivariate_normal_tests.jl:31-33), in the EXPANDED quadratic form
 *               sum_i x~_i'A x~_i - 2 y.(x~_i summed through the matrix cores) + N mu~'A mu~, y = A mu~, A = Sigma^-1, data centred:
 *               2 N d flop per proposal on v_mfma_f64 -- FP64 MFMA bound.  (After centring the streamed cross term
 *               sum_i y.x~_i equals y.(sum_i x~_i), i.e. rounding noise around zero: STREAMING = SUFFSTAT + that product;
 *               the data dependence of the result sits in the data-only constant and in N mu~'A mu~.)
 *   SUFFSTAT  : one pass over the data at demc_set_model, O(D^2) per proposal -- HBM bound.
 *   DIRECT    : the residual form sum_i |L^-1 (x_i - mu)|^2 term by term (whitened once: z_i = L^-1 x~_i at
 *               demc_set_model, m = L^-1 mu~ per proposal): 3 N d flop per proposal that do NOT separate into data-only
 *               and proposal-only factors -- FP64 vector bound; MvNormal families with d <= 64.
 * Same proposals and accept decisions in all three (log-densities agree to rounding); reported separately and labelled
 * (SURVEY.md 8d). */
enum { DEMC_LOGLIKE_STREAMING = 0, DEMC_LOGLIKE_SUFFSTAT = 1, DEMC_LOGLIKE_DIRECT = 2 };

/* Registered model family evaluated on device (user closures cannot run on the GPU). */
enum {
    DEMC_FAM_GAUSSIAN = 0,       /* theta=(mu,sigma); data x[N]; dims=[N]                 Examples/Gaussian_Example.jl:26-28 */
    DEMC_FAM_MVN_ISO = 1,        /* theta=(mu[d],sigma); data X[N][d]; dims=[N,d]         test/multivariate_normal_tests.jl:31-33 */
    DEMC_FAM_MVN_FULL = 2,       /* theta=mu[d]; data X[N][d]; dims=[N,d]; hyper=Sigma[d][d]   BASELINE cfg2/cfg3 */
    DEMC_FAM_BINOMIAL = 3,       /* theta=p; data=[n[N], k[N]]; dims=[N]                  test/binomial_tests.jl:15-17 */
    DEMC_FAM_HIER_BINOMIAL = 4,  /* theta=(mu_b0,sd_b0,b0[S]); data k[S]; dims=[S]; hyper=[n]  BASELINE cfg4 */
    DEMC_FAM_HIER_GAUSSIAN = 5,  /* theta=(mu_b0,sd_b0,b0[S],sigma); data Y[S][n]; dims=[S,n]  Examples/Hierarchical_Example.jl:36-44 */
    DEMC_FAM_LBA = 6,            /* theta=(nu[A],A,k,tau); data=[choice[N], rt[N]]; dims=[N,A] Examples/Run_LBA.jl:33-37 */
    DEMC_FAM_LNR = 7,            /* theta=(nu[A],tau); data=[choice[N], rt[N]]; dims=[N,A]; hyper=[sigma] test/lognormal_race_tests.jl:9-12 */
    DEMC_FAM_RASTRIGIN = 8,      /* objective only; dims=[]                                test/optimization_tests.jl:15-23 */
    DEMC_FAM_USER = 100          /* per-observation log-density supplied as HIP source, see demc_set_model_source */
};

/* Per-scalar prior table = the registered form of model.prior_loglike (structs.jl:185). */
enum {
    DEMC_PRIOR_FLAT = 0,
    DEMC_PRIOR_NORMAL = 1,      /* Normal(a, b) */
    DEMC_PRIOR_HALFCAUCHY = 2,  /* truncated(Cauchy(a, b), 0, Inf) */
    DEMC_PRIOR_UNIFORM = 3,     /* Uniform(a, b) */
    DEMC_PRIOR_BETA = 4,        /* Beta(a, b) */
    DEMC_PRIOR_NORMAL_REF = 5,  /* Normal(a, theta[ref]) */
    DEMC_PRIOR_GAMMA = 6,       /* Gamma(shape a, scale b) */
    DEMC_PRIOR_EXPONENTIAL = 7, /* Exponential(scale b)      (Distributions.jl parameterisation) */
    DEMC_PRIOR_LOGNORMAL = 8,   /* LogNormal(a, b) */
    DEMC_PRIOR_CAUCHY = 9       /* Cauchy(a, b) */
};

/* POD mirror of the DE keyword constructor (structs.jl:80-131). */
typedef struct demc_config {
    int32_t n_groups;        /* groups owned by THIS handle (shard) */
    int32_t Np;              /* particles per group */
    int32_t D;               /* scalar parameters after flattening nested Theta */
    int32_t n_blocks;        /* 0: blocking_on(de) == false */
    int64_t burnin;
    int64_t n_initial;
    int64_t n_rows;          /* history rows = n_iter + n_initial (utilities.jl:29-34) */
    double alpha;            /* migration probability      (main.jl:85)      */
    double beta;             /* mutation probability       (main.jl:200)     */
    double eps;              /* crossover noise            (crossover.jl:166)*/
    double sigma;            /* mutation sd                (mutation.jl:15)  */
    double kappa;            /* recombination              (crossover.jl:302)*/
    double theta_snooker;    /* snooker probability        (crossover.jl:31) */
    int32_t proposal_kind;
    int32_t partner_kind;
    int32_t update_kind;
    int32_t fitness_kind;
    int32_t schedule;
    int32_t store_history;   /* 1: keep samples/accept/lp history on device (utilities.jl:161-180) */
    int32_t group_offset;    /* global index of this shard's first group */
    int32_t n_groups_total;  /* groups over all shards (0 -> n_groups) */
    uint64_t seed;           /* Philox key */
    int32_t device_id;
    int32_t loglike_mode;
    int32_t trace;           /* 1: keep the per-slot diagnostic trace readable by demc_get_trace (tests) */
    int32_t fuse;            /* 0: auto -- when the likelihood allows (MvNormal in SUFFSTAT mode, Gaussian / Binomial with
                                few observations) the whole update runs inside the proposal kernel, and with two_colour
                                and the group in LDS that kernel stays resident: one launch runs every iteration up to
                                the next migration; 1: never fuse (proposal, likelihood, accept as separate kernels);
                                2: fuse, but one launch per colour phase (no resident form).  Same samples in all three; a
                                log-density can differ in its last bits between them (a particle may be split over a
                                different number of lanes, which changes the order of the sums). */
    int32_t geometry_groups; /* 0: size the kernels' lane geometry (lanes per particle) for this shard's n_groups.  > 0: size
                                it as if the handle owned this many groups -- shards of one population then all use the
                                geometry of the unsharded run, and an N-shard run reproduces the 1-shard run bit for bit
                                (without it the log-densities can differ in their last bits, see `fuse`). */
    int32_t reserved0;
} demc_config;

typedef struct demc_handle demc_handle;

int32_t demc_version(void);
int32_t demc_create(const demc_config* cfg, demc_handle** out);
int32_t demc_destroy(demc_handle* h);
/* The message of the last failed call on the handle -- or, after a SUCCESSFUL call, a line that starts with "note:" when a
 * documented deviation is in force (demc_create on a shard with history partners: the partner pool is the shard's own history;
 * demc_set_model / demc_set_blocks: a scratch buffer that could not be allocated and the slower path taken instead). */
const char* demc_last_error(demc_handle* h);
/* Enqueue on an existing HIP stream (hipStream_t passed as void*; NULL -> the handle's own stream). */
int32_t demc_set_stream(demc_handle* h, void* hip_stream);

/* replaces the closure pair built by DEModel(...) (structs.jl:176-189) */
int32_t demc_set_model(demc_handle* h, int32_t family, const double* host_data, const int64_t* dims, int32_t ndims,
                       const double* host_hyper, int32_t nhyper);
/* Plug-in for models outside the registered family: the reference's `loglike(data, theta...)` closures are, in every
 * test and example, sums of per-observation log-densities (e.g. `sum(logpdf.(Normal(mu, sigma), data))`,
 * Examples/Gaussian_Example.jl:26-28).  `hip_source` must define
 *     __device__ double demc_user_obs(const double* theta, int D, const double* data, long long N, long long i,
 *                                     const double* hyper, int nhyper);
 * = the log-density of observation i (data is the flat host_data array, N = dims[0]); log-likelihood = sum over i.
 * The source is compiled for gfx950 at this call (hiprtc) into the same thread-per-proposal streaming kernel the
 * registered scalar-data families use; compile errors are returned through demc_last_error. */
int32_t demc_set_model_source(demc_handle* h, const char* hip_source, const double* host_data, const int64_t* dims,
                              int32_t ndims, const double* host_hyper, int32_t nhyper);
/* Whole-row plug-in: a log-likelihood that is NOT a plain sum of per-observation terms of a flat data array, and / or a
 * prior the per-scalar table cannot express -- what DEModel's arbitrary prior_loglike / loglike closures (structs.jl:176-189)
 * look like in the reference's own hierarchical example (Examples/Hierarchical_Example.jl:26-44: the prior of beta_s depends on
 * another parameter, the likelihood indexes the data by subject).  `hip_source` must define
 *     __device__ double demc_user_loglike_row(const double* theta, int D, const double* data, const long long* dims, int ndims,
 *                                             const double* hyper, int nhyper, int lane, int n_lanes);
 * and, with flags & DEMC_USER_HAS_PRIOR,
 *     __device__ double demc_user_prior_row(const double* theta, int D, const double* hyper, int nhyper, int lane, int n_lanes);
 * Each proposal row is evaluated by ONE WORKGROUP of n_lanes = 256 lanes: a function returns the share of lane `lane`
 * (typically `for (s = lane; s < S; s += n_lanes) acc += term(s);` -- or everything on lane 0 and 0.0 elsewhere); the shares
 * are summed in a fixed order.  The prior value is added to what the per-scalar table of demc_set_priors gives (flat by
 * default; bounds still come from demc_set_bounds) and is left out under evaluate_fun! (utilities.jl:113-120).
 * data = the flat host_data array with shape dims[0..ndims) (ndims <= 8). */
#define DEMC_USER_HAS_PRIOR 1
int32_t demc_set_model_source_row(demc_handle* h, const char* hip_source, const double* host_data, const int64_t* dims,
                                  int32_t ndims, const double* host_hyper, int32_t nhyper, int32_t flags);
int32_t demc_set_priors(demc_handle* h, const int32_t* kind, const double* a, const double* b, const int32_t* ref);
/* de.bounds flattened to one (lo,hi) per scalar; +-Inf allowed (utilities.jl:70-78) */
int32_t demc_set_bounds(demc_handle* h, const double* lo, const double* hi);
/* de.blocks flattened to n_blocks x D byte masks (crossover.jl:336-352) */
int32_t demc_set_blocks(demc_handle* h, const uint8_t* masks, int32_t n_blocks);

/* sample_init (main.jl:263-271): particles as flat rows; weight == NULL -> evaluate_fitness! on device
 * (utilities.jl:19); id == NULL -> group_offset*Np + slot. */
int32_t demc_set_state(demc_handle* h, const double* theta, const double* weight, const int64_t* id);
int32_t demc_get_state(demc_handle* h, double* theta, double* weight, int64_t* id);
/* initialize_samples (utilities.jl:35-39): prior draws for history rows [row0, row0+nrows), [nrows][P][D] by slot */
int32_t demc_set_history_rows(demc_handle* h, int64_t row0, int64_t nrows, const double* theta_rows);
/* Raw history rows [row0,row1), keyed by SLOT, plus the particle id that occupied the slot at that
 * row; the host re-keys by id (samples[iter, :, p.id], utilities.jl:170-180).  Any output may be NULL. */
int32_t demc_get_history(demc_handle* h, int64_t row0, int64_t row1, double* theta_hist, uint8_t* accept_hist,
                         double* lp_hist, int64_t* id_hist);

/* bundle_samples' gather (main.jl:232-241) done on the device: history rows [row0,row1) re-keyed by particle id and laid
 * out as the value array of the Chains object, n = row1 - row0, j < D: parameter j, j = D: acceptance, j = D+1: lp.
 *   layout 0: out[(id*(D+2) + j)*n + (row-row0)]   = Julia Array{Float64,3}(n, D+2, P), column-major (iteration fastest)
 *   layout 1: out[((row-row0)*(D+2) + j)*P + id]   = C order [n][D+2][P]
 * One kernel + one device-to-host copy; single-shard handles only (ids must be local). */
int32_t demc_export_chains(demc_handle* h, int64_t row0, int64_t row1, int32_t layout, double* host_out);

/* n_iters of step!/pstep! (main.jl:84-107) starting at de.iter == iter0 (1-based, n_initial included):
 * migration coin + exchange, update of every group, store.  On a sharded handle (n_groups_total > n_groups) the exchange is
 * the one collective of the path and needs the handle's communicator (demc_comm_init below); without one the call fails
 * at the first migration. */
int32_t demc_step(demc_handle* h, int64_t iter0, int32_t n_iters);
/* the same, ENQUEUED ONLY on the handle's stream; demc_synchronize drains it and reports what went wrong meanwhile (for a
 * host that drives several handles from one thread, like one task per group in p_update!, main.jl:135-148) */
int32_t demc_step_async(demc_handle* h, int64_t iter0, int32_t n_iters);
int32_t demc_synchronize(demc_handle* h);
/* update! + store_samples! only (main.jl:86-87): for drivers that run the migration exchange themselves */
int32_t demc_update(demc_handle* h, int64_t iter0, int32_t n_iters);

/* migration! (migration.jl:11-19) in two halves around the one exchange.
 * demc_migration_due : the alpha coin of iteration `iter` (main.jl:85) -- pure function of (seed, iter).
 * demc_migration_pack: select_particle for each local group (migration.jl:64-70); writes
 *                      [n_groups][D+3] doubles (slot, theta[D], weight, id) to dev_rows.
 * demc_migration_apply: select_groups + shift_particles! (migration.jl:31-35, :84-91) given the rows of ALL
 *                      groups [n_groups_total][D+3] (after an all-gather); only local groups are written. */
int32_t demc_migration_due(const demc_config* cfg, int64_t iter);
int32_t demc_migration_pack(demc_handle* h, int64_t iter, double* dev_rows);
int32_t demc_migration_apply(demc_handle* h, int64_t iter, const double* dev_all_rows);
/* The same two halves ENQUEUED ONLY (no stream drain): for a driver whose collective runs stream-ordered on the handle's
 * stream (demc_set_stream with the stream the collective is issued from) -- pack -> all-gather -> apply -> the next
 * demc_update then need no host synchronisation in between.  dev pointers must stay valid until the stream has passed. */
int32_t demc_migration_pack_async(demc_handle* h, int64_t iter, double* dev_rows);
int32_t demc_migration_apply_async(demc_handle* h, int64_t iter, const double* dev_all_rows);
/* Per-group-asynchronous migration: groups the migration of an iteration did not select are untouched by the
 * exchange, so their update need not wait for it.
 *   demc_migration_groups     : select_groups' ordered sub-group of iteration `iter` (GLOBAL group indices, sel has room for
 *                               n_groups_total entries) -- a pure function of (seed, iter), what k_mig_apply derives on the device.
 *   demc_update_groups_async  : update! + store_samples! (like demc_update) for a SUBSET of this handle's groups (local indices),
 *                               enqueued on the handle's stream WITHOUT draining it.  Refused while a migration sub-group is
 *                               replayed (demc_migration_groups does not see the replay), and for n_iters > 1 with history
 *                               partners (resample, crossover.jl:113-124: the cells of a history row come from every group, so
 *                               the other groups must have made iteration t before anybody makes t + 1).
 * A sharded driver enqueues  pack -> [all-gather on a side stream] ; update(groups not selected) ; wait for the gather ;
 * apply ; update(selected groups)  -- the collective overlaps the update of the unselected groups (distributed.py). */
int32_t demc_migration_groups(const demc_config* cfg, int64_t iter, int32_t* sel, int32_t* n_sel);
int32_t demc_update_groups_async(demc_handle* h, int64_t iter0, int32_t n_iters, const int32_t* local_groups, int32_t n);
/* ---- The one collective behind the boundary: an RCCL communicator owned by the handle. ----
 * migration! (migration.jl:11-19, called from step!/pstep!, main.jl:85,103) exchanges one candidate particle per group
 * between groups; with groups sharded over GPUs that is ONE all-gather of [n_groups][D+3] doubles per rank and migration
 * event (xGMI, RCCL), everything else of the iteration is shard-local (groups never interact inside update!/p_update!,
 * main.jl:135-167).  One process per GPU:
 *   rank 0:      demc_comm_unique_id(id)            -- ncclGetUniqueId; the host carries the 128 bytes to the other ranks by
 *                                                      whatever it has (MPI.bcast, Distributed.jl, a file, a pipe)
 *   every rank:  demc_create(cfg with n_groups = G, group_offset = rank*G, n_groups_total = world*G, device_id = local GPU)
 *                demc_comm_init(h, id, rank, world) -- ncclCommInitRank on the handle's device (collective: all ranks call it)
 *   then         demc_step(h, iter0, n)             -- the whole sharded iteration: alpha coin (same on every rank), pack ->
 *                                                      ncclAllGather on the handle's stream -> apply, update, store; an
 *                                                      N-rank run makes the draws and decisions of the 1-rank run
 * demc_comm_set_overlap(h, 1): per-group-asynchronous migration -- the all-gather runs on a side stream while
 *   the groups the exchange did not select are updated; the selected groups wait for it.  Same draws and decisions; a
 *   log-density can differ in its last bits where the subset update takes another kernel form (MvNormal STREAMING on small
 *   populations: the subset update does not use the streaming-resident form).  Ignored while a replay is set, and with
 *   history partners (no subset of the groups may run ahead of the others there: the plain stream-ordered exchange is taken).
 * demc_migration_exchange[_async]: pack -> all-gather -> apply of iteration `iter` alone, for a host that calls demc_update
 *   itself (the `_async` form only enqueues).
 * demc_comm_allreduce: host doubles reduced over the ranks (op 0 sum, 1 max, 2 min; n = 0: a barrier) -- what a host without
 *   MPI needs for the reference's own reductions at the end (timing, posterior means); one rank: no-op.
 * demc_comm_stats: out3 = (world, rank, all-gathers issued so far).
 * Failures of RCCL come back as DEMC_ERCCL with ncclGetErrorString in demc_last_error. */
#define DEMC_COMM_ID_BYTES 128
int32_t demc_comm_unique_id(void* id_out, int32_t nbytes);
int32_t demc_comm_init(demc_handle* h, const void* unique_id, int32_t rank, int32_t world);
int32_t demc_comm_destroy(demc_handle* h);
int32_t demc_comm_set_overlap(demc_handle* h, int32_t on);
int32_t demc_migration_exchange(demc_handle* h, int64_t iter);
int32_t demc_migration_exchange_async(demc_handle* h, int64_t iter);
int32_t demc_comm_allreduce(demc_handle* h, double* host_inout, int32_t n, int32_t op);
int32_t demc_comm_stats(demc_handle* h, int64_t* out3);

/* ---- The same for a SINGLE-PROCESS host (one Julia task driving all GPUs of the node; SURVEY 8b "demc_create_multi"). ----
 * cfg describes the WHOLE population (n_groups = all groups, group_offset = 0); shard r owns groups [r*G, (r+1)*G),
 * G = n_groups / n_shards, on device device_ids[r] (NULL: 0..n_shards-1).  Distinct devices get one RCCL communicator each
 * (ncclCommInitAll) and the exchange is one grouped ncclAllGather; shards that share a device (several shards per GPU)
 * hand their rows over by device-to-device copies.  Every shard is sized with the lane geometry of the whole population
 * (demc_config.geometry_groups) -- kernel form and observation-chunk count of the STREAMING likelihood included -- so the set
 * reproduces a single handle of n_groups groups bit for bit.  Shards that share a device run on ONE stream (the first such
 * shard's; do not give them streams of their own with demc_set_stream): the streaming-resident kernels assume the chip to
 * themselves; demc_set_stream on a shard of a built set -- of ANY built set, a one-shard set included: the set's event
 * ordering is wired to the streams it was built with -- is refused, DEMC_EINVAL).  demc_comm_init / _destroy / _set_overlap are
 * refused on a shard (DEMC_EINVAL): its communicator belongs to the set.
 * History partners (partner_kind = DEMC_PARTNER_HISTORY, `resample`, crossover.jl:113-124) draw their cells from the history of
 * ALL particles of the population; a shard holds the history of its own groups only.  A set of more than one shard could
 * therefore not reproduce the single handle it stands for, and demc_create_multi refuses the combination (DEMC_EUNSUPPORTED).
 * On the one-process-per-GPU road (group_offset / n_groups_total + demc_comm_init) DE-MC_Z runs with the SHARD-LOCAL pool --
 * rows 1:(iter-1) x the rank's own particles -- the deviation SURVEY 8(e) names; it is not bit-equal to an unsharded run.
 *   demc_multi_shard(m, r) : the shard's handle, for the per-shard calls -- demc_set_model / _priors / _bounds / _blocks (the
 *                            same on every shard), demc_set_state / demc_get_state / demc_get_history with the shard's own
 *                            P/n_shards particles.  Shards are destroyed with the set.
 *   demc_multi_step        : n_iters of step! over the set: every shard's work is enqueued before any shard is waited for. */
typedef struct demc_multi demc_multi;
int32_t demc_create_multi(const demc_config* cfg, int32_t n_shards, const int32_t* device_ids, demc_multi** out);
int32_t demc_destroy_multi(demc_multi* m);
const char* demc_multi_last_error(demc_multi* m);
int32_t demc_multi_size(demc_multi* m);
demc_handle* demc_multi_shard(demc_multi* m, int32_t r);
int32_t demc_multi_step(demc_multi* m, int64_t iter0, int32_t n_iters);

/* shift_particles! (migration.jl:84-91) with a HOST-drawn plan: for every k, slot dst_slot[k] receives the row
 * (theta, weight, id) that slot src_slot[k] held BEFORE the call -- all reads precede all writes, so a cycle is a
 * rotation.  For a caller that keeps migration!'s own random choices (select_groups / select_particles,
 * migration.jl:31-35, :48-53) on the host, e.g. the Julia side replaying its task-local RNG, and drives the rest
 * through demc_update.  Slots are local (0 <= slot < n_groups*Np); dst slots must be distinct.  Arrays are host memory. */
int32_t demc_apply_migration(demc_handle* h, const int32_t* src_slot, const int32_t* dst_slot, int32_t n);
/* the weights alone ([P] doubles): what select_particles' inverse-weight draw needs on the host (migration.jl:64-70) */
int32_t demc_get_weights(demc_handle* h, double* weight);

/* compute_posterior! / evaluate_fun! for n arbitrary host rows [n][D] (utilities.jl:92-120) */
int32_t demc_logpost(demc_handle* h, const double* theta, int64_t n, double* out);

/* Test/diagnostic mode: the last sweep's proposals, proposal weights, snooker adjustments,
 * partner indices [P][4] = (kind 0 DE / 1 snooker / 2 mutation, i0, i1, i2) and accept flags. */
int32_t demc_get_trace(demc_handle* h, double* proposal, double* w_prop, double* log_adj, int32_t* idx,
                       uint8_t* accepted);

/* Diagnostic: the kernel instances the last update launched on this handle, e.g. "k_res_mvn<512,false,32>" or
 * "k_propose<256,true,TAIL_PREP_MFMA,false,true> + k_cross_mfma<8,4> + k_accept_store" (template arguments as in csrc/:
 * workgroup, LDS tile, fused tail, resident, plain[, streaming]) -- so that a parity test can say which instance it
 * compared with the oracle.  Empty before the first update. */
int32_t demc_last_kernels(demc_handle* h, char* out, int32_t nbytes);

/* Test mode: caller-supplied random numbers in place of the library's addressed Philox draws, so that a
 * host that owns the reference's RNG (Julia's task-local stream) can feed ITS draws -- in the order of SURVEY Appendix A
 * -- and compare index bookkeeping and proposals bit for bit.  All pointers are HOST arrays, copied at the call; a NULL
 * member, a NaN uniform or a negative index means "draw as usual".  The replay stays in force for every following sweep
 * until demc_set_replay(h, NULL).  While it is set the general (unfused-plan) form of the kernels runs.
 * Partner rows are 0-based positions INSIDE the particle's group and are the caller's responsibility to keep legal for
 * the schedule (two_colour: rows of the resting half); they are range-checked against [0, Np). */
typedef struct demc_replay {
    const double* u_step;        /* [1]       alpha coin of step! (main.jl:85)                                              */
    const double* u_group;       /* [n_groups] beta coin of mutate_or_crossover! for each local group (main.jl:200)         */
    const double* u_part;        /* [P][5]    snooker coin (crossover.jl:31), select_base uniform (:156), gamma_1 uniform
                                              (:162, snooker gamma :249), gamma_2 uniform (:164), accept uniform (utilities.jl:57) */
    const int64_t* partner;      /* [P][3]    DE: (Pm, Pn, Pb) (crossover.jl:156-160); snooker: (Pz, Pm, Pn) (:241)         */
    const double* u_noise;       /* [P][D]    uniforms behind b ~ Uniform(-eps, eps), Theta order (crossover.jl:166)        */
    const double* z_noise;       /* [P][D]    standard normals of mutation! (mutation.jl:15; utilities.jl:291-306)          */
    const double* u_recomb;      /* [P][D]    recombination! uniforms (crossover.jl:308,318)                                */
    const int32_t* mig_groups;   /* [n_mig_groups] select_groups' ordered sub-group, GLOBAL group indices (migration.jl:31-35) */
    int32_t n_mig_groups;        /* 0: draw the sub-group as usual                                                          */
    int32_t reserved;
    const int64_t* mig_particle; /* [n_groups] select_particle's pick for each local group (migration.jl:64-70)             */
} demc_replay;
int32_t demc_set_replay(demc_handle* h, const demc_replay* replay);

/* Device time (ms) spent in each kernel class since the last reset, measured with HIP events on the handle's
 * stream when timing is enabled: out[0]=propose (K1), [1]=likelihood preparation, [2]=main likelihood kernel (K2),
 * [3]=accept/store (K3), [4]=migration; the number of timed launch groups of each class in out[5..9]. */
int32_t demc_timing_enable(demc_handle* h, int32_t on);
int32_t demc_timing_read(demc_handle* h, double* out10, int32_t reset);
/* The shader clock a compute-bound likelihood kernel -- the DIRECT MvNormal kernel (DEMC_LOGLIKE_DIRECT) or the LBA's wave-per-proposal
 * kernel: FP64 vector-pipe work, whose rate scales with the clock -- held during its LAST launch with timing enabled: every workgroup stamps s_memtime (shader cycles), s_memrealtime (100 MHz
 * reference) and its CU as it ends; per CU the clock is the shader cycles over the reference ticks between the first and the last
 * workgroup to finish there.  out[0] = median over the CUs in MHz, [1] = min, [2] = max, [3] = CUs that reported (0: no such
 * launch ran, or it was a single round of workgroups: nothing to difference).  Diagnostic, like demc_timing_read: the chip lowers its
 * clock under load and devices differ, so a measured TFLOP/s is quoted with the clock it was measured at (replaces nothing in
 * the reference). */
int32_t demc_timing_clock(demc_handle* h, double* out4);

#ifdef __cplusplus
}
#endif
#endif
