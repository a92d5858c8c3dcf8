// demc_hip.cpp -- host runtime and C-ABI (include/demc.h) of libdemc_hip.so.
//
// The handle owns every device buffer (particles, proposals, partial sums, history, model data laid out
// for the kernels) and one HIP stream; demc_step() enqueues, per iteration, the launch schedule that
// step!/update!/block_update!/mutate_or_crossover! (main.jl:84-207) imply:
//     [migration pack + apply]  then per sweep (block) and per colour phase:  K1 propose -> K2 loglike -> K3 accept/store
// All per-iteration randomness is addressed by (seed, iteration, entity) inside the kernels, so the host
// loop needs no device->host traffic: the only host-side draw is the alpha coin (pure function of seed, iter).
//
// There is NO CPU fallback in this file: every compute entry point launches HIP kernels or fails.
#include "../../include/demc.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <hip/hiprtc.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <vector>

#define DEMC_K1_EXTERN  // the k_propose instances live in demc_k1_{phase,res,stream}.cpp
#include "demc_kernels.hpp"
#define DEMC_LONGROW_EXTERN  // k_longrow<256 / 512> are instantiated in demc_longrow.cpp
#include "demc_longrow.hpp"
#define DEMC_FROZEN_EXTERN  // ... k_frozen_sweep in demc_frozen.cpp, k_res_mvn in demc_resmvn.cpp (compiled side by side)
#include "demc_frozen.hpp"
#define DEMC_RESMVN_EXTERN
#include "demc_resmvn.hpp"
#define DEMC_RESOBS_EXTERN
#include "demc_resobs.hpp"

using namespace demc;

#ifndef DEMC_ARCH_STR
#define DEMC_ARCH_STR "gfx950"  // the Makefile passes its ARCH, so that the JIT-compiled plug-in matches the library
#endif
constexpr size_t kMaxDynLds = 150 * 1024;  // of the 160 KB per CU; the rest covers the kernels' static __shared__

namespace {

struct Timed {
    hipEvent_t a, b;
    int cls;
    bool started = false;  // a kernel of the bracket carries the events (LAUNCH_T)
};

}  // namespace

struct demc_handle {
    demc_config c{};
    long long P = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // state
    double *theta = nullptr, *weight = nullptr, *prop = nullptr, *prop_prior = nullptr, *prop_adj = nullptr;
    double *tr_w = nullptr, *partial = nullptr, *aux = nullptr;
    DimTab* dimtab = nullptr;
    std::vector<DimTab> h_tab;  // host copy: bounds and priors arrive in separate calls
    DimSeg* dimseg = nullptr;   // run-length form of the table (kMaxDimSeg entries)
    int n_seg = 0;              // 0: more segments than kMaxDimSeg, the kernels read dimtab
    int seg_start[kMaxDimSeg] = {0};
    unsigned seg_plain = 0;     // segments with a flat / Normal / Normal(a, theta[ref]) prior
    struct MaskRuns { int n = 0; unsigned in = 0; int start[kMaxMaskRun] = {0}; };
    std::vector<MaskRuns> mask_runs;  // per block sweep: the mask run-length encoded (n = 0: too many runs)
    double *hist = nullptr, *lp_hist = nullptr, *mig_rows = nullptr, *scratch_theta = nullptr, *scratch_w = nullptr;
    long long* id = nullptr;
    unsigned char *prop_oob = nullptr, *tr_acc = nullptr, *masks = nullptr, *acc_hist = nullptr;
    int *tr_idx = nullptr, *id_hist = nullptr;
    // model
    int family = -1;
    long long N = 0;
    int d = 0, n_acc = 0, dpad = 0;
    int n_tiles = 0;
    int ks_t = 0, n_kpass = 0;  // MFMA k-steps per pass (template) and passes over the dimensions
    int dp_direct = 0;          // DIRECT mode: padded length of a whitened observation row (8 / 16 / 32 / 64)
    int direct_wgs_per_cu = 0;  // ... resident workgroups of its kernel per CU (asked once)
    int obs_wgs_per_cu = 0;     // the same for k_obs_loglike
    int lba_wave_wgs_per_cu = 0;  // ... and k_lba_wave
    bool lba_wide_ok = false;   // k_lba_loglike's 140 KB of dynamic LDS were granted (demc_create)
    double *data = nullptr, *Ainv = nullptr, *Ypad = nullptr, *Xf = nullptr, *sx = nullptr, *xbar = nullptr;
    size_t data2_off = 0;
    // user plug-in (demc_set_model_source): JIT-compiled module, kernel and its hyper-parameters
    hipModule_t user_module = nullptr;
    hipFunction_t user_kernel = nullptr;
    double* user_hyper = nullptr;
    int user_nhyper = 0;
    bool user_row = false;        // whole-row plug-in (demc_set_model_source_row): one workgroup per proposal
    bool user_has_prior = false;  // ... whose source also defines demc_user_prior_row
    long long* user_dims = nullptr;
    int user_ndims = 0;
    double c0 = 0, c1 = 0, c2 = 0;
    int partial_cap = 64;
    int lpp = 1;
    int tile_in_lds = 0;
    size_t k1_lds = 0, k1_tile_bytes = 0, k1_scr_bytes = 0;
    bool hier_scr = false;  // hierarchical family whose theta' scratch fits in LDS
    bool res_ok = false;  // resident K1 (plan_resident)
    int res_lpp = 0, res_wg = 0, res_scr_doubles = 0;
    size_t res_lds = 0;
    // streaming-resident form (plan_stream): the MvNormal observation stream inside the resident kernel
    bool st_ok = false;
    bool tf_cheap_obs = false;  // set_tail_flags' "the likelihood is cheap enough for K1" of the last call (launch_phase reads it)
    // k_frozen_sweep: launch order of the groups per (iteration, block sweep) of the current demc_step call -- groups whose
    // mutation coin fires (main.jl:199-207) first: their workgroups move the whole row and take twice as long, and a slow
    // workgroup that starts last ends the launch.  A hint only: any order gives the same results.
    // by-product snapshot (KParams::snap_theta / snap_weight): the rows and weights a frozen sweep over the whole population ended
    // with, valid as the sweep-start snapshot of sweep `snap2_sweep` of iteration `snap2_iter` (and of nothing else)
    double *snap2 = nullptr, *snap2_w = nullptr;
    int64_t snap2_iter = -1;
    int snap2_sweep = -1;
    int* frozen_order_d = nullptr;
    size_t frozen_order_cap = 0;
    int* frozen_order_pin = nullptr;        // pinned staging of the table (a truly asynchronous copy: nothing is drained)
    size_t frozen_order_pin_cap = 0;
    hipEvent_t frozen_order_ev = nullptr;   // the last copy out of the staging buffer has been consumed
    long long frozen_iter0 = 0;
    int frozen_iters = 0;
    int st_C = 0, st_nact_max = 0, st_rows = 0, st_x_lds = 0, st_chunk_tiles = 0, st_lpp = 0, st_scr_doubles = 0, st_wg = 512;
    // the lean streaming kernel's own cut of the observation tiles (plan_lean): st_C, or twice that with two workgroups per CU
    int lean_st_C = 0, lean_st_chunk_tiles = 0, lean_st_x_lds = 0, lean_st_occ = 1;
    size_t st_lds = 0;
    unsigned long long* st_gran = nullptr;  // hand-over granules (device)
    unsigned* st_err = nullptr;             // time-out flag (host-mapped, zero-copy)
    int n_cus = 0;
    bool bracket_open = false;  // timing: between tick(begin) and tick(end) of a bracket whose kernels carry the events
    size_t lr_two_lds = 0;     // long-row kernel: the dynamic LDS size lr_two_fit was asked for
    bool lr_two_fit = false;   // ... two 256-thread workgroups with that much LDS fit on a CU
    int ainv_lds = 1;
    // lean resident kernel of the default sampler on MvNormal-full (demc_resmvn.hpp): geometry for SUFFSTAT / STREAMING
    bool lean_ok = false, lean_stream_ok = false, lean_hist_ok = false;  // (lean_hist: DE-MC_Z past burn-in, k_res_mvn<..., HIST>)
    bool st_dir_geo = false;   // plan_stream's geometry is valid for the DIRECT likelihood (k_res_mvn<..., DIR>; st_ok stays false)
    bool lean_direct_ok = false;  // ... and the lean kernel's DIRECT streaming-resident instance serves this model
    bool lean_obs_ok = false;  // lean resident kernel of the default sampler on the per-observation families (demc_resobs.hpp)
    size_t lean_obs_lds = 0;
    int lean_wg = 0;
    size_t lean_lds = 0, lean_stream_lds = 0, lean_hist_lds = 0;
    // update of a subset of the groups (demc_update_groups_async): two device lists used alternately, and the one in force
    // Each call takes the next of kGlistRing slots: a pinned host copy of the list, a device copy, and an event that marks
    // the host copy as consumed.  The device copy is refilled by a copy ON THE HANDLE'S STREAM, i.e. behind every kernel that
    // still reads its previous content.
    static constexpr int kGlistRing = 8;
    int* glist_buf[kGlistRing] = {};
    int* glist_pin[kGlistRing] = {};
    hipEvent_t glist_ev[kGlistRing] = {};
    int glist_next = 0;
    const int* cur_glist = nullptr;
    int cur_ng = 0;
    std::string err;
    // replay (demc_set_replay): device copies of the caller's draws
    double *rp_group = nullptr, *rp_part = nullptr, *rp_noise = nullptr, *rp_znoise = nullptr, *rp_recomb = nullptr;
    long long *rp_partner = nullptr, *rp_mig_particle = nullptr;
    int* rp_mig_groups = nullptr;
    int rp_n_mig = 0;
    bool rp_active = false, rp_has_step = false;
    double rp_u_step = 0.0;
    int geo_groups = 0;  // groups the lane geometry is sized for (demc_config.geometry_groups, else n_groups)
    int hist_ld = 0;     // doubles between consecutive history cells (KParams::hist_ld)
    // the one collective of the path: an RCCL communicator owned by the handle (demc_comm_init), or lent by the
    // single-process multi-GPU set the handle is a shard of (demc_create_multi)
    ncclComm_t comm = nullptr;
    bool own_comm = false;
    int comm_rank = 0, comm_world = 1;
    bool comm_overlap = false;        // demc_comm_set_overlap: unselected groups update while the all-gather is in flight
    hipStream_t side = nullptr;       // ... on this stream
    hipEvent_t ev_pack = nullptr, ev_gath = nullptr;
    double* red_dev = nullptr;        // demc_comm_allreduce staging
    size_t red_cap = 0;
    long long n_exchanges = 0;        // all-gathers issued (diagnostic, demc_comm_stats)
    struct demc_multi* multi = nullptr;
    bool multi_sealed = false;  // the set is built: demc_set_stream is refused from here on
    // which kernel instances the last update launched (demc_last_kernels: lets a test name the instance it compared)
    struct LastPlan {
        int k1 = -1;  // 0 k_propose per phase, 1 k_longrow, 2 k_propose resident, 3 k_propose streaming-resident, 4 k_res_mvn, 5 k_frozen_sweep, 6 k_res_obs
        int wg = 0, tile = 0, tail = 0, plain = 0, dt = 0, stream = 0, hist = 0, iso = 0, big = 0;
        int k2 = 0;   // 0 none (fused into K1), 1 k_cross_mfma, 2 k_obs_loglike, 3 k_hier_loglike, 4 user plug-in
        int ks = 0, k3 = 0;
    } last;
    // timing
    bool timing = false;
    std::vector<Timed> events;
    std::vector<hipEvent_t> event_pool;  // recycled events: a timed launch costs two hipEventRecord, no create/destroy
    double t_ms[5] = {0, 0, 0, 0, 0};
    long long t_n[5] = {0, 0, 0, 0, 0};
    // clock probe of the DIRECT likelihood kernel (demc_timing_clock): per workgroup {s_memtime, s_memrealtime, XCD} at its end
    unsigned long long* clk_dev = nullptr;
    size_t clk_cap = 0, clk_n = 0;  // workgroups the buffer holds / the last timed launch wrote
};

namespace {

int fail(demc_handle* h, int code, const std::string& msg) noexcept {
    if (h) {
        try {
            h->err = msg;
        } catch (...) {  // not even the message could be stored: the code alone goes back
        }
    }
    return code;
}

// No C++ exception crosses the C-ABI (include/demc.h): every extern "C" body runs inside this guard.
template <typename F>
int32_t guarded(demc_handle* h, F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return fail(h, DEMC_ENOMEM, "out of host memory");
    } catch (const std::exception& e) {
        return fail(h, DEMC_EHIP, std::string("internal error: ") + e.what());
    } catch (...) {
        return fail(h, DEMC_EHIP, "internal error (unknown exception)");
    }
}

// A/B switches for kernel experiments (tools/, profiles/README.md).  They exist only in a build made with
// -DDEMC_EXPERIMENTS (make EXPERIMENTS=1); the product library never reads the environment.
inline const char* experiment(const char* name) {
#ifdef DEMC_EXPERIMENTS
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return fail(h, DEMC_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));                    \
    } while (0)

// A handle belongs to one GPU; a host that drives several handles from one thread may have another device current.
#define USE_DEVICE(h) HIPCHK(hipSetDevice((h)->c.device_id))

template <typename T>
int dev_alloc(demc_handle* h, T** p, size_t n) {
    if (n == 0) n = 1;
    hipError_t e = hipMalloc((void**)p, n * sizeof(T));
    if (e != hipSuccess) return fail(h, DEMC_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    // hipMemset runs on the null stream and may return before the fill has happened; the handle's own stream is
    // non-blocking, i.e. NOT ordered behind the null stream -- without the wait a kernel launched right after an
    // allocation (demc_export_chains) could have its output zeroed underneath it.
    e = hipMemset(*p, 0, n * sizeof(T));
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) return fail(h, DEMC_EHIP, std::string("hipMemset: ") + hipGetErrorString(e));
    return DEMC_OK;
}
#define ALLOC(ptr, n)                          \
    do {                                       \
        int rc_ = dev_alloc(h, &(ptr), (n));   \
        if (rc_ != DEMC_OK) return rc_;        \
    } while (0)

// Device time per kernel class (demc_timing_enable): the events of a bracket travel IN the dispatch packets of its kernels
// (hipExtLaunchKernelGGL: start event on the first, stop event on every one -- the last record stands), so they read the
// kernels' own begin / end timestamps and put no barrier packets between the launches.  (Events recorded around each launch
// cost the cfg4 share 28 us per iteration of 160.)  `classic`: record the events around the bracket instead (module launch).
void tick(demc_handle* h, int cls, bool begin, bool classic = false) {
    if (!h->timing) return;
    if (begin) {
        Timed t;
        auto take = [&](hipEvent_t* e) {
            if (!h->event_pool.empty()) {
                *e = h->event_pool.back();
                h->event_pool.pop_back();
            } else
                hipEventCreate(e);
        };
        take(&t.a);
        take(&t.b);
        t.cls = cls;
        if (classic) {
            hipEventRecord(t.a, h->stream);
            t.started = true;
        }
        h->events.push_back(t);
        h->bracket_open = !classic;
    } else {
        if (classic) hipEventRecord(h->events.back().b, h->stream);
        h->bracket_open = false;
    }
}

#define LAUNCH_T(h, kern, grid, block, lds, ...)                                                                          \
    do {                                                                                                                  \
        if ((h)->timing && (h)->bracket_open) {                                                                           \
            Timed& t__ = (h)->events.back();                                                                              \
            hipExtLaunchKernelGGL(kern, grid, block, lds, (h)->stream, t__.started ? nullptr : t__.a, t__.b, 0, __VA_ARGS__); \
            t__.started = true;                                                                                           \
        } else                                                                                                            \
            hipLaunchKernelGGL(kern, grid, block, lds, (h)->stream, __VA_ARGS__);                                         \
    } while (0)

void drain_events(demc_handle* h) {
    for (auto& t : h->events) {
        if (t.started) {  // (a bracket may hold no launch at all)
            hipEventSynchronize(t.b);
            float ms = 0.f;
            hipEventElapsedTime(&ms, t.a, t.b);
            h->t_ms[t.cls] += ms;
            h->t_n[t.cls] += 1;
        }
        h->event_pool.push_back(t.a);
        h->event_pool.push_back(t.b);
    }
    h->events.clear();
}

bool is_mvn(int fam);

int pow2_ceil(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

KParams base_params(demc_handle* h) {
    KParams k;
    std::memset(&k, 0, sizeof k);
    const demc_config& c = h->c;
    k.n_groups = c.n_groups; k.Np = c.Np; k.D = c.D; k.group_offset = c.group_offset;
    k.a_lo = 0; k.n_act = c.Np; k.pool_lo = 0; k.pool_n = c.Np; k.exclude_self = 1;
    k.lpp = h->lpp; k.lpp3 = h->lpp > 256 ? 256 : h->lpp; k.mode = MODE_STEP;
    k.iter = 0; k.burnin = c.burnin; k.sweep = 0; k.seed = c.seed;
    k.beta = c.beta; k.eps = c.eps; k.sigma = c.sigma; k.kappa = c.kappa; k.theta_snooker = c.theta_snooker;
    k.proposal_kind = c.proposal_kind; k.partner_kind = c.partner_kind; k.update_kind = c.update_kind;
    k.fitness_kind = c.fitness_kind;
    k.theta = h->theta; k.weight = h->weight; k.id = h->id; k.prop = h->prop; k.prop_prior = h->prop_prior;
    k.prop_adj = h->prop_adj; k.prop_oob = h->prop_oob; k.tr_idx = h->tr_idx; k.tr_w = h->tr_w; k.tr_acc = h->tr_acc;
    k.dimtab = h->dimtab; k.dimseg = h->dimseg; k.n_seg = h->n_seg; k.ainv_lds = h->ainv_lds; k.mask = nullptr;
    std::memcpy(k.seg_start, h->seg_start, sizeof k.seg_start);
    k.seg_plain = h->seg_plain;
    std::memset(k.mrun_start, 0, sizeof k.mrun_start);
    k.n_mrun = 1; k.mrun_in = 1u;  // no block mask: one run, inside
    k.hist = h->hist; k.acc_hist = h->acc_hist; k.lp_hist = h->lp_hist; k.id_hist = h->id_hist; k.hist_ld = h->hist_ld;
    k.P = h->P; k.store_row = -1; k.tile_in_lds = h->tile_in_lds;
    k.family = h->family; k.N = h->N; k.d = h->d; k.n_acc = h->n_acc; k.n_partials = 1;
    k.partial = h->partial; k.aux = h->aux; k.data = h->data; k.data2 = h->data ? h->data + h->data2_off : nullptr;
    k.c0 = h->c0; k.c1 = h->c1; k.c2 = h->c2;
    k.direct = (is_mvn(h->family) && c.loglike_mode == DEMC_LOGLIKE_DIRECT) ? 1 : 0;
    k.n_split = 1; k.fuse_prep = 0; k.prep_mfma = 0; k.fuse_obs = 0; k.fuse_accept = 0; k.plan = 0;
    k.scr_doubles = (int)(h->k1_scr_bytes / sizeof(double)); k.write_prop = 1; k.trace = c.trace;
    k.Ainv = h->Ainv; k.sx = nullptr; k.xbar = h->xbar; k.Ypad = h->Ypad; k.dpad = h->dpad;
    k.rp_group = h->rp_group; k.rp_part = h->rp_part; k.rp_partner = h->rp_partner; k.rp_noise = h->rp_noise;
    k.rp_znoise = h->rp_znoise; k.rp_recomb = h->rp_recomb; k.rp_mig_groups = h->rp_mig_groups;
    k.rp_mig_particle = h->rp_mig_particle; k.rp_n_mig = h->rp_n_mig;
    k.glist = h->cur_glist;
    if (h->cur_glist) k.n_groups = h->cur_ng;
    return k;
}

template <int KS>
void launch_cross(demc_handle* h, const KParams& k, int grid, int k0, int n_chunks, int part0) {
    LAUNCH_T(h, (k_cross_mfma<KS, 4>), dim3(grid), dim3(256), 0, k, h->Ypad, h->dpad, k0, h->Xf, h->n_tiles, n_chunks, part0);
}

// Kernarg of the JIT-compiled user-likelihood kernel; the same text is prepended to the user's source.
struct UserKParams {
    int n_groups, Np, D, a_lo, n_act, n_chunks, nhyper, with_prior;
    long long N, P;
    const double* prop;
    double* partial;
    const double* data;
    const double* hyper;
    const long long* dims;
    const int* glist;
    int ndims, pad;
};
const char* kUserStruct = R"SRC(
#ifndef INFINITY
#define INFINITY __builtin_huge_val()
#endif
#ifndef NAN
#define NAN __builtin_nan("")
#endif
struct UserKParams {
    int n_groups, Np, D, a_lo, n_act, n_chunks, nhyper, with_prior;
    long long N, P;
    const double* prop;
    double* partial;
    const double* data;
    const double* hyper;
    const long long* dims;
    const int* glist;
    int ndims, pad;
};
)SRC";
const char* kUserPrologue = R"SRC(
__device__ double demc_user_obs(const double* theta, int D, const double* data, long long N, long long i,
                                const double* hyper, int nhyper);
)SRC";
// whole-row plug-in: the user's functions return the share of lane `lane` of `n_lanes` cooperating lanes
const char* kUserRowPrologue = R"SRC(
__device__ double demc_user_loglike_row(const double* theta, int D, const double* data, const long long* dims, int ndims,
                                        const double* hyper, int nhyper, int lane, int n_lanes);
#ifdef DEMC_USER_HAS_PRIOR
__device__ double demc_user_prior_row(const double* theta, int D, const double* hyper, int nhyper, int lane, int n_lanes);
#endif
)SRC";
// one workgroup of 256 lanes per proposal; the lanes' shares are summed in a fixed order (wave butterfly, then the four
// waves left to right), so the result does not depend on scheduling
const char* kUserRowKernel = R"SRC(
extern "C" __global__ __launch_bounds__(256) void k_user_row(UserKParams p) {
    __shared__ double s_part[4];
    const int q = blockIdx.x;
    const int g = q / p.n_act;
    const size_t slot = (size_t)(p.glist ? p.glist[g] : g) * p.Np + p.a_lo + (q - g * p.n_act);
    const double* th = p.prop + slot * p.D;
    double acc = demc_user_loglike_row(th, p.D, p.data, p.dims, p.ndims, p.hyper, p.nhyper, (int)threadIdx.x, 256);
#ifdef DEMC_USER_HAS_PRIOR
    if (p.with_prior) acc += demc_user_prior_row(th, p.D, p.hyper, p.nhyper, (int)threadIdx.x, 256);
#endif
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) p.partial[slot] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}
)SRC";
// thread per proposal x observation chunks: same mapping as k_obs_loglike
const char* kUserKernel = R"SRC(
extern "C" __global__ __launch_bounds__(256) void k_user_loglike(UserKParams p) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int chunk = blockIdx.y;
    if (q >= p.n_groups * p.n_act) return;
    const int g = q / p.n_act;
    const size_t slot = (size_t)(p.glist ? p.glist[g] : g) * p.Np + p.a_lo + (q - g * p.n_act);
    const double* th = p.prop + slot * p.D;
    const long long per = (p.N + p.n_chunks - 1) / p.n_chunks;
    const long long i0 = chunk * per, i1 = (i0 + per < p.N) ? i0 + per : p.N;
    double acc = 0.0;
    for (long long i = i0; i < i1; ++i) acc += demc_user_obs(th, p.D, p.data, p.N, i, p.hyper, p.nhyper);
    p.partial[(size_t)chunk * p.P + slot] = acc;
}
)SRC";

// Number of observation chunks for a K2 kernel that is one long uniform loop per workgroup: such a launch takes
// ceil(workgroups / resident workgroups) rounds of equal length, so among the admissible counts (1..cap) the one that fills
// the last round best wins; ties go to the larger count while the launch stays within four rounds.
int chunks_filling_rounds(long long blocks, long long cap, double resident_wgs) {
    long long best = 1;
    double best_fill = 0.0;
    for (long long nc = 1; nc <= cap; ++nc) {
        const double wgs = (double)blocks * (double)nc;
        const double fill = wgs / (std::ceil(wgs / resident_wgs) * resident_wgs);
        if (fill > best_fill + 1e-9 || (fill > best_fill - 1e-9 && wgs <= 4.0 * resident_wgs)) { best = nc; best_fill = fill; }
    }
    return (int)best;
}

// demc_timing_clock: with timing enabled the compute-bound likelihood kernels (k_direct_mvn, k_lba_wave) leave per workgroup
// {s_memtime, s_memrealtime, CU} as they end; the buffer is zeroed on the stream ahead of the launch (a workgroup whose first wave
// has no proposal reports nothing) and sized for the launch's grid.  Null when timing is off.
int clock_buffer(demc_handle* h, size_t wgs, unsigned long long** out) {
    *out = nullptr;
    if (!h->timing) return DEMC_OK;
    if (wgs > h->clk_cap) {
        if (h->clk_dev) { HIPCHK(hipStreamSynchronize(h->stream)); hipFree(h->clk_dev); h->clk_dev = nullptr; h->clk_cap = 0; }
        ALLOC(h->clk_dev, 3 * wgs);
        h->clk_cap = wgs;
    }
    HIPCHK(hipMemsetAsync(h->clk_dev, 0, 3 * wgs * sizeof(unsigned long long), h->stream));
    h->clk_n = wgs;
    *out = h->clk_dev;
    return DEMC_OK;
}

// K2 dispatch for the active set described by k.  Sets k.n_partials.
int launch_loglike(demc_handle* h, KParams& k) {
    const long long n_prop = (long long)k.n_groups * k.n_act;
    if (n_prop == 0) return DEMC_OK;
    switch (h->family) {
        case FAM_MVN_FULL:
        case FAM_MVN_ISO: {
            const bool suff = h->c.loglike_mode == DEMC_LOGLIKE_SUFFSTAT;  // then K1 already formed S
            k.n_partials = 1;
            if (k.direct) {
                // proposal blocks of 256 x observation chunks: enough workgroups for 8 waves per SIMD, chunks no shorter
                // than 64 observations, at most the partial-sum workspace
                const long long blocks = (n_prop + 255) / 256;
                long long cap = h->N / 64;
                if (cap > h->partial_cap) cap = h->partial_cap;
                if (cap < 1) cap = 1;
                // The kernel is one long uniform loop per workgroup: the launch takes ceil(workgroups / resident workgroups)
                // rounds of equal length, so the chunk count is chosen to fill the last round (cfg3: 128 blocks x 48 chunks =
                // 4 full rounds of 1536 resident workgroups; 32 chunks would leave a third of the chip idle in round 3).
                if (h->direct_wgs_per_cu == 0) {
                    int nb = 0;
                    const void* fn = h->dp_direct == 8 ? (const void*)k_direct_mvn<8> : h->dp_direct == 16 ? (const void*)k_direct_mvn<16>
                                     : h->dp_direct == 32 ? (const void*)k_direct_mvn<32> : (const void*)k_direct_mvn<64>;
                    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 256, 0));
                    h->direct_wgs_per_cu = nb > 0 ? nb : 1;
                }
                const int n_chunks = chunks_filling_rounds(blocks, cap, (double)h->direct_wgs_per_cu * h->n_cus);
                h->last.k2 = 5; h->last.ks = h->dp_direct;
                // timing on: every workgroup also leaves its shader-clock / reference-clock ticks (demc_timing_clock)
                unsigned long long* clk = nullptr;
                if (int rc = clock_buffer(h, (size_t)blocks * (size_t)n_chunks, &clk)) return rc;
                tick(h, 2, true);
                const dim3 grid((unsigned)blocks, (unsigned)n_chunks);
                switch (h->dp_direct) {
                    case 8: LAUNCH_T(h, k_direct_mvn<8>, grid, dim3(256), 0, k, n_chunks, clk); break;
                    case 16: LAUNCH_T(h, k_direct_mvn<16>, grid, dim3(256), 0, k, n_chunks, clk); break;
                    case 32: LAUNCH_T(h, k_direct_mvn<32>, grid, dim3(256), 0, k, n_chunks, clk); break;
                    default: LAUNCH_T(h, k_direct_mvn<64>, grid, dim3(256), 0, k, n_chunks, clk); break;
                }
                tick(h, 2, false);
                k.n_partials = n_chunks;
            } else if (!suff) {
                h->last.k2 = 1; h->last.ks = h->ks_t <= 1 ? 1 : h->ks_t <= 2 ? 2 : h->ks_t <= 4 ? 4 : h->ks_t <= 8 ? 8 : 16;
                tick(h, 2, true);
                // particle tiles of 256 (4 waves x MT=4 x 16) x observation chunks; chunks in multiples of 8 so that
                // blockIdx % 8 (the XCD a block lands on) selects the chunk it streams
                const long long n_ptiles = (n_prop + 255) / 256;
                int n_chunks = 8;
                while (n_ptiles * n_chunks < 1024 && n_chunks * 2 <= 32 && h->n_tiles / (n_chunks * 2) >= 16) n_chunks *= 2;
                if (h->n_tiles < n_chunks) n_chunks = (int)(h->n_tiles > 0 ? h->n_tiles : 1);
                const int n_kpass = h->n_kpass;
                if (n_chunks * n_kpass > h->partial_cap) n_chunks = h->partial_cap / n_kpass;
                if (n_chunks < 1) return fail(h, DEMC_EINVAL, "data dimension too large for the partial-sum workspace");
                const int grid2 = (int)(n_ptiles * n_chunks);
                for (int kp = 0; kp < n_kpass; ++kp) {
                    const int ks_here = h->ks_t;
                    const int k0 = kp * 4 * h->ks_t, part0 = kp * n_chunks;
                    if (ks_here <= 1) launch_cross<1>(h, k, grid2, k0, n_chunks, part0);
                    else if (ks_here <= 2) launch_cross<2>(h, k, grid2, k0, n_chunks, part0);
                    else if (ks_here <= 4) launch_cross<4>(h, k, grid2, k0, n_chunks, part0);
                    else if (ks_here <= 8) launch_cross<8>(h, k, grid2, k0, n_chunks, part0);
                    else launch_cross<16>(h, k, grid2, k0, n_chunks, part0);
                }
                tick(h, 2, false);
                k.n_partials = n_chunks * n_kpass;
            }
        } break;
        case FAM_GAUSSIAN:
        case FAM_BINOMIAL:
        case FAM_LBA:
        case FAM_LNR:
        case FAM_RASTRIGIN: {
            long long cap = h->N / 32;
            if (cap > h->partial_cap) cap = h->partial_cap;
            if (cap < 1 || h->family == FAM_RASTRIGIN) cap = 1;
            if (h->obs_wgs_per_cu == 0) {
                int nb = 0;
                HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_obs_loglike, 256, 0));
                h->obs_wgs_per_cu = nb > 0 ? nb : 1;
            }
            int n_chunks = chunks_filling_rounds((n_prop + 255) / 256, cap, (double)h->obs_wgs_per_cu * h->n_cus);
            if (const char* e = experiment("DEMC_OBS_CHUNKS")) {  // A/B experiments
                std::fprintf(stderr, "k_obs_loglike: %d workgroups per CU, %d chunks chosen\n", h->obs_wgs_per_cu, n_chunks);
                if (std::atoi(e) > 0 && std::atoi(e) <= cap) n_chunks = std::atoi(e);
            }
            // (A/B build only: k_lba_loglike, eight shifted copies of the Phi table with 128-byte rows -- measured 4 % SLOWER
            // than the packed single copy: 1.584 vs 1.522 ms per launch on cfg5, profiles/r04/ab_experiments.txt)
            bool lba_wide = false;
            if (const char* e = experiment("DEMC_LBA_WIDE")) lba_wide = h->family == FAM_LBA && h->lba_wide_ok && e[0] == '1';
#ifdef DEMC_EXPERIMENTS
            if (lba_wide) {
                // the conflict-free eight-copy table (k_lba_loglike): 140 KB of LDS, one workgroup of 512 per CU
                n_chunks = chunks_filling_rounds((n_prop + 511) / 512, cap, (double)h->n_cus);
                if (const char* e = experiment("DEMC_OBS_CHUNKS"))
                    if (std::atoi(e) > 0 && std::atoi(e) <= cap) n_chunks = std::atoi(e);
                h->last.k2 = 7;
                tick(h, 2, true);
                LAUNCH_T(h, k_lba_loglike<512>, dim3((unsigned)((n_prop + 511) / 512), (unsigned)n_chunks), dim3(512), kLbaTableBytes, k,
                         n_chunks);
                tick(h, 2, false);
                k.n_partials = n_chunks;
                break;
            }
#endif
            (void)lba_wide;
            if (h->family == FAM_LBA) {
                // a wave per proposal, lanes across the (sorted) trials: k_lba_wave.  Chunks of whole batches (512 trials), at least two a chunk.
                bool on = true;
                if (const char* e = experiment("DEMC_LBA_WAVE")) on = e[0] == '1';  // A/B experiments
                if (on) {
                    if (h->lba_wave_wgs_per_cu == 0) {
                        int nb = 0;
                        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_lba_wave<3>, 256, 0));
                        h->lba_wave_wgs_per_cu = nb > 0 ? nb : 1;
                    }
                    long long capw = h->N / 1024;
                    if (capw > h->partial_cap) capw = h->partial_cap;
                    if (capw < 1) capw = 1;
                    int nc = chunks_filling_rounds((n_prop + 3) / 4, capw, (double)h->lba_wave_wgs_per_cu * h->n_cus);
                    if (const char* e = experiment("DEMC_OBS_CHUNKS"))  // A/B experiments
                        if (std::atoi(e) > 0 && std::atoi(e) <= capw) nc = std::atoi(e);
                    h->last.k2 = 8;
                    unsigned long long* clk = nullptr;  // timing on: the clock the vector pipe held (demc_timing_clock)
                    if (int rc = clock_buffer(h, (size_t)((n_prop + 3) / 4) * (size_t)nc, &clk)) return rc;
                    tick(h, 2, true);
                    const dim3 grid((unsigned)((n_prop + 3) / 4), (unsigned)nc);
                    if (h->n_acc == 3) LAUNCH_T(h, k_lba_wave<3>, grid, dim3(256), 0, k, nc, clk);
                    else if (h->n_acc == 2) LAUNCH_T(h, k_lba_wave<2>, grid, dim3(256), 0, k, nc, clk);
                    else LAUNCH_T(h, k_lba_wave<0>, grid, dim3(256), 0, k, nc, clk);
                    tick(h, 2, false);
                    k.n_partials = nc;
                    break;
                }
            }
            h->last.k2 = 2;
            tick(h, 2, true);
            LAUNCH_T(h, k_obs_loglike, dim3((unsigned)((n_prop + 255) / 256), (unsigned)n_chunks), dim3(256), 0,
                               k, n_chunks);
            tick(h, 2, false);
            k.n_partials = n_chunks;
        } break;
        case FAM_USER: {
            long long want = (262144 + n_prop - 1) / n_prop;
            long long cap = h->N / 32;
            if (cap < 1) cap = 1;
            if (want > cap) want = cap;
            if (want > h->partial_cap) want = h->partial_cap;
            if (h->user_row) want = 1;
            UserKParams u;
            u.n_groups = k.n_groups; u.Np = k.Np; u.D = k.D; u.a_lo = k.a_lo; u.n_act = k.n_act; u.n_chunks = (int)want;
            u.nhyper = h->user_nhyper; u.with_prior = h->c.fitness_kind == DEMC_FITNESS_POSTERIOR ? 1 : 0;
            u.N = h->N; u.P = h->P; u.prop = k.prop; u.partial = k.partial;
            u.data = h->data; u.hyper = h->user_hyper; u.dims = h->user_dims; u.glist = k.glist; u.ndims = h->user_ndims; u.pad = 0;
            size_t sz = sizeof u;
            void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &u, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
            h->last.k2 = h->user_row ? 6 : 4;
            tick(h, 2, true, true);
            hipError_t e = h->user_row
                ? hipModuleLaunchKernel(h->user_kernel, (unsigned)n_prop, 1, 1, 256, 1, 1, 0, h->stream, nullptr, cfg)
                : hipModuleLaunchKernel(h->user_kernel, (unsigned)((n_prop + 255) / 256), (unsigned)want, 1, 256, 1, 1, 0,
                                        h->stream, nullptr, cfg);
            tick(h, 2, false, true);
            if (e != hipSuccess) return fail(h, DEMC_EHIP, std::string("hipModuleLaunchKernel: ") + hipGetErrorString(e));
            k.n_partials = (int)want;
        } break;
        case FAM_HIER_BINOMIAL:
        case FAM_HIER_GAUSSIAN: {
            h->last.k2 = 3;
            tick(h, 2, true);
            LAUNCH_T(h, k_hier_loglike, dim3((unsigned)n_prop), dim3(256), 0, k);
            tick(h, 2, false);
            k.n_partials = 1;
        } break;
        default:
            return fail(h, DEMC_EUNSUPPORTED, "model family not set or not registered");
    }
    return DEMC_OK;
}

bool is_mvn(int fam) { return fam == FAM_MVN_FULL || fam == FAM_MVN_ISO; }

// the K1 instance for (tile in LDS?, fused tail)
using K1Fn = void (*)(KParams);
// `lean`: 0 the general instance, 1 the default sampler only, 2 the default sampler + snooker (k_propose's LEAN)
K1Fn k1_instance(bool tile, int tail, int lean, int wg = 256) {
#define K1_ROW(WG_, TILE, RES_, LEAN_)                                                                              \
    {k_propose<WG_, TILE, TAIL_NONE, RES_, LEAN_>, k_propose<WG_, TILE, TAIL_PREP, RES_, LEAN_>,                   \
     k_propose<WG_, TILE, TAIL_PREP_MFMA, RES_, LEAN_>, k_propose<WG_, TILE, TAIL_OBS, RES_, LEAN_>}
    static const K1Fn tab[7][4] = {K1_ROW(256, false, false, 0), K1_ROW(256, true, false, 0), K1_ROW(256, true, false, 1),
                                   K1_ROW(512, false, false, 0), K1_ROW(256, true, false, 2), K1_ROW(256, false, false, 1),
                                   K1_ROW(256, false, false, 2)};
#undef K1_ROW
    if (wg == 512) return tab[3][tail];  // a 512-thread workgroup per particle (very long rows, no tile)
    // without a tile: the general instance, or the default sampler (+ snooker / block updates) with partners from the history
    // (DE-MC_Z: lean_hist)
    return tab[tile ? (lean == 1 ? 2 : lean == 2 ? 4 : 1) : (lean == 1 ? 5 : lean == 2 ? 6 : 0)][tail];
}

// which tail K1 carries for this model, mode and schedule
void set_tail_flags(demc_handle* h, KParams& k) {
    const demc_config& c = h->c;
    const bool suff = c.loglike_mode == DEMC_LOGLIKE_SUFFSTAT;
    // K1 tails: MvNormal preparation always; the whole update when the likelihood is O(D^2) given data-only
    // statistics and a phase writes only rows that no other workgroup reads (two_colour, or the identity pass)
    k.fuse_prep = is_mvn(h->family) ? 1 : 0;
    k.prep_mfma = (h->family == FAM_MVN_FULL && k.lpp >= 4 && k.lpp <= 16 && h->d <= 32) ? 1 : 0;
    if (const char* e = experiment("DEMC_PREP_MFMA")) k.prep_mfma = k.prep_mfma && e[0] == '1';  // A/B experiments
    k.sx = (k.fuse_prep && suff) ? h->sx : nullptr;
    k.Ainv = (h->family == FAM_MVN_FULL) ? h->Ainv : nullptr;
    // small-N scalar-data families: the sub-group of a particle sums the per-observation terms itself
    // (hierarchical families: one term per subject, i.e. O(D) like the proposal itself; their Gaussian form has p.d
    // observations behind every subject)
    const long long obs_work = (h->family == FAM_HIER_GAUSSIAN) ? h->N * (h->d > 0 ? h->d : 1) : h->N;
    const bool cheap_obs = ((h->family == FAM_GAUSSIAN || h->family == FAM_BINOMIAL || h->family == FAM_RASTRIGIN) &&
                            h->N / k.lpp <= 512) ||
                           (h->hier_scr && obs_work / k.lpp <= 4096);
    // nothing a moving particle reads can move in the same launch: two_colour (partners rest), the identity pass, and the
    // sequential schedule (one particle per group and launch, handled by that group's only workgroup)
    // ... and the synchronous schedule when every partner row comes from the HISTORY (resample: rows of earlier iterations,
    // which this launch does not write) and no base particle is read from the current population (random_gamma reads one
    // during burn-in only, crossover.jl:156-164): the reference's own parallel-safe scheme runs as ONE kernel
    const bool hist_private = c.schedule == DEMC_SCHED_SYNCHRONOUS && c.partner_kind == DEMC_PARTNER_HISTORY && k.mode == MODE_STEP &&
                              !(c.proposal_kind == 0 && k.iter <= c.burnin);
    const bool phase_private = c.schedule == DEMC_SCHED_TWO_COLOUR || c.schedule == DEMC_SCHED_SEQUENTIAL || k.mode == MODE_IDENT || hist_private;
    h->tf_cheap_obs = cheap_obs;
    k.fuse_obs = (cheap_obs && phase_private && c.fuse != 1) ? 1 : 0;
    k.fuse_accept = (((k.fuse_prep && suff) || k.fuse_obs) && c.fuse != 1 && phase_private) ? 1 : 0;
    k.write_prop = (!k.fuse_accept || k.trace) ? 1 : 0;
}
// the default sampler and nothing else: K1 has an instance with every other branch compiled out
int lean_level(const demc_handle* h, const KParams& k) {
    const demc_config& c = h->c;
    const bool base = k.mode == MODE_STEP && c.proposal_kind == 0 && c.partner_kind == DEMC_PARTNER_CURRENT && c.update_kind == 0 &&
                      c.fitness_kind == 0 && c.kappa == 1.0 && !k.trace && !h->rp_active;
    return !base ? 0 : (c.theta_snooker == 0.0 && c.n_blocks == 0) ? 1 : 2;  // + snooker / block updates: their own lean instance
}
bool is_plain(const demc_handle* h, const KParams& k) { return lean_level(h, k) == 1; }
// DE-MC_Z (sample = resample, crossover.jl:113-124) with the rest of the sampler at its defaults: partners are cells of the
// history, so there is no tile, and the plain instance of the no-tile form serves it
// (1: the default sampler; 2: + snooker updates / block updates -- the reference's own DE-MC_Z runs use theta_snooker = 0.1,
// test/multivariate_normal_tests.jl:50-59, Examples/Hierarchical_Example.jl:103-114; 0: the general instance)
int lean_hist(const demc_handle* h, const KParams& k) {
    const demc_config& c = h->c;
    const bool base = k.mode == MODE_STEP && c.proposal_kind == 0 && c.partner_kind == DEMC_PARTNER_HISTORY && c.update_kind == 0 &&
                      c.fitness_kind == 0 && c.kappa == 1.0 && !k.trace && !h->rp_active;
    return !base ? 0 : (c.theta_snooker == 0.0 && c.n_blocks == 0) ? 1 : 2;
}
int tail_of(const KParams& k) { return k.prep_mfma ? TAIL_PREP_MFMA : k.fuse_prep ? TAIL_PREP : k.fuse_obs ? TAIL_OBS : TAIL_NONE; }

// (`k` by value: the frozen-sweep / snapshot overrides below -- glist, base_*, snap_*, the fuse flags -- stay inside this launch;
// run_sweep reuses its own copy for the second colour phase and for every particle of the sequential schedule)
int launch_phase(demc_handle* h, KParams k) {
    const long long n_prop = (long long)k.n_groups * k.n_act;
    if (n_prop == 0) return DEMC_OK;
    const demc_config& c = h->c;
    set_tail_flags(h, k);
    const int wg = k.lpp > 256 ? 512 : 256;  // K1 workgroup: 512 threads when one particle takes 512 lanes
    const int ppp = wg / k.lpp, ppp3 = 256 / k.lpp3;
    const int max_split = (k.n_act + ppp - 1) / ppp;
    int target_wgs = 512;
    if (const char* e = experiment("DEMC_K1_WGS")) target_wgs = std::atoi(e);  // A/B experiments
    int n_split = (target_wgs + k.n_groups - 1) / k.n_groups;
    if (n_split > max_split) n_split = max_split;
    if (n_split < 1) n_split = 1;
    k.n_split = n_split;
    const bool tile = k.tile_in_lds && c.partner_kind == DEMC_PARTNER_CURRENT && wg == 256;
    // plan stage (per-particle scalars once per workgroup): 4 doubles + 4 ints per particle of the workgroup's slice
    const size_t per_split = (size_t)(k.n_act + n_split - 1) / n_split;
    const size_t plan_bytes = per_split * (4 * sizeof(double) + 4 * sizeof(int));
    // tile = the partner pool, plus the workgroup's own slice of moving rows when those lie outside the pool
    k.own_in_pool = (k.a_lo >= k.pool_lo && k.a_lo + k.n_act <= k.pool_lo + k.pool_n) ? 1 : 0;
    k.tile_rows = k.pool_n + (k.own_in_pool ? 0 : (int)per_split);
    const size_t lds_tile = h->k1_lds - h->k1_tile_bytes + (size_t)k.tile_rows * c.D * sizeof(double);  // <= k1_lds
    k.plan = (tile && k.mode == MODE_STEP && k.lpp >= 4 && k.lpp <= 64 && lds_tile + plan_bytes <= kMaxDynLds && !h->rp_active) ? 1 : 0;
    if (const char* e = experiment("DEMC_K1_PLAN")) k.plan = k.plan && e[0] == '1';  // A/B experiments
    // long rows of a hierarchical family, whole update fused: the dedicated one-pass kernel (demc_longrow.hpp)
    const size_t cdf_doubles = k.pool_n > 256 ? (size_t)k.pool_n + ((size_t)k.pool_n + 15) / 16 : 0;
    const size_t lr_lds = ((((size_t)c.D + 1) & ~(size_t)1) + cdf_doubles) * sizeof(double);
    const bool lr_shape = k.lpp > 64 && h->hier_scr && k.mode == MODE_STEP && !h->rp_active && c.fuse != 2 && h->n_seg > 0 && lr_lds <= kMaxDynLds;
    // DE-MC_Z INSIDE burn-in (Examples/Hierarchical_Example.jl:103-114 spends half its run there): random_gamma's base particle
    // (crossover.jl:156-164) is a row of the current population, which another workgroup of this synchronous launch may be writing,
    // so set_tail_flags left the sweep unfused (K1 -> k_hier_loglike -> K3 through the proposal buffer: 2.3x the time).  The rows
    // and weights of the sweep's START are what the synchronous schedule reads anyway: two device-to-device copies (into the
    // proposal buffers, which a fused sweep does not use) make them a snapshot nobody writes, and the sweep is ONE k_longrow
    // launch again -- base rows and select_base's weights from the snapshot (KParams::base_theta / base_weight).
    // The same holds for every family whose update fuses into K1 past burn-in (the general kernel's no-tile form reads its base
    // rows through the same two pointers): with the snapshot, DE-MC_Z is one launch per sweep inside burn-in as well as past it.
    const bool suff_mvn = k.fuse_prep && c.loglike_mode == DEMC_LOGLIKE_SUFFSTAT;
    bool snapshot = false, have_snap2 = false;
    if (!k.fuse_accept && k.mode == MODE_STEP && c.schedule == DEMC_SCHED_SYNCHRONOUS && c.partner_kind == DEMC_PARTNER_HISTORY &&
        c.proposal_kind == 0 && k.iter <= c.burnin && (h->tf_cheap_obs || suff_mvn) && c.fuse == 0 && !k.trace && !h->rp_active &&
        k.theta == h->theta) {
        bool on = true;
        if (const char* e = experiment("DEMC_LR_SNAPSHOT")) on = e[0] == '1';  // A/B experiments
        if (on) {  // (the copies follow the choice of the kernel: a frozen sweep reads its base row at the block's scalars only)
            snapshot = true;
            k.base_theta = h->prop; k.base_weight = h->prop_prior;
            k.fuse_obs = h->tf_cheap_obs ? 1 : 0; k.fuse_accept = 1; k.write_prop = 0;
            // ... and when the sweep before this one was a frozen sweep over the whole population, it left this snapshot behind
            if (h->snap2 && h->snap2_iter == (int64_t)k.iter && h->snap2_sweep == (int)k.sweep && !h->cur_glist) {
                h->snap2_iter = -1;  // (used once: by the sweep it was written for, in the demc_step call that wrote it)
                snapshot = false;
                have_snap2 = true;
                k.base_theta = h->snap2; k.base_weight = h->snap2_w;
            }
        }
    }
    auto take_snapshot = [&](bool block_only) -> int {
        HIPCHK(hipMemcpyAsync(h->prop_prior, h->weight, sizeof(double) * (size_t)h->P, hipMemcpyDeviceToDevice, h->stream));
        if (!block_only) {
            HIPCHK(hipMemcpyAsync(h->prop, h->theta, sizeof(double) * (size_t)h->P * c.D, hipMemcpyDeviceToDevice, h->stream));
            return DEMC_OK;
        }
        for (int r = 0; r < k.n_mrun; ++r)  // the columns of the block: a strided copy per run of the mask (a few doubles per row)
            if ((k.mrun_in >> r) & 1u) {
                const int lo = k.mrun_start[r], hi = r + 1 < k.n_mrun ? k.mrun_start[r + 1] : c.D;
                HIPCHK(hipMemcpy2DAsync(h->prop + lo, sizeof(double) * (size_t)c.D, h->theta + lo, sizeof(double) * (size_t)c.D,
                                        sizeof(double) * (size_t)(hi - lo), (size_t)h->P, hipMemcpyDeviceToDevice, h->stream));
            }
        return DEMC_OK;
    };
    // A block sweep that FREEZES the row (the block holds a few hyper-parameters): the sweep reduced to what it is -- one pass
    // over the particle's own row, partner rows read at the block's scalars only (whole only by the one particle in ten whose
    // snooker coin fired: its projections run over the row), no LDS row, four workgroups per CU (<256>: 120 registers; three for
    // <256,big>: 155; demc_frozen.hpp) -- instead of
    // the subject-sweep machinery of k_longrow at eight waves per CU.  Partners from the population or from the history, the
    // base row from the sweep-start snapshot when there is one.
    if (lr_shape && k.fuse_obs && k.fuse_accept && k.mask && k.n_mrun > 0 && !k.trace &&
        (h->family == FAM_HIER_BINOMIAL || h->family == FAM_HIER_GAUSSIAN) && c.kappa == 1.0 && k.pool_n <= 256) {
        int inside = 0;
        for (int r = 0; r < k.n_mrun; ++r)
            if ((k.mrun_in >> r) & 1u) inside += (r + 1 < k.n_mrun ? k.mrun_start[r + 1] : c.D) - k.mrun_start[r];
        // (enough moving particles for several workgroups per CU, counted on the geometry's groups so that a shard takes the form
        // of the whole run: with one particle per CU the launch is that particle's dependent prologue whatever follows it --
        // measured on cfg4's share: no form of this kernel beats k_longrow<512> there, profiles/r05/NOTES.md)
        int longest = 0;
        for (int r = 0; r < k.n_mrun; ++r)
            if ((k.mrun_in >> r) & 1u) longest = std::max(longest, (r + 1 < k.n_mrun ? k.mrun_start[r + 1] : c.D) - k.mrun_start[r]);
        // (a long run inside the block -- the subject block -- takes the BIG instance: proposals formed on the fly, four scalars
        // per thread and round behind 16-byte loads, which the hierarchical Binomial family's even rows allow)
        bool big = longest > 256;
        long long min_particles = 2LL * h->n_cus;
        if (const char* e = experiment("DEMC_FROZEN_MIN")) min_particles = std::atoll(e);  // A/B experiments
        bool on = inside >= 1 && (long long)h->geo_groups * k.n_act >= min_particles;
        if (big) {
            on = on && h->family == FAM_HIER_BINOMIAL && (c.D & 1) == 0;
            if (const char* e = experiment("DEMC_FROZEN_BIG")) on = on && e[0] == '1';  // A/B experiments
        } else
            on = on && inside <= kFrozenMax;
        if (const char* e = experiment("DEMC_FROZEN")) on = on && e[0] == '1';  // A/B experiments
        if (const char* e = experiment("DEMC_FROZEN_EXT"))  // A/B experiments: only what round 5's first form of the kernel served
            if (e[0] == '0') on = on && !k.base_theta && c.theta_snooker == 0.0 && c.partner_kind == DEMC_PARTNER_CURRENT;
        if (on) {
            if (snapshot) {
                bool cols = true;
                if (const char* e = experiment("DEMC_FROZEN_SNAP_COLS")) cols = e[0] == '1';  // A/B experiments
                if (int rc = take_snapshot(cols)) return rc;
            }
            int wg_f = 256;
            if (const char* e = experiment("DEMC_FROZEN_WG")) wg_f = std::atoi(e);  // A/B experiments
            if (!k.glist && h->frozen_order_d && k.iter >= h->frozen_iter0 && k.iter < h->frozen_iter0 + h->frozen_iters &&
                (int)k.sweep < c.n_blocks)
                k.glist = h->frozen_order_d + ((size_t)(k.iter - h->frozen_iter0) * c.n_blocks + k.sweep) * (size_t)c.n_groups;
            // the next sweep of this iteration will want a snapshot of its start as well: this sweep writes it on its way (every
            // row passes through the kernel once) -- when it moves the whole population and does not itself read the buffer
            const int n_sweeps = c.n_blocks > 0 ? c.n_blocks : 1;
            if (!big && snapshot && !have_snap2 && (int)k.sweep + 1 < n_sweeps && !h->cur_glist && k.a_lo == 0 && k.n_act == c.Np &&
                k.n_groups == c.n_groups) {
                bool on2 = true;
                if (const char* e = experiment("DEMC_FROZEN_SNAP2")) on2 = e[0] == '1';  // A/B experiments
                if (on2 && h->snap2) {  // (allocated at plan time -- plan_snap2 -- never inside an enqueued step)
                    k.snap_theta = h->snap2; k.snap_weight = h->snap2_w;
                    h->snap2_iter = (int64_t)k.iter; h->snap2_sweep = (int)k.sweep + 1;
                }
            }
            h->last = demc_handle::LastPlan();
            h->last.k1 = 5; h->last.wg = wg_f; h->last.big = big;
            tick(h, 0, true);
            if (big) {
#ifdef DEMC_EXPERIMENTS
                if (wg_f == 512)
                    LAUNCH_T(h, (k_frozen_sweep<512, 2, 2, true>), dim3((unsigned)n_prop), dim3(512), 0, k);
                else if (wg_f == 1024)
                    LAUNCH_T(h, (k_frozen_sweep<1024, 4, 2, true>), dim3((unsigned)n_prop), dim3(1024), 0, k);
                else if (wg_f == 2562)
                    LAUNCH_T(h, (k_frozen_sweep<256, 2, 2, true>), dim3((unsigned)n_prop), dim3(256), 0, k);
                else
#endif
                    LAUNCH_T(h, (k_frozen_sweep<256, 3, 2, true>), dim3((unsigned)n_prop), dim3(256), 0, k);
                tick(h, 0, false);
                return DEMC_OK;
            }
#ifdef DEMC_EXPERIMENTS  // (A/B builds: other workgroup sizes, register budgets and pairs per round -- profiles/r05/NOTES.md section 8)
            if (wg_f == 64)
                LAUNCH_T(h, k_frozen_sweep<64>, dim3((unsigned)n_prop), dim3(64), 0, k);
            else if (wg_f == 128)
                LAUNCH_T(h, k_frozen_sweep<128>, dim3((unsigned)n_prop), dim3(128), 0, k);
            else if (wg_f == 512)
                LAUNCH_T(h, k_frozen_sweep<512>, dim3((unsigned)n_prop), dim3(512), 0, k);
            else if (wg_f == 768)
                LAUNCH_T(h, k_frozen_sweep<768>, dim3((unsigned)n_prop), dim3(768), 0, k);
            else if (wg_f == 1024)
                LAUNCH_T(h, (k_frozen_sweep<1024, 4>), dim3((unsigned)n_prop), dim3(1024), 0, k);
            else if (wg_f == 2564)  // 256 threads, four waves per SIMD, two pairs per round
                LAUNCH_T(h, (k_frozen_sweep<256, 4, 2>), dim3((unsigned)n_prop), dim3(256), 0, k);
            else if (wg_f == 2541)  // ... one pair per round
                LAUNCH_T(h, (k_frozen_sweep<256, 4, 1>), dim3((unsigned)n_prop), dim3(256), 0, k);
            else if (wg_f == 2531)  // three waves per SIMD, one pair per round
                LAUNCH_T(h, (k_frozen_sweep<256, 3, 1>), dim3((unsigned)n_prop), dim3(256), 0, k);
            else if (wg_f == 2551)  // five waves per SIMD (96 registers), one pair per round
                LAUNCH_T(h, (k_frozen_sweep<256, 5, 1>), dim3((unsigned)n_prop), dim3(256), 0, k);
            else
#endif
                LAUNCH_T(h, k_frozen_sweep<256>, dim3((unsigned)n_prop), dim3(256), 0, k);
            tick(h, 0, false);
            return DEMC_OK;
        }
    }
    if (snapshot)
        if (int rc = take_snapshot(false)) return rc;
    if (lr_shape && k.fuse_obs && k.fuse_accept) {
        {
            if (const char* e = experiment("DEMC_LR_EXIT")) k.n_split = -std::atoi(e);  // A/B experiments
            if (const char* e = experiment("DEMC_LR_DEFER"))
                if (e[0] == '0' && k.n_split >= 0) k.n_split = -100;
            // Enough moving particles for two workgroups per CU (counted on the geometry's groups, so that a shard takes the
            // same form as the whole run): 256 threads each -- one wave per SIMD per workgroup, two particles per CU out of
            // step, one's prologue and row moves under the other's pass -- when two rows fit in the CU's LDS.
            if (h->lr_two_lds != lr_lds) {  // (asked once per row size)
                int nb = 0;
                HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_longrow<256>, 256, lr_lds));
                h->lr_two_lds = lr_lds;
                h->lr_two_fit = nb >= 2;
            }
            const bool two_per_cu = (long long)h->geo_groups * k.n_act >= 2LL * h->n_cus && h->lr_two_fit;
            int wg_lr = two_per_cu ? 256 : 512;
            if (const char* e = experiment("DEMC_LR_WG")) wg_lr = std::atoi(e);  // A/B experiments
#ifdef DEMC_EXPERIMENTS
            {
                static bool said = false;
                if (!said) {
                    said = true;
                    int n256 = 0, n384 = 0, n512 = 0;
                    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n256, (const void*)k_longrow<256>, 256, lr_lds);
                    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n384, (const void*)k_longrow<384, 2>, 384, lr_lds);
                    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n512, (const void*)k_longrow<512, 2>, 512, lr_lds);
                    std::fprintf(stderr, "experiment: k_longrow workgroups per CU at %zu B of dynamic LDS: <256> %d, <384,2> %d, <512,2> %d\n", lr_lds, n256, n384, n512);
                }
            }
#endif
            h->last = demc_handle::LastPlan();
            h->last.k1 = 1; h->last.wg = wg_lr;
            tick(h, 0, true);
            // persistent: as many workgroups as are resident at once, each takes particles blockIdx.x, + gridDim.x, ... (the
            // row moves of one particle then run inside the span loops of the next: demc_longrow.hpp).  A multiple of 8 keeps
            // a workgroup's particles on its own XCD (the kernel's blockIdx -> group mapping).
            long long grid_lr = (long long)(wg_lr == 512 ? 1 : 2) * h->n_cus;
            if (const char* e = experiment("DEMC_LR_GRID")) grid_lr = std::atoll(e);  // A/B experiments
            if (grid_lr > n_prop || grid_lr < 1) grid_lr = n_prop;
            if ((k.n_groups & 7) == 0 && grid_lr >= 8) grid_lr &= ~7LL;
            if (wg_lr == 256)
                LAUNCH_T(h, k_longrow<256>, dim3((unsigned)grid_lr), dim3(256), lr_lds, k);
#ifdef DEMC_EXPERIMENTS
            else if (wg_lr == 384)
                LAUNCH_T(h, (k_longrow<384, 2>), dim3((unsigned)grid_lr), dim3(384), lr_lds, k);
            else if (wg_lr == 1024)  // (512 threads, two workgroups per CU)
                LAUNCH_T(h, (k_longrow<512, 2>), dim3((unsigned)grid_lr), dim3(512), lr_lds, k);
#endif
            else
                LAUNCH_T(h, k_longrow<512>, dim3((unsigned)grid_lr), dim3(512), lr_lds, k);
            tick(h, 0, false);
            return DEMC_OK;
        }
    }
    tick(h, 0, true);
    const int tail = tail_of(k);
    const size_t lds = tile ? lds_tile + (k.plan ? plan_bytes : 0) : h->k1_lds - h->k1_tile_bytes;
    const int lean = (tile && wg == 256) ? lean_level(h, k) : (!tile && wg == 256) ? lean_hist(h, k) : 0;
    h->last = demc_handle::LastPlan();
    h->last.k1 = 0; h->last.wg = wg; h->last.tile = tile; h->last.tail = tail; h->last.plain = lean;
    LAUNCH_T(h, k1_instance(tile, tail, lean, wg), dim3(k.n_groups * n_split), dim3(wg), lds, k);
    tick(h, 0, false);
    if (k.fuse_accept) return DEMC_OK;
    int rc = launch_loglike(h, k);
    if (rc != DEMC_OK) return rc;
    h->last.k3 = 1;
    tick(h, 3, true);
    LAUNCH_T(h, k_accept_store, dim3((unsigned)((n_prop + ppp3 - 1) / ppp3)), dim3(256), 0, k);
    tick(h, 3, false);
    return DEMC_OK;
}

// ---- resident form of K1: one workgroup per group, both colour phases of several iterations in one launch ----
K1Fn k1_resident_instance(int wg, int tail, int lean) {
#define K1_ROW(WG_, LEAN_)                                                                                      \
    {k_propose<WG_, true, TAIL_NONE, true, LEAN_>, k_propose<WG_, true, TAIL_PREP, true, LEAN_>,               \
     k_propose<WG_, true, TAIL_PREP_MFMA, true, LEAN_>, k_propose<WG_, true, TAIL_OBS, true, LEAN_>}
    static const K1Fn tab[6][4] = {K1_ROW(256, 0), K1_ROW(512, 0), K1_ROW(256, 1), K1_ROW(512, 1), K1_ROW(256, 2), K1_ROW(512, 2)};
#undef K1_ROW
    return tab[(wg == 512 ? 1 : 0) + 2 * lean][tail];
}

// Decides once per model whether the resident form applies and with which geometry (lanes per particle, workgroup size,
// LDS bytes).  It needs the fused accept tail (the whole update inside K1), the two_colour schedule (a phase writes only
// rows nobody reads), partners from the current population, and the whole group + its scratch in LDS.
void plan_resident(demc_handle* h) {
    const demc_config& c = h->c;
    h->res_ok = false;
    if (c.fuse != 0 || c.schedule != DEMC_SCHED_TWO_COLOUR || c.partner_kind != DEMC_PARTNER_CURRENT || c.Np < 4) return;
    if (const char* e = experiment("DEMC_RESIDENT"))  // A/B experiments
        if (e[0] == '0') return;
    int lpp_max = pow2_ceil((c.D + 1) / 2);
    if (lpp_max > 64) return;
    const int n_act = c.Np - c.Np / 2;
    const size_t D = (size_t)c.D, Np = (size_t)c.Np, d = (size_t)h->d;
    const bool mvn = is_mvn(h->family);
    // geometry for a thread budget: lanes per particle so that the moving half fills one pass when possible (never below
    // 4 lanes), workgroup size, LDS bytes; false when the update cannot be fused or the group does not fit
    int lpp = 0, wg = 0;
    size_t bytes = 0, scr_doubles = 0;
    auto geometry = [&](int budget) -> bool {
        lpp = 4;
        while (lpp * 2 <= lpp_max && n_act * lpp * 2 <= budget) lpp *= 2;
        if (lpp > lpp_max) lpp = lpp_max;
        wg = (n_act * lpp > 256 && budget > 256) ? 512 : 256;
        KParams k = base_params(h);
        k.lpp = lpp;
        k.mode = MODE_STEP;
        set_tail_flags(h, k);
        if (!k.fuse_accept) return false;
        scr_doubles = (k.fuse_prep || k.fuse_obs) ? (size_t)(wg / lpp) * (D + 2) : 0;
        const size_t doubles =
            Np * D + Np + (Np + (Np + 15) / 16) + ((h->family == FAM_MVN_FULL && h->ainv_lds) ? d * d : 0) + (mvn ? d : 0) + scr_doubles;
        bytes = doubles * sizeof(double) + (size_t)n_act * (4 * sizeof(double) + 4 * sizeof(int));
        return bytes <= kMaxDynLds;
    };
    // More groups than CUs: two 256-thread workgroups per CU keep twice as many groups in flight as one of 512 -- when
    // two of them fit in a CU's LDS (measured at 1024 groups x 64: 0.082 -> see DESIGN.md section 6).
    const bool two_per_cu = h->geo_groups > 256 && geometry(256) && bytes <= 75 * 1024;
    if (!two_per_cu && !geometry(512)) return;
    h->res_ok = true; h->res_lpp = lpp; h->res_wg = wg; h->res_lds = bytes; h->res_scr_doubles = (int)scr_doubles;
}

int launch_resident(demc_handle* h, long long iter0, int n_iters) {
    const demc_config& c = h->c;
    KParams k = base_params(h);
    k.lpp = h->res_lpp;
    set_tail_flags(h, k);
    k.iter = iter0; k.n_iters = n_iters; k.n_sweeps = c.n_blocks > 0 ? c.n_blocks : 1;
    k.mask = c.n_blocks > 0 ? h->masks : nullptr;
    k.n_rows = h->hist ? c.n_rows : 0;
    k.n_split = 1; k.exclude_self = 0; k.own_in_pool = 1; k.tile_rows = c.Np; k.tile_in_lds = 1;
    k.scr_doubles = h->res_scr_doubles;
    k.plan = (k.lpp >= 4) ? 1 : 0;
    if (const char* e = experiment("DEMC_K1_PLAN")) k.plan = k.plan && e[0] == '1';  // A/B experiments
    h->last = demc_handle::LastPlan();
    h->last.k1 = 2; h->last.wg = h->res_wg; h->last.tile = 1; h->last.tail = tail_of(k); h->last.plain = lean_level(h, k);
    tick(h, 0, true);
    LAUNCH_T(h, k1_resident_instance(h->res_wg, tail_of(k), lean_level(h, k)), dim3(k.n_groups), dim3(h->res_wg), h->res_lds,
                       k);
    tick(h, 0, false);
    return DEMC_OK;
}

// ---- lean resident kernel (demc_resmvn.hpp): default sampler, MvNormal full Sigma, D = d <= 32, one pass per phase ----
void plan_lean(demc_handle* h) {
    const demc_config& c = h->c;
    h->lean_ok = h->lean_stream_ok = h->lean_hist_ok = h->lean_direct_ok = false;
    // The per-observation families under the default sampler (demc_resobs.hpp): Gaussian, Binomial and the LNR with a handful of
    // parameters (a lane per scalar of a sixteen-lane particle) and few enough observations for sixteen lanes to walk them; the
    // group, its scratch rows and -- LNR -- the log Phi(-z) table in LDS.  Whether the SAMPLER is the default one is asked per step
    // (is_plain: a replay or a trace takes the general kernel).
    h->lean_obs_ok = false;
    if ((h->family == FAM_GAUSSIAN || h->family == FAM_BINOMIAL || h->family == FAM_LNR) && c.D <= 16 && c.fuse == 0 &&
        c.schedule == DEMC_SCHED_TWO_COLOUR && c.partner_kind == DEMC_PARTNER_CURRENT && c.Np >= 4 && c.Np <= 512 && h->n_seg >= 1 &&
        h->N / 16 <= 512) {
        bool ref = false;
        for (const DimTab& t : h->h_tab) ref = ref || t.kind == PR_NORMAL_REF;
        const size_t doubles = (size_t)c.Np * c.D + (size_t)c.Np + (size_t)(c.Np - c.Np / 2) + (size_t)(256 / 16) * c.D +
                               (h->family == FAM_LNR ? (size_t)kLogPhiRows * kLogPhiRow : 0);
        bool on = !ref && doubles * sizeof(double) <= kMaxDynLds;
        if (const char* e = experiment("DEMC_LEAN_OBS")) on = on && e[0] == '1';  // A/B experiments
        if (on) { h->lean_obs_ok = true; h->lean_obs_lds = doubles * sizeof(double); }
    }
    // DE-MC_Z (history partners, the synchronous schedule) on the same family in SUFFSTAT mode: the lean body with partner rows
    // from the history, one launch per iteration (step_body) -- all it needs in LDS are select_base's cumulative weights and the
    // centred rows
    // (... and on MvNormal(mu, sigma^2 I) with sigma a parameter, D = d + 1: the ISO instances of the same body)
    const bool lean_fam = (h->family == FAM_MVN_FULL && c.D == h->d) || (h->family == FAM_MVN_ISO && c.D == h->d + 1);
    if (lean_fam && c.D <= 32 && c.fuse == 0 && c.schedule == DEMC_SCHED_SYNCHRONOUS &&
        c.partner_kind == DEMC_PARTNER_HISTORY && c.loglike_mode == DEMC_LOGLIKE_SUFFSTAT && c.Np >= 4 && h->n_seg >= 1 &&
        (c.Np - c.Np / 2) * 4 <= 512 && h->hist) {
        bool ref = false;
        for (const DimTab& t : h->h_tab) ref = ref || t.kind == PR_NORMAL_REF;
        bool on = !ref;
        if (const char* e = experiment("DEMC_LEAN_HIST")) on = on && e[0] == '1';  // A/B experiments
        if (on) {
            const int wgh = (c.Np - c.Np / 2) * 4 > 256 ? 512 : 256;
            h->lean_hist_ok = true; h->lean_wg = wgh;
            // cdf | chunk offsets | centred rows | A^-1 fragments [2][8][64]
            // ... | the first half's held-back rows [wg][8] (+ alignment slack; the snooker instance parks them) | ISO: xbar and
            // sum_i x~_i [2][32]: demc_resmvn.hpp, pend_l / iso_l
            h->lean_hist_lds = ((size_t)c.Np + 16 + (size_t)(wgh / 4) * ((size_t)c.D + 2) + 16 * 64 + 2 + (size_t)wgh * 8 + 64) * sizeof(double);
        }
        return;
    }
    if (h->family != FAM_MVN_FULL || c.D != h->d || h->d > 32 || c.fuse != 0 || c.schedule != DEMC_SCHED_TWO_COLOUR ||
        c.partner_kind != DEMC_PARTNER_CURRENT || c.Np < 4)
        return;
    if (const char* e = experiment("DEMC_LEAN"))  // A/B experiments
        if (e[0] == '0') return;
    if (h->n_seg < 1) return;  // (the prior table must fit its run-length form)
    for (const DimTab& t : h->h_tab)
        if (t.kind == PR_NORMAL_REF) return;  // (hierarchical scale priors: the general kernel)
    const int nact_max = c.Np - c.Np / 2;
    if (nact_max * 4 > 512) return;
    // (DIRECT at D = 8: 512 threads whatever the group's size -- two waves per SIMD for the residual loop, which one wave per SIMD runs
    // at half the vector pipe's rate: nothing else hides its LDS reads and dependent FP64 pairs)
    const bool dir8 = c.loglike_mode == DEMC_LOGLIKE_DIRECT && c.D == 8 && nact_max * 4 <= 256;
    const int wg = (nact_max * 4 > 256 || dir8) ? 512 : 256;
    const size_t D = (size_t)c.D, Np = (size_t)c.Np;
    const size_t base = (Np * D + Np + (size_t)nact_max + 16 + (size_t)(wg / 4) * (D + 2)) * sizeof(double);
    if (c.loglike_mode == DEMC_LOGLIKE_SUFFSTAT) {
        if (base > kMaxDynLds || !h->res_ok) return;
        h->lean_ok = true; h->lean_wg = wg; h->lean_lds = base;
        return;
    }
    // STREAMING: only where the streaming-resident form applies (plan_stream: small populations), and only with one wave per
    // SIMD (256 threads: what the observation stage needs beyond 256 VGPRs then lives in AGPRs, not scratch)
    const bool dir = c.loglike_mode == DEMC_LOGLIKE_DIRECT;
    if (!(dir ? h->st_dir_geo : h->st_ok) || (wg != 256 && !dir8)) return;
    // (DIRECT: the instances with the row length compiled in, the chunk of whitened rows in LDS, MvNormal-full)
    if (dir && !(h->family == FAM_MVN_FULL && h->n_seg == 1 && (c.D == 8 || c.D == 32) && h->dp_direct == c.D && h->dpad == c.D && h->st_x_lds)) return;
    size_t bytes = base + ((size_t)(wg / 4) * h->dpad + (size_t)(dir ? wg / 16 : wg / 64) * nact_max) * sizeof(double) + 16;  // (DIRECT: a partial sum per 16-lane row)
    if (bytes > kMaxDynLds) return;
    const size_t xbytes = (size_t)(h->st_chunk_tiles + 1) * (h->dpad / 4) * 64 * sizeof(double);
    // (the X chunk rides in LDS exactly when plan_stream found room for it; this kernel's other buffers are no larger)
    if (h->st_x_lds) {
        if (bytes + xbytes > kMaxDynLds) return;
        bytes += xbytes;
    }
    h->lean_stream_ok = !dir; h->lean_direct_ok = dir; h->lean_wg = wg; h->lean_stream_lds = bytes;
    h->lean_st_C = h->st_C; h->lean_st_chunk_tiles = h->st_chunk_tiles; h->lean_st_x_lds = h->st_x_lds; h->lean_st_occ = 1;
    if (dir) return;
#ifdef DEMC_EXPERIMENTS
    // A/B builds only (DEMC_LEAN_OCC2=1) -- TWO workgroups per CU (D = 8, the instance compiled for it): twice the chunks, so that
    // the grid is twice the CU count and a CU holds workgroups of two different groups ("two latency chains on a
    // CU") -- taken when BOTH are resident at once by the runtime's own occupancy figure (every workgroup of the grid must be: the
    // chunks of a group spin on each other's hand-over).  Measured in round 5 at BASELINE cfg2 (profiles/r05/NOTES.md): 0.1664 ms
    // per launch against 0.1407 with one workgroup per CU (0.257 / 0.304 of the matrix peak; past burn-in 0.291 / 0.343): the
    // two workgroups of a CU run in step, their matrix stages collide, and sixteen chunks make the hand-over longer.  Not shipped.
    {
        const int gg = h->geo_groups > c.n_groups ? h->geo_groups : c.n_groups;
        const int C2 = 2 * h->st_C;
        bool want = h->n_seg == 1 && c.D == 8 && h->st_x_lds && C2 <= 16 && (long long)C2 * gg <= 2LL * h->n_cus && h->n_tiles / C2 >= 8;
        const char* e = experiment("DEMC_LEAN_OCC2");
        want = want && e && e[0] == '1';
        if (want) {
            const int chunk2 = (h->n_tiles + C2 - 1) / C2;
            const size_t bytes2 = bytes - xbytes + (size_t)(chunk2 + 1) * (h->dpad / 4) * 64 * sizeof(double);
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_res_mvn<256, true, 8, 0, 2>, 256, bytes2) == hipSuccess && nb >= 2) {
                h->lean_st_C = C2; h->lean_st_chunk_tiles = chunk2; h->lean_st_occ = 2; h->lean_stream_lds = bytes2;
            }
        }
    }
#endif
}

int launch_lean(demc_handle* h, long long iter0, int n_iters, bool stream) {
    const demc_config& c = h->c;
    KParams k = base_params(h);
    k.iter = iter0; k.n_iters = n_iters; k.n_rows = h->hist ? c.n_rows : 0;
    k.sx = stream ? nullptr : h->sx;
    k.Ainv = h->Ainv;
    if (stream) {
        k.st_C = h->lean_st_C; k.st_nact_max = h->st_nact_max; k.st_x_lds = h->lean_st_x_lds; k.st_chunk_tiles = h->lean_st_chunk_tiles;
        k.n_tiles = h->n_tiles; k.Xf = h->Xf; k.st_gran = h->st_gran; k.st_err = h->st_err;
        k.st_rows = 0;
        if (const char* e = experiment("DEMC_OCC2_DELAY")) k.st_rows = std::atoi(e);  // A/B experiments (k_res_mvn<...,OCC 2>: start offset in cycles)
        if (k.n_groups * h->lean_st_C > h->n_cus * h->lean_st_occ) return fail(h, DEMC_EINVAL, "streaming-resident grid exceeds what is resident at once");
        HIPCHK(hipMemsetAsync(h->st_gran, 0, 2 * (size_t)c.n_groups * h->lean_st_C * h->st_nact_max * 2 * sizeof(unsigned long long), h->stream));
    }
    tick(h, 0, true);
    const unsigned grid = (unsigned)(k.n_groups * (stream ? h->lean_st_C : 1));
    const size_t lds = stream ? h->lean_stream_lds : h->lean_lds;
    // instances with the row length folded in (cfg3: 32, cfg2: 8) when the prior table is one segment
    const int dt = (h->n_seg == 1 && (c.D == 32 || c.D == 8)) ? c.D : 0;
    h->last = demc_handle::LastPlan();
    h->last.k1 = 4; h->last.wg = h->lean_wg; h->last.stream = stream; h->last.dt = dt;
    h->last.big = stream && h->lean_direct_ok;  // (the DIRECT instance: named below)
    void (*fn)(KParams) = nullptr;
#ifdef DEMC_EXPERIMENTS
    if (stream && h->lean_st_occ == 2) fn = k_res_mvn<256, true, 8, 0, 2>;  // (plan_lean: D = 8 only)
    else
#endif
    if (stream && h->lean_direct_ok) fn = dt == 8 ? k_res_mvn<512, true, 8, 0, 1, false, true> : k_res_mvn<256, true, 32, 0, 1, false, true>;
    else if (stream) fn = dt == 8 ? k_res_mvn<256, true, 8> : dt == 32 ? k_res_mvn<256, true, 32> : k_res_mvn<256, true, 0>;
    else if (h->lean_wg == 512) fn = dt == 8 ? k_res_mvn<512, false, 8> : dt == 32 ? k_res_mvn<512, false, 32> : k_res_mvn<512, false, 0>;
    else fn = dt == 8 ? k_res_mvn<256, false, 8> : dt == 32 ? k_res_mvn<256, false, 32> : k_res_mvn<256, false, 0>;
    LAUNCH_T(h, fn, dim3(grid), dim3(h->lean_wg), lds, k);
    tick(h, 0, false);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, DEMC_EHIP, std::string("lean resident launch: ") + hipGetErrorString(e));
    return DEMC_OK;
}

// the default sampler on a per-observation family: every iteration up to the next migration in one launch of k_res_obs
int launch_lean_obs(demc_handle* h, long long iter0, int n_iters) {
    const demc_config& c = h->c;
    KParams k = base_params(h);
    k.iter = iter0; k.n_iters = n_iters; k.n_rows = h->hist ? c.n_rows : 0;
    h->last = demc_handle::LastPlan();
    h->last.k1 = 6; h->last.wg = 256;
    tick(h, 0, true);
    LAUNCH_T(h, k_res_obs<256>, dim3((unsigned)k.n_groups), dim3(256), h->lean_obs_lds, k);
    tick(h, 0, false);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, DEMC_EHIP, std::string("lean per-observation launch: ") + hipGetErrorString(e));
    return DEMC_OK;
}

// DE-MC_Z: ONE iteration of every group (both halves) in the lean body, partner rows from the history
int launch_lean_hist(demc_handle* h, long long iter, bool snooker) {
    const demc_config& c = h->c;
    KParams k = base_params(h);
    k.iter = iter; k.n_iters = 1; k.n_rows = h->hist ? c.n_rows : 0;
    k.sx = h->sx;
    k.Ainv = h->Ainv;
    tick(h, 0, true);
    const bool iso = h->family == FAM_MVN_ISO;
    // (ISO: the row length of the reference's own test -- 30 means and sigma -- has an instance with D compiled in)
    // (... which reads the prior table as [one entry for the 30 means | one for sigma]: any other table takes the general row length)
    const int dt = iso ? ((c.D == 31 && h->n_seg == 2 && h->seg_start[1] == 30) ? 31 : 0) : (h->n_seg == 1 && (c.D == 32 || c.D == 8)) ? c.D : 0;
    h->last = demc_handle::LastPlan();
    const bool base = iter <= c.burnin;  // random_gamma reads a base particle (crossover.jl:164): the instance that loads its row
    h->last.k1 = 4; h->last.wg = h->lean_wg; h->last.stream = 0; h->last.dt = dt; h->last.hist = snooker ? 3 : base ? 2 : 1;
    h->last.iso = iso;
    void (*fn)(KParams) = nullptr;
    if (iso) {
        const int hi = snooker ? 3 : base ? 2 : 1;
        static void (*const tab[2][2][3])(KParams) = {
            {{k_res_mvn<256, false, 0, 1, 1, true>, k_res_mvn<256, false, 0, 2, 1, true>, k_res_mvn<256, false, 0, 3, 1, true>},
             {k_res_mvn<256, false, 31, 1, 1, true>, k_res_mvn<256, false, 31, 2, 1, true>, k_res_mvn<256, false, 31, 3, 1, true>}},
            {{k_res_mvn<512, false, 0, 1, 1, true>, k_res_mvn<512, false, 0, 2, 1, true>, k_res_mvn<512, false, 0, 3, 1, true>},
             {k_res_mvn<512, false, 31, 1, 1, true>, k_res_mvn<512, false, 31, 2, 1, true>, k_res_mvn<512, false, 31, 3, 1, true>}}};
        fn = tab[h->lean_wg == 512 ? 1 : 0][dt == 31 ? 1 : 0][hi - 1];
    } else if (snooker && h->lean_wg == 512) fn = dt == 8 ? k_res_mvn<512, false, 8, 3> : dt == 32 ? k_res_mvn<512, false, 32, 3> : k_res_mvn<512, false, 0, 3>;
    else if (snooker) fn = dt == 8 ? k_res_mvn<256, false, 8, 3> : dt == 32 ? k_res_mvn<256, false, 32, 3> : k_res_mvn<256, false, 0, 3>;
    else if (h->lean_wg == 512 && !base) fn = dt == 8 ? k_res_mvn<512, false, 8, 1> : dt == 32 ? k_res_mvn<512, false, 32, 1> : k_res_mvn<512, false, 0, 1>;
    else if (h->lean_wg == 512) fn = dt == 8 ? k_res_mvn<512, false, 8, 2> : dt == 32 ? k_res_mvn<512, false, 32, 2> : k_res_mvn<512, false, 0, 2>;
    else if (!base) fn = dt == 8 ? k_res_mvn<256, false, 8, 1> : dt == 32 ? k_res_mvn<256, false, 32, 1> : k_res_mvn<256, false, 0, 1>;
    else fn = dt == 8 ? k_res_mvn<256, false, 8, 2> : dt == 32 ? k_res_mvn<256, false, 32, 2> : k_res_mvn<256, false, 0, 2>;
    // (the HIST instances hold back the first half's stores for ONE iteration's store_row and form the cdf once per launch: the
    // precondition is checked here, and again by the kernel -- its phase loop runs no trip for any other count)
    if (k.n_iters != 1) return fail(h, DEMC_EINVAL, "lean DE-MC_Z kernel: one iteration per launch");
    LAUNCH_T(h, fn, dim3((unsigned)k.n_groups), dim3(h->lean_wg), h->lean_hist_lds, k);
    tick(h, 0, false);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, DEMC_EHIP, std::string("lean DE-MC_Z launch: ") + hipGetErrorString(e));
    return DEMC_OK;
}

// ---- streaming-resident form: resident K1 with the observation stream inside (k_propose<..., STREAM>) ----
// 256-thread form: one wave per SIMD, i.e. the whole register file (512 per lane, AGPRs included) behind each wave -- what does
// not fit the 256 architectural VGPRs spills to AGPRs instead of scratch memory; 512 threads when a colour needs the lanes.
K1Fn k1_stream_instance(int wg, int tail, int lean) {
    static const K1Fn tab[2][3][2] = {
        {{k_propose<256, true, TAIL_PREP, true, 0, true>, k_propose<256, true, TAIL_PREP_MFMA, true, 0, true>},
         {k_propose<256, true, TAIL_PREP, true, 1, true>, k_propose<256, true, TAIL_PREP_MFMA, true, 1, true>},
         {k_propose<256, true, TAIL_PREP, true, 2, true>, k_propose<256, true, TAIL_PREP_MFMA, true, 2, true>}},
        {{k_propose<512, true, TAIL_PREP, true, 0, true>, k_propose<512, true, TAIL_PREP_MFMA, true, 0, true>},
         {k_propose<512, true, TAIL_PREP, true, 1, true>, k_propose<512, true, TAIL_PREP_MFMA, true, 1, true>},
         {k_propose<512, true, TAIL_PREP, true, 2, true>, k_propose<512, true, TAIL_PREP_MFMA, true, 2, true>}}};
    return tab[wg == 512 ? 1 : 0][lean][tail == TAIL_PREP_MFMA ? 1 : 0];
}

// Decides once per model whether the streaming-resident form applies.  It is for populations too small to fill the chip
// with one K1 -> K2 -> K3 chain per colour phase (launch- and latency-bound): MvNormal family, STREAMING likelihood,
// two_colour, current-population partners, at most one group per CU, the group and its per-particle scratch in LDS, and a
// colour phase whose matrix work is in the launch-overhead range.  Large populations keep the per-phase chain, whose
// k_cross_mfma runs at the matrix peak.
void plan_stream(demc_handle* h) {
    const demc_config& c = h->c;
    h->st_ok = false;
    h->st_dir_geo = false;
    // (DIRECT: the same geometry -- chunks of the observations per workgroup, hand-over granules -- serves the lean kernel's DIRECT
    // instance only, plan_lean; the general streaming-resident kernel has no such form, so st_ok stays false)
    const bool direct = c.loglike_mode == DEMC_LOGLIKE_DIRECT;
    if (!is_mvn(h->family) || (c.loglike_mode != DEMC_LOGLIKE_STREAMING && !direct) || c.fuse != 0 || c.schedule != DEMC_SCHED_TWO_COLOUR ||
        c.partner_kind != DEMC_PARTNER_CURRENT || c.Np < 4 || h->n_cus < 1 || h->geo_groups > h->n_cus || c.n_groups > h->n_cus ||
        h->dpad > 64 || h->n_kpass != 1)
        return;
    // Whether the form applies and into how many chunks C a group's observation tiles are cut is decided from the groups of
    // the WHOLE population (geometry_groups), not from this shard's: C fixes the summation order of the cross terms, and a
    // shard must make the choices of the unsharded run to reproduce it bit for bit (demc_create_multi, demc.h).
    const int gg = h->geo_groups > c.n_groups ? h->geo_groups : c.n_groups;
    if (const char* e = experiment("DEMC_STREAM_RES"))  // A/B experiments
        if (e[0] == '0') return;
    const int nact_max = c.Np - c.Np / 2;
    if (nact_max > 512) return;
    const double phase_flop = 2.0 * (double)nact_max * gg * (double)h->N * h->dpad;
    if (phase_flop / 78.6e12 > 300e-6) return;
    int lpp_max = pow2_ceil((c.D + 1) / 2);
    if (lpp_max > 64) return;
    int lpp = 4;
    while (lpp * 2 <= lpp_max && nact_max * lpp * 2 <= 512) lpp *= 2;
    if (lpp > lpp_max) lpp = lpp_max;
    const int wg = nact_max * lpp <= 256 ? 256 : 512, ppp = wg / lpp;
    const int rows = ((nact_max + ppp - 1) / ppp) * ppp;
    int C = 1;
    while (2 * C * gg <= h->n_cus && h->n_tiles / (2 * C) >= 8 && 2 * C <= 32) C *= 2;
    const int chunk = (h->n_tiles + C - 1) / C;
    const size_t D = (size_t)c.D, Np = (size_t)c.Np, d = (size_t)h->d;
    const size_t scr_doubles = (size_t)rows * (D + 6);
    const size_t doubles = Np * D + Np + (Np + (Np + 15) / 16) + ((h->family == FAM_MVN_FULL && h->ainv_lds) ? d * d : 0) + d + scr_doubles +
                           (size_t)rows * h->dpad + (size_t)(wg / 64) * nact_max;
    size_t bytes = doubles * sizeof(double) + (size_t)nact_max * (4 * sizeof(double) + 4 * sizeof(int)) +
                   2 * sizeof(unsigned) * (size_t)C * nact_max + 16;
    if (bytes > kMaxDynLds) return;
    const size_t xbytes = (size_t)(chunk + 1) * (h->dpad / 4) * 64 * sizeof(double);  // + the all-zero tail tile
    const int x_lds = (bytes + xbytes <= kMaxDynLds) ? 1 : 0;
    if (x_lds) bytes += xbytes;
    // hand-over granules [2][n_groups][C][nact_max][2] and the time-out word
    if (h->st_gran) { hipFree(h->st_gran); h->st_gran = nullptr; }
    // (room for 2 C chunks: the lean kernel may cut the tiles twice as fine, plan_lean)
    if (hipMalloc((void**)&h->st_gran, 2 * (size_t)c.n_groups * 2 * C * nact_max * 2 * sizeof(unsigned long long)) != hipSuccess) return;
    if (!h->st_err) {
        if (hipHostMalloc((void**)&h->st_err, sizeof(unsigned), hipHostMallocMapped) != hipSuccess) return;
        *h->st_err = 0u;
    }
    h->st_ok = !direct; h->st_dir_geo = direct;
    h->st_C = C; h->st_nact_max = nact_max; h->st_rows = rows; h->st_x_lds = x_lds; h->st_chunk_tiles = chunk;
    h->st_lpp = lpp; h->st_scr_doubles = (int)scr_doubles; h->st_lds = bytes; h->st_wg = wg;
}

int launch_stream(demc_handle* h, long long iter0, int n_iters) {
    const demc_config& c = h->c;
    KParams k = base_params(h);
    k.lpp = h->st_lpp;
    set_tail_flags(h, k);
    k.sx = nullptr; k.fuse_accept = 1; k.write_prop = k.trace ? 1 : 0;  // STREAMING: the cross term comes from the tiles
    k.iter = iter0; k.n_iters = n_iters; k.n_sweeps = c.n_blocks > 0 ? c.n_blocks : 1;
    k.mask = c.n_blocks > 0 ? h->masks : nullptr;
    k.n_rows = h->hist ? c.n_rows : 0;
    k.n_split = 1; k.exclude_self = 0; k.own_in_pool = 1; k.tile_rows = c.Np; k.tile_in_lds = 1;
    k.scr_doubles = h->st_scr_doubles;
    k.plan = (k.lpp >= 4) ? 1 : 0;
    k.st_C = h->st_C; k.st_nact_max = h->st_nact_max; k.st_rows = h->st_rows; k.st_x_lds = h->st_x_lds;
    k.st_chunk_tiles = h->st_chunk_tiles; k.n_tiles = h->n_tiles; k.Xf = h->Xf; k.st_gran = h->st_gran; k.st_err = h->st_err;
    if (c.n_groups * h->st_C > h->n_cus) return fail(h, DEMC_EINVAL, "streaming-resident grid exceeds the CU count");
    // epoch tags restart at 1 in every launch: the granules of the previous launch must not match them
    HIPCHK(hipMemsetAsync(h->st_gran, 0, 2 * (size_t)c.n_groups * h->st_C * h->st_nact_max * 2 * sizeof(unsigned long long), h->stream));
    tick(h, 0, true);
    // Every workgroup of this grid must be resident at once (the workgroups of a group spin on each other's progress).  The
    // grid is sized for that by construction -- at most one workgroup per CU (plan_stream: n_groups * C <= CUs, and each
    // takes most of a CU's LDS) -- so a plain launch has the same residency as a cooperative one, without its launch-time
    // cost (+15-19 us, MI355X_MICROARCH.md "coop-launch"); every spin in the kernel is bounded regardless.
    h->last = demc_handle::LastPlan();
    h->last.k1 = 3; h->last.wg = h->st_wg; h->last.tile = 1; h->last.tail = tail_of(k); h->last.plain = lean_level(h, k); h->last.stream = 1;
    LAUNCH_T(h, k1_stream_instance(h->st_wg, tail_of(k), lean_level(h, k)), dim3(c.n_groups * h->st_C), dim3(h->st_wg), h->st_lds,
                       k);
    const hipError_t e = hipGetLastError();
    tick(h, 0, false);
    if (e != hipSuccess) return fail(h, DEMC_EHIP, std::string("streaming-resident launch: ") + hipGetErrorString(e));
    return DEMC_OK;
}

// one sweep of every group: mutate_or_crossover! for all groups (main.jl:161-167, 199-207)
int run_sweep(demc_handle* h, long long iter, unsigned sweep, const unsigned char* mask, long long store_row) {
    const int Np = h->c.Np;
    KParams k = base_params(h);
    k.iter = iter; k.sweep = sweep; k.mask = mask; k.store_row = store_row;
    if (mask && sweep < h->mask_runs.size()) {
        const auto& mr = h->mask_runs[sweep];
        k.n_mrun = mr.n; k.mrun_in = mr.in;
        std::memcpy(k.mrun_start, mr.start, sizeof k.mrun_start);
    } else if (mask)
        k.n_mrun = 0;
    if (h->c.schedule == DEMC_SCHED_TWO_COLOUR) {
        const int half = Np / 2;
        k.a_lo = 0; k.n_act = half; k.pool_lo = half; k.pool_n = Np - half; k.exclude_self = 0;
        int rc = launch_phase(h, k);
        if (rc != DEMC_OK) return rc;
        k.a_lo = half; k.n_act = Np - half; k.pool_lo = 0; k.pool_n = half;
        return launch_phase(h, k);
    }
    if (h->c.schedule == DEMC_SCHED_SEQUENTIAL) {
        // crossover.jl:13-15 / mutation.jl:16: particle pl of every group moves after 0..pl-1 were updated in place;
        // the groups advance side by side (one task per group in p_update!, main.jl:135-148).  Partners come from the
        // whole group minus self (setdiff, crossover.jl:158), snooker's three from the whole group (crossover.jl:241).
        for (int pl = 0; pl < Np; ++pl) {
            k.a_lo = pl; k.n_act = 1; k.pool_lo = 0; k.pool_n = Np; k.exclude_self = 1;
            int rc = launch_phase(h, k);
            if (rc != DEMC_OK) return rc;
        }
        return DEMC_OK;
    }
    return launch_phase(h, k);
}

int migration_enqueue(demc_handle* h, long long iter, double* dev_rows, const double* dev_all_rows, bool pack, bool apply) {
    KParams k = base_params(h);
    k.iter = iter;
    tick(h, 4, true);
    if (pack)
        LAUNCH_T(h, k_mig_pack, dim3(h->c.n_groups), dim3(256),
                           sizeof(double) * ((size_t)h->c.Np + ((size_t)h->c.Np + 15) / 16), k, dev_rows);
    if (apply) {
        const int ngt = h->c.n_groups_total;
        const int grid = ngt < 64 ? ngt : 64;
        LAUNCH_T(h, k_mig_apply, dim3(grid), dim3(256), 2 * sizeof(int) * (size_t)ngt, k, dev_all_rows, ngt);
    }
    tick(h, 4, false);
    return DEMC_OK;
}

int evaluate_rows(demc_handle* h, double* theta_dev, double* weight_dev) {
    // init_particle: evaluate_fitness! on the current rows (utilities.jl:13-22) -- identity proposal, forced accept
    KParams k = base_params(h);
    k.mode = MODE_IDENT; k.theta = theta_dev; k.weight = weight_dev; k.store_row = -1; k.iter = 0;
    k.tile_in_lds = 0;
    return launch_phase(h, k);
}

// The by-product snapshot (KParams::snap_theta / snap_weight: a frozen sweep over the whole population leaves the rows it streamed
// behind as the NEXT sweep's sweep-start snapshot) costs another P x D doubles -- 328 MB at cfg4, the size of theta itself.  The
// buffers exist exactly while the configuration can take that path (DE-MC_Z inside burn-in, long hierarchical rows, more than one
// block, enough particles for the row-streaming kernel): allocated here, when demc_set_model / demc_set_blocks learn it -- not by a
// hipMalloc in the middle of an enqueued step -- and freed when the model or the blocks change so that it no longer can.  Out of
// memory is not an error (the sweeps copy the population instead); demc_last_error then says so.
int plan_snap2(demc_handle* h) {
    const demc_config& c = h->c;
    const bool hier = h->family == FAM_HIER_BINOMIAL || h->family == FAM_HIER_GAUSSIAN;
    const bool want = hier && h->hier_scr && h->lpp > 64 && c.n_blocks > 1 && c.schedule == DEMC_SCHED_SYNCHRONOUS &&
                      c.partner_kind == DEMC_PARTNER_HISTORY && c.proposal_kind == 0 && c.burnin >= 1 && c.fuse == 0 && !c.trace &&
                      c.kappa == 1.0 && (long long)h->geo_groups * c.Np >= 2LL * h->n_cus;
    if (!want) {
        if (h->snap2 || h->snap2_w) {
            HIPCHK(hipStreamSynchronize(h->stream));
            if (h->snap2) hipFree(h->snap2);
            if (h->snap2_w) hipFree(h->snap2_w);
            h->snap2 = nullptr; h->snap2_w = nullptr; h->snap2_iter = -1;
        }
        return DEMC_OK;
    }
    if (h->snap2) return DEMC_OK;
    const size_t bytes = sizeof(double) * (size_t)h->P * c.D;
    if (hipMalloc((void**)&h->snap2, bytes) != hipSuccess) { h->snap2 = nullptr; (void)hipGetLastError(); }
    if (h->snap2 && hipMalloc((void**)&h->snap2_w, sizeof(double) * (size_t)h->P) != hipSuccess) {
        hipFree(h->snap2); h->snap2 = nullptr; h->snap2_w = nullptr; (void)hipGetLastError();
    }
    if (!h->snap2)
        h->err = "note: no memory for the by-product snapshot (" + std::to_string(bytes) + " bytes): DE-MC_Z sweeps inside burn-in copy the population instead";
    return DEMC_OK;
}

// K1 LDS carve-up (must match k_propose): group tile (if it fits) | Np prefix sums | A^-1 [d][d] | theta' scratch
int size_k1_lds(demc_handle* h) {
    const demc_config& c = h->c;
    const size_t D = (size_t)c.D;
    const size_t cdf = ((size_t)c.Np + ((size_t)c.Np + 15) / 16) * sizeof(double);
    size_t ainv = (h->family == FAM_MVN_FULL) ? (size_t)h->d * h->d * sizeof(double) : 0;
    h->ainv_lds = ainv <= 40 * 1024 ? 1 : 0;  // d <= 71: beside the tile; wider data read A^-1 from L2
    if (!h->ainv_lds) ainv = 0;
    const size_t xb = is_mvn(h->family) ? (size_t)h->d * sizeof(double) : 0;
    const size_t scr_rows = (size_t)(h->lpp >= 256 ? 1 : 256 / h->lpp) * (D + 2) * sizeof(double);
    const bool hier = h->family == FAM_HIER_BINOMIAL || h->family == FAM_HIER_GAUSSIAN;
    h->hier_scr = hier && scr_rows <= 96 * 1024;  // theta' of the pass fits in LDS: the subjects can be summed in K1
    const bool scr_fam = is_mvn(h->family) || h->family == FAM_GAUSSIAN || h->family == FAM_BINOMIAL ||
                         h->family == FAM_RASTRIGIN || h->hier_scr;
    const size_t scr = scr_fam ? scr_rows : 0;
    const size_t tile = (size_t)c.Np * D * sizeof(double);
    h->tile_in_lds = (tile + cdf + ainv + xb + scr <= 96 * 1024) ? 1 : 0;
    if (const char* e = experiment("DEMC_K1_TILE")) h->tile_in_lds = (e[0] == '1') && h->tile_in_lds;  // A/B experiments
    h->k1_tile_bytes = h->tile_in_lds ? tile : 0;
    h->k1_scr_bytes = scr;
    h->k1_lds = h->k1_tile_bytes + cdf + ainv + xb + scr;
    if (h->k1_lds > kMaxDynLds) return fail(h, DEMC_EINVAL, "K1 LDS budget exceeded (Np too large for this D)");
    // the attribute is per function, not per handle: always raise it to the same ceiling so that handles of different
    // sizes in one process do not lower each other's limit
    for (int t = 0; t < 2; ++t)
        for (int tail = 0; tail < 4; ++tail)
            for (int lean = 0; lean < 3; ++lean) {
                HIPCHK(hipFuncSetAttribute((const void*)k1_instance(t != 0, tail, lean),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds));
                HIPCHK(hipFuncSetAttribute((const void*)k1_resident_instance(t ? 512 : 256, tail, lean),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds));
                HIPCHK(hipFuncSetAttribute((const void*)k1_instance(false, tail, 0, 512),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds));
            }
    for (int wgs = 256; wgs <= 512; wgs += 256)
        for (int tail = 1; tail <= 2; ++tail)
            for (int lean = 0; lean < 3; ++lean)
                HIPCHK(hipFuncSetAttribute((const void*)k1_stream_instance(wgs, tail, lean),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds));
#ifdef DEMC_EXPERIMENTS
    h->lba_wide_ok = kLbaTableBytes <= kMaxDynLds &&
                     hipFuncSetAttribute((const void*)k_lba_loglike<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLbaTableBytes) == hipSuccess;
#endif
    HIPCHK(hipFuncSetAttribute((const void*)k_longrow<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds));
    HIPCHK(hipFuncSetAttribute((const void*)k_longrow<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds));
#ifdef DEMC_EXPERIMENTS
    HIPCHK(hipFuncSetAttribute((const void*)k_longrow<384, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds));
    HIPCHK(hipFuncSetAttribute((const void*)k_longrow<512, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds));
#endif
    {
        void (*lean[])(KParams) = {k_res_mvn<256, false, 0>, k_res_mvn<256, false, 8>, k_res_mvn<256, false, 32>,
                                   k_res_mvn<512, false, 0>, k_res_mvn<512, false, 8>, k_res_mvn<512, false, 32>,
                                   k_res_mvn<256, true, 0>,  k_res_mvn<256, true, 8>,  k_res_mvn<256, true, 32>,
                                   k_res_mvn<512, true, 8, 0, 1, false, true>, k_res_mvn<256, true, 32, 0, 1, false, true>,
#ifdef DEMC_EXPERIMENTS
                                   k_res_mvn<256, true, 8, 0, 2>,
#endif
                                   k_res_mvn<256, false, 0, 1>, k_res_mvn<256, false, 8, 1>, k_res_mvn<256, false, 32, 1>,
                                   k_res_mvn<512, false, 0, 1>, k_res_mvn<512, false, 8, 1>, k_res_mvn<512, false, 32, 1>,
                                   k_res_mvn<256, false, 0, 2>, k_res_mvn<256, false, 8, 2>, k_res_mvn<256, false, 32, 2>,
                                   k_res_mvn<512, false, 0, 2>, k_res_mvn<512, false, 8, 2>, k_res_mvn<512, false, 32, 2>,
                                   k_res_mvn<256, false, 0, 3>, k_res_mvn<256, false, 8, 3>, k_res_mvn<256, false, 32, 3>,
                                   k_res_mvn<512, false, 0, 3>, k_res_mvn<512, false, 8, 3>, k_res_mvn<512, false, 32, 3>,
                                   k_res_mvn<256, false, 0, 1, 1, true>, k_res_mvn<256, false, 0, 2, 1, true>, k_res_mvn<256, false, 0, 3, 1, true>,
                                   k_res_mvn<512, false, 0, 1, 1, true>, k_res_mvn<512, false, 0, 2, 1, true>, k_res_mvn<512, false, 0, 3, 1, true>,
                                   k_res_mvn<256, false, 31, 1, 1, true>, k_res_mvn<256, false, 31, 2, 1, true>, k_res_mvn<256, false, 31, 3, 1, true>,
                                   k_res_mvn<512, false, 31, 1, 1, true>, k_res_mvn<512, false, 31, 2, 1, true>, k_res_mvn<512, false, 31, 3, 1, true>};
        for (auto f : lean) HIPCHK(hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds));
        HIPCHK(hipFuncSetAttribute((const void*)k_res_obs<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds));
    }
    plan_resident(h);
    plan_stream(h);
    plan_lean(h);
    HIPCHK(hipFuncSetAttribute((const void*)k_mig_pack, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds));
    return plan_snap2(h);
}

void free_replay(demc_handle* h) {
    void* ptrs[] = {h->rp_group, h->rp_part, h->rp_noise, h->rp_znoise, h->rp_recomb, h->rp_partner, h->rp_mig_particle, h->rp_mig_groups};
    for (void* p : ptrs)
        if (p) hipFree(p);
    h->rp_group = h->rp_part = h->rp_noise = h->rp_znoise = h->rp_recomb = nullptr;
    h->rp_partner = h->rp_mig_particle = nullptr;
    h->rp_mig_groups = nullptr;
    h->rp_n_mig = 0;
    h->rp_active = h->rp_has_step = false;
}

bool chol_inv(const double* S, int d, std::vector<double>& Ainv, double& logdet, std::vector<double>* Linv = nullptr) {
    std::vector<double> L((size_t)d * d, 0.0), Li((size_t)d * d, 0.0);
    for (int i = 0; i < d; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = S[i * d + j];
            for (int k = 0; k < j; ++k) s -= L[i * d + k] * L[j * d + k];
            if (i == j) {
                if (!(s > 0.0)) return false;
                L[i * d + i] = std::sqrt(s);
            } else
                L[i * d + j] = s / L[j * d + j];
        }
    logdet = 0.0;
    for (int i = 0; i < d; ++i) logdet += 2.0 * std::log(L[i * d + i]);
    for (int c = 0; c < d; ++c)  // Li = L^-1 by forward substitution on unit vectors
        for (int i = 0; i < d; ++i) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = 0; k < i; ++k) s -= L[i * d + k] * Li[k * d + c];
            Li[i * d + c] = s / L[i * d + i];
        }
    Ainv.assign((size_t)d * d, 0.0);
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            double s = 0.0;
            for (int k = 0; k < d; ++k) s += Li[k * d + i] * Li[k * d + j];
            Ainv[i * d + j] = s;
        }
    if (Linv) *Linv = Li;
    return true;
}

}  // namespace

extern "C" {

static int upload_dimtab(demc_handle* h);  // (defined next to demc_set_priors)

int32_t demc_version(void) { return DEMC_VERSION; }

const char* demc_last_error(demc_handle* h) { return h ? h->err.c_str() : "null handle"; }

int32_t demc_create(const demc_config* cfg, demc_handle** out) {
    if (!cfg || !out) return DEMC_EINVAL;
    *out = nullptr;
    demc_handle* h = new (std::nothrow) demc_handle();
    if (!h) return DEMC_ENOMEM;
    *out = h;  // returned even on failure so that demc_last_error() can be read; caller destroys it
    return guarded(h, [&]() -> int32_t {
    h->c = *cfg;
    demc_config& c = h->c;
    if (c.n_groups < 1 || c.Np < 3 || c.D < 1 || c.n_rows < 0)
        return fail(h, DEMC_EINVAL, "need n_groups >= 1, Np >= 3 (structs.jl:43), D >= 1, n_rows >= 0");
    if (c.n_groups_total <= 0) c.n_groups_total = c.n_groups;
    if (c.n_groups_total == 1) c.alpha = 0.0;  // structs.jl:102-105
    if (c.schedule != DEMC_SCHED_SEQUENTIAL && c.schedule != DEMC_SCHED_SYNCHRONOUS && c.schedule != DEMC_SCHED_TWO_COLOUR)
        return fail(h, DEMC_EINVAL, "unknown schedule");
    if (c.geometry_groups < 0) return fail(h, DEMC_EINVAL, "geometry_groups < 0");
    if (c.loglike_mode < DEMC_LOGLIKE_STREAMING || c.loglike_mode > DEMC_LOGLIKE_DIRECT) return fail(h, DEMC_EINVAL, "unknown loglike_mode");
    h->geo_groups = c.geometry_groups > 0 ? c.geometry_groups : c.n_groups;
    if (c.schedule == DEMC_SCHED_TWO_COLOUR && c.partner_kind == DEMC_PARTNER_CURRENT) {
        const int need = c.theta_snooker > 0.0 ? 6 : 4;
        if (c.Np < need) return fail(h, DEMC_EINVAL, "two_colour needs Np >= 4 (>= 6 with snooker)");
    }
    if (c.proposal_kind < 0 || c.proposal_kind > 2 || c.partner_kind < 0 || c.partner_kind > 1 || c.update_kind < 0 ||
        c.update_kind > 2 || c.fitness_kind < 0 || c.fitness_kind > 1)
        return fail(h, DEMC_EUNSUPPORTED, "hook outside the registered set (structs.jl:71-74): no fallback");
    if (c.partner_kind == DEMC_PARTNER_HISTORY && (!c.store_history || c.n_initial < 1))
        return fail(h, DEMC_EINVAL, "history partners (resample) need store_history and n_initial > 0");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(h, DEMC_EHIP, "no HIP device visible");
    HIPCHK(hipSetDevice(c.device_id));
    HIPCHK(hipDeviceGetAttribute(&h->n_cus, hipDeviceAttributeMultiprocessorCount, c.device_id));
    HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
    h->P = (long long)c.n_groups * c.Np;
    const size_t P = (size_t)h->P, D = (size_t)c.D;
    ALLOC(h->theta, P * D); ALLOC(h->weight, P); ALLOC(h->id, P); ALLOC(h->prop, P * D);
    ALLOC(h->prop_prior, P); ALLOC(h->prop_adj, P); ALLOC(h->prop_oob, P);
    ALLOC(h->tr_idx, P * 4); ALLOC(h->tr_w, P); ALLOC(h->tr_acc, P);
    ALLOC(h->partial, (size_t)h->partial_cap * P); ALLOC(h->aux, P);
    ALLOC(h->dimtab, D);
    ALLOC(h->dimseg, (size_t)kMaxDimSeg);
    for (int i = 0; i < demc_handle::kGlistRing; ++i) {
        ALLOC(h->glist_buf[i], (size_t)c.n_groups);
        HIPCHK(hipHostMalloc((void**)&h->glist_pin[i], (size_t)c.n_groups * sizeof(int), hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&h->glist_ev[i], hipEventDisableTiming));
    }
    ALLOC(h->mig_rows, (size_t)c.n_groups_total * (D + 3));
    ALLOC(h->scratch_theta, P * D); ALLOC(h->scratch_w, P);
    if (c.store_history && c.n_rows > 0) {
        // History cells that partners are GATHERED from (DE-MC_Z: `resample`, crossover.jl:113-124) are padded to whole cache lines --
        // rows of up to 16 scalars to the next power of two, up to 64 to the next multiple of 16 doubles: a cell of 31 doubles at an
        // 8-byte boundary touches 2.9 lines of 128 bytes on average (376 B fetched for 248), at a 256-byte boundary exactly two.  Only
        // then: a history that is only written (partners from the population) stays dense -- padded cells would be partial-line stores.
        h->hist_ld = (int)D;
        if (c.partner_kind == DEMC_PARTNER_HISTORY && D <= 64) h->hist_ld = D <= 16 ? pow2_ceil((int)D) : (int)((D + 15) & ~(size_t)15);
        ALLOC(h->hist, (size_t)c.n_rows * P * (size_t)h->hist_ld);
        ALLOC(h->acc_hist, (size_t)c.n_rows * P);
        ALLOC(h->lp_hist, (size_t)c.n_rows * P);
        ALLOC(h->id_hist, (size_t)c.n_rows * P);
    }
    {
        DimTab t0;
        t0.lo = -INFINITY; t0.hi = INFINITY; t0.a = 0.0; t0.b = 1.0; t0.c = 0.0; t0.kind = PR_FLAT; t0.ref = 0;
        h->h_tab.assign(D, t0);
        int rc_tab = upload_dimtab(h);
        if (rc_tab != DEMC_OK) return rc_tab;
        std::vector<long long> id(P);
        for (size_t s = 0; s < P; ++s) id[s] = (long long)c.group_offset * c.Np + (long long)s;
        HIPCHK(hipMemcpy(h->id, id.data(), P * sizeof(long long), hipMemcpyHostToDevice));
        if (h->id_hist) {
            std::vector<int> row(P);
            for (size_t s = 0; s < P; ++s) row[s] = (int)id[s];
            for (long long r = 0; r < c.n_rows; ++r)
                HIPCHK(hipMemcpy(h->id_hist + (size_t)r * P, row.data(), P * sizeof(int), hipMemcpyHostToDevice));
        }
    }
    // lanes per particle: every lane owns dim pairs {2k,2k+1}, k = sl, sl+lpp, ...  Upper end: one pair per lane (or a
    // whole workgroup per particle for very long rows).  Several pairs per lane give each lane independent work to
    // overlap and fewer cross-lane reduction steps, so take the FEWEST lanes (>= 4) that still (i) fill every lane of
    // a pass with a particle and (ii) leave at least two workgroups per CU; small populations keep the widest split
    // (measured at D = 32: 256x256 particles -> 4, 128x256 -> 8, 64x128 and below -> 16).
    h->lpp = pow2_ceil((c.D + 1) / 2);
    if (h->lpp > 64) h->lpp = (c.D >= 4096) ? 512 : (c.D >= 2048) ? 256 : 64;  // a whole workgroup per particle for long rows
    if (h->lpp <= 64) {
        // moving particles per group and launch
        const int n_act = (c.schedule == DEMC_SCHED_TWO_COLOUR) ? c.Np / 2 : (c.schedule == DEMC_SCHED_SEQUENTIAL) ? 1 : c.Np;
        for (int l = 4; l < h->lpp; l *= 2) {
            const int ppp = 256 / l;
            if (ppp <= n_act && (long long)h->geo_groups * ((n_act + ppp - 1) / ppp) >= 512) {
                h->lpp = l;
                break;
            }
        }
    }
    if (const char* e = experiment("DEMC_LPP")) {  // A/B experiments: fewer lanes per particle = less replicated scalar work
        const int v = std::atoi(e);
        if (v >= 1 && v <= 512 && v != 128 && (v & (v - 1)) == 0 && (v <= h->lpp || v == 256 || v == 512)) h->lpp = v;
    }
    int rc_lds = size_k1_lds(h);
    if (rc_lds != DEMC_OK) return rc_lds;
    if (2 * (size_t)c.n_groups_total * sizeof(int) > 48 * 1024) return fail(h, DEMC_EINVAL, "n_groups_total too large");
    // A documented deviation says so at run time: `resample` draws its partner cells from the history of ALL
    // particles (crossover.jl:116-124); a shard holds the history of its own groups only.  Not an error -- SURVEY 8(e) allows the
    // shard-local pool -- but the caller is told: demc_last_error() carries the note after a successful demc_create.
    if (c.partner_kind == DEMC_PARTNER_HISTORY && c.n_groups_total > c.n_groups)
        h->err = "note: sharded handle (" + std::to_string(c.n_groups) + " of " + std::to_string(c.n_groups_total) +
                 " groups) with history partners: DE-MC_Z draws its partner cells from THIS shard's history only "
                 "(the reference draws from all particles, crossover.jl:116-124)";
    return DEMC_OK;
    });
}

int32_t demc_destroy(demc_handle* h) {
    return guarded(nullptr, [&]() -> int32_t {
    if (!h) return DEMC_OK;
    if (h->multi) return DEMC_EINVAL;  // a shard of a multi-GPU set goes with the set (demc_destroy_multi)
    if (h->stream) hipStreamSynchronize(h->stream);
    drain_events(h);
    for (hipEvent_t e : h->event_pool) hipEventDestroy(e);
    h->event_pool.clear();
    void* ptrs[] = {h->theta, h->weight, h->prop, h->prop_prior, h->prop_adj, h->tr_w, h->partial, h->aux,
                    h->dimtab, h->dimseg, h->hist, h->lp_hist, h->mig_rows, h->scratch_theta, h->scratch_w, h->id, h->prop_oob,
                    h->tr_acc, h->masks, h->acc_hist, h->tr_idx, h->id_hist, h->data, h->Ainv, h->Ypad,
                    h->Xf, h->sx, h->xbar};
    for (void* p : ptrs)
        if (p) hipFree(p);
    if (h->user_module) hipModuleUnload(h->user_module);
    if (h->user_hyper) hipFree(h->user_hyper);
    if (h->user_dims) hipFree(h->user_dims);
    free_replay(h);
    for (int i = 0; i < demc_handle::kGlistRing; ++i) {
        if (h->glist_buf[i]) hipFree(h->glist_buf[i]);
        if (h->glist_pin[i]) hipHostFree(h->glist_pin[i]);
        if (h->glist_ev[i]) hipEventDestroy(h->glist_ev[i]);
    }
    if (h->st_gran) hipFree(h->st_gran);
    if (h->frozen_order_d) hipFree(h->frozen_order_d);
    if (h->frozen_order_pin) hipHostFree(h->frozen_order_pin);
    if (h->frozen_order_ev) hipEventDestroy(h->frozen_order_ev);
    if (h->snap2) hipFree(h->snap2);
    if (h->snap2_w) hipFree(h->snap2_w);
    if (h->clk_dev) hipFree(h->clk_dev);
    if (h->st_err) hipHostFree(h->st_err);
    if (h->side) hipStreamSynchronize(h->side);
    if (h->comm && h->own_comm) ncclCommDestroy(h->comm);
    if (h->ev_pack) hipEventDestroy(h->ev_pack);
    if (h->ev_gath) hipEventDestroy(h->ev_gath);
    if (h->side) hipStreamDestroy(h->side);
    if (h->red_dev) hipFree(h->red_dev);
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    delete h;
    return DEMC_OK;
    });
}

int32_t demc_set_stream(demc_handle* h, void* hip_stream) {
    return guarded(h, [&]() -> int32_t {
    if (!h) return DEMC_EINVAL;
    // a shard of a demc_multi set: shards on one device run on the first such shard's stream (demc_create_multi); a caller who
    // re-points one would either destroy a stream its borrowers still hold or put two spinning resident kernels in flight at once
    if (h->multi && h->multi_sealed) return fail(h, DEMC_EINVAL, "demc_set_stream: the streams of a demc_multi set belong to the set");
    USE_DEVICE(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    h->own_stream = false;
    if (hip_stream)
        h->stream = (hipStream_t)hip_stream;
    else {
        HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->own_stream = true;
    }
    return DEMC_OK;
    });
}

int32_t demc_set_model(demc_handle* h, int32_t family, const double* data, const int64_t* dims, int32_t ndims,
                       const double* hyper, int32_t nhyper) {
    return guarded(h, [&]() -> int32_t {
    if (!h) return DEMC_EINVAL;
    USE_DEVICE(h);
    if (ndims < 0 || ndims > 4 || (ndims > 0 && !dims)) return fail(h, DEMC_EINVAL, "bad dims");
    long long dm[4] = {0, 0, 0, 0};
    for (int i = 0; i < ndims; ++i) dm[i] = dims[i];
    const int D = h->c.D;
    for (double** p : {&h->data, &h->Ainv, &h->Ypad, &h->Xf, &h->sx, &h->xbar})
        if (*p) { hipFree(*p); *p = nullptr; }
    h->family = -1; h->N = 0; h->d = 0; h->n_acc = 0; h->dpad = 0; h->n_tiles = 0; h->c0 = h->c1 = h->c2 = 0; h->data2_off = 0;
    h->dp_direct = 0; h->direct_wgs_per_cu = 0;
    std::vector<double> dev;  // what goes to h->data
    switch (family) {
        case DEMC_FAM_GAUSSIAN:
            if (D != 2 || dm[0] < 1) return fail(h, DEMC_EINVAL, "GAUSSIAN: theta=(mu,sigma), dims=[N]");
            h->N = dm[0];
            dev.assign(data, data + dm[0]);
            break;
        case DEMC_FAM_BINOMIAL: {
            if (D != 1 || dm[0] < 1) return fail(h, DEMC_EINVAL, "BINOMIAL: theta=p, dims=[N], data=[n[N],k[N]]");
            const long long N = dm[0];
            h->N = N;
            dev.assign(data, data + 2 * N);
            dev.resize(3 * N);
            for (long long i = 0; i < N; ++i) {
                const double n = data[i], k = data[N + i];
                dev[2 * N + i] = std::lgamma(n + 1.0) - std::lgamma(k + 1.0) - std::lgamma(n - k + 1.0);
            }
            h->data2_off = (size_t)N;
        } break;
        case DEMC_FAM_HIER_BINOMIAL: {
            if (D != dm[0] + 2 || nhyper < 1) return fail(h, DEMC_EINVAL, "HIER_BINOMIAL: D=S+2, dims=[S], hyper=[n]");
            const long long S = dm[0];
            h->N = S;
            h->c0 = hyper[0];
            dev.assign(data, data + S);
            dev.resize(2 * S);
            double lgc_sum = 0.0;
            for (long long s = 0; s < S; ++s) {
                const double n = hyper[0], k = data[s];
                dev[S + s] = std::lgamma(n + 1.0) - std::lgamma(k + 1.0) - std::lgamma(n - k + 1.0);
                lgc_sum += dev[S + s];
            }
            h->c2 = lgc_sum;  // the long-row kernel adds the coefficients as one data-only constant
        } break;
        case DEMC_FAM_HIER_GAUSSIAN:
            if (D != dm[0] + 3 || dm[1] < 1) return fail(h, DEMC_EINVAL, "HIER_GAUSSIAN: D=S+3, dims=[S,n]");
            h->N = dm[0];
            h->d = (int)dm[1];
            dev.assign(data, data + dm[0] * dm[1]);
            break;
        case DEMC_FAM_LBA:
        case DEMC_FAM_LNR: {
            const int extra = (family == DEMC_FAM_LBA) ? 3 : 1;
            if (dm[1] < 1 || dm[1] > 8 || D != dm[1] + extra) return fail(h, DEMC_EINVAL, "LBA/LNR: dims=[N,n_acc<=8]");
            // data = [choice[N] ; rt[N]]: the kernels pick the winning accumulator by comparing the choice code with 1..n_acc
            // (k_lba_wave: exact double equality) and the sort below compares decision times -- a non-integral or out-of-range
            // code would silently make every accumulator a loser, a NaN is no strict weak ordering: refused here
            for (long long i = 0; i < dm[0]; ++i) {
                const double ch = data[i], rt = data[dm[0] + i];
                if (!(ch >= 1.0 && ch <= (double)dm[1] && ch == std::floor(ch)))
                    return fail(h, DEMC_EINVAL, "LBA/LNR: choice of trial " + std::to_string(i) + " is not an integer in [1, n_acc]");
                if (!std::isfinite(rt)) return fail(h, DEMC_EINVAL, "LBA/LNR: decision time of trial " + std::to_string(i) + " is not finite");
            }
            h->N = dm[0];
            h->n_acc = (int)dm[1];
            h->c0 = (family == DEMC_FAM_LNR) ? (nhyper > 0 ? hyper[0] : 1.0) : 0.0;
            dev.assign(data, data + 2 * dm[0]);
            h->data2_off = (size_t)dm[0];
            if (family == DEMC_FAM_LBA) {
                // The log-likelihood is a sum over trials: their order is the library's to choose.  Sorted by (choice, decision time)
                // the 64 lanes of k_lba_wave -- 64 consecutive trials of one proposal -- read the same row or two of the Phi table and
                // share their winner (demc_kernels.hpp).
                const size_t N = (size_t)dm[0];
                std::vector<size_t> order(N);
                for (size_t i = 0; i < N; ++i) order[i] = i;
                std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b2) {
                    return data[a] != data[b2] ? data[a] < data[b2] : data[N + a] < data[N + b2];
                });
                for (size_t i = 0; i < N; ++i) { dev[i] = data[order[i]]; dev[N + i] = data[N + order[i]]; }
            }
        } break;
        case DEMC_FAM_RASTRIGIN:
            h->N = 1;
            break;
        case DEMC_FAM_MVN_ISO:
        case DEMC_FAM_MVN_FULL: {
            const long long N = dm[0];
            const int d = (int)dm[1];
            if (N < 1 || d < 1) return fail(h, DEMC_EINVAL, "MVN: dims=[N,d]");
            if (d > 1024) return fail(h, DEMC_EUNSUPPORTED, "MVN families: data dimension d <= 1024");
            if (family == DEMC_FAM_MVN_ISO && D != d + 1) return fail(h, DEMC_EINVAL, "MVN_ISO: D=d+1");
            if (family == DEMC_FAM_MVN_FULL && (D != d || nhyper != d * d)) return fail(h, DEMC_EINVAL, "MVN_FULL: D=d, hyper=Sigma[d][d]");
            h->N = N;
            h->d = d;
            // k-steps of 4 dims: one pass of KS in {1,2,4,8,16} steps for d <= 64, else passes of 16 steps
            if (d <= 64) { h->ks_t = pow2_ceil((d + 3) / 4); h->n_kpass = 1; }
            else { h->ks_t = 16; h->n_kpass = (d + 63) / 64; }
            h->dpad = 4 * h->ks_t * h->n_kpass;
            if ((N + 15) / 16 > 0x3fffffff) return fail(h, DEMC_EINVAL, "too many observations");
            h->n_tiles = (int)((N + 15) / 16);
            const bool direct = h->c.loglike_mode == DEMC_LOGLIKE_DIRECT;
            if (direct && d > 64) return fail(h, DEMC_EUNSUPPORTED, "DIRECT likelihood: data dimension d <= 64 (the whitened proposal lives in registers)");
            std::vector<double> Ainv, Linv;
            double logdet = 0.0;
            if (family == DEMC_FAM_MVN_FULL) {
                if (!chol_inv(hyper, d, Ainv, logdet, &Linv)) return fail(h, DEMC_EINVAL, "Sigma is not positive definite");
                h->c0 = -0.5 * (double)N * (d * kLog2Pi + logdet);
                ALLOC(h->Ainv, (size_t)d * d);
                // K1's preparation forms y_c = sum_k M[k][c] (theta' - xbar)_k.  M = Sigma^-1 (symmetric) gives y = Sigma^-1 mu~;
                // in DIRECT mode M = (L^-1)' gives the whitened proposal m = L^-1 mu~ instead.
                std::vector<double> M = Ainv;
                if (direct)
                    for (int r = 0; r < d; ++r)
                        for (int cc = 0; cc < d; ++cc) M[(size_t)r * d + cc] = Linv[(size_t)cc * d + r];
                HIPCHK(hipMemcpy(h->Ainv, M.data(), sizeof(double) * d * d, hipMemcpyHostToDevice));
            }
            // The data are CENTRED once (x~_i = x_i - xbar) and proposals are shifted the same way in K1
            // (mu~ = theta' - xbar): (x_i - mu) = (x~_i - mu~), but every term of the expanded quadratic form
            //   sum_i x~_i' A^-1 x~_i  -  2 y . sum_i x~_i  +  N mu~' A^-1 mu~ ,   y = A^-1 mu~
            // is then O(N d) whatever the offset of the data, so the expansion loses no digits to cancellation.
            // data-only constants: c1 = sum_i x~_i' A^-1 x~_i (A = I for ISO), sx = sum_i x~_i (~ 0)
            std::vector<double> xbar(d, 0.0);
            for (long long i = 0; i < N; ++i)
                for (int k = 0; k < d; ++k) xbar[k] += data[i * d + k];
            for (int k = 0; k < d; ++k) xbar[k] /= (double)N;
            std::vector<double> xc((size_t)N * d);
            for (long long i = 0; i < N; ++i)
                for (int k = 0; k < d; ++k) xc[(size_t)i * d + k] = data[i * d + k] - xbar[k];
            data = xc.data();
            std::vector<double> sx(d, 0.0), t(d);
            double c1 = 0.0;
            for (long long i = 0; i < N; ++i) {
                const double* x = data + i * d;
                double q = 0.0;
                if (family == DEMC_FAM_MVN_FULL) {
                    for (int r = 0; r < d; ++r) {
                        double sr = 0.0;
                        for (int k = 0; k < d; ++k) sr += Ainv[r * d + k] * x[k];
                        q += x[r] * sr;
                    }
                } else
                    for (int k = 0; k < d; ++k) q += x[k] * x[k];
                c1 += q;
                for (int k = 0; k < d; ++k) sx[k] += x[k];
            }
            h->c1 = c1;
            ALLOC(h->sx, (size_t)d);
            HIPCHK(hipMemcpy(h->sx, sx.data(), sizeof(double) * d, hipMemcpyHostToDevice));
            ALLOC(h->xbar, (size_t)d);
            HIPCHK(hipMemcpy(h->xbar, xbar.data(), sizeof(double) * d, hipMemcpyHostToDevice));
            ALLOC(h->Ypad, (size_t)h->P * h->dpad);
            // fragment-ordered copy of X for v_mfma_f64_16x16x4_f64's B operand:
            //   Xf[tile][kstep][lane] = X[16*tile + (lane&15)][4*kstep + (lane>>4)], zero padded
            const int ksx = h->dpad / 4;
            std::vector<double> xf(((size_t)h->n_tiles + 1) * ksx * 64, 0.0);  // +1: the all-zero tail tile
            for (long long tI = 0; tI < h->n_tiles; ++tI)
                for (int ks = 0; ks < ksx; ++ks)
                    for (int l = 0; l < 64; ++l) {
                        const long long i = 16 * tI + (l & 15);
                        const int k = 4 * ks + (l >> 4);
                        if (i < N && k < d) xf[((size_t)tI * ksx + ks) * 64 + l] = data[i * d + k];
                    }
            ALLOC(h->Xf, xf.size());
            HIPCHK(hipMemcpy(h->Xf, xf.data(), sizeof(double) * xf.size(), hipMemcpyHostToDevice));
            if (direct) {
                // whitened, centred observations z_i = L^-1 x~_i (ISO: x~_i), rows padded with zeros to DP scalars
                h->dp_direct = d <= 8 ? 8 : d <= 16 ? 16 : d <= 32 ? 32 : 64;
                const int DP = h->dp_direct;
                dev.assign((size_t)N * DP, 0.0);
                for (long long i = 0; i < N; ++i) {
                    const double* x = data + i * d;
                    double* zr = dev.data() + (size_t)i * DP;
                    if (family == DEMC_FAM_MVN_FULL) {
                        for (int r = 0; r < d; ++r) {
                            double sr = 0.0;
                            for (int k = 0; k <= r; ++k) sr += Linv[(size_t)r * d + k] * x[k];
                            zr[r] = sr;
                        }
                    } else
                        for (int k = 0; k < d; ++k) zr[k] = x[k];
                }
            }
        } break;
        default:
            return fail(h, DEMC_EUNSUPPORTED, "model family is not registered; arbitrary closures cannot run on the device");
    }
    if (!dev.empty()) {
        ALLOC(h->data, dev.size());
        HIPCHK(hipMemcpy(h->data, dev.data(), sizeof(double) * dev.size(), hipMemcpyHostToDevice));
    }
    h->family = family;
    return size_k1_lds(h);
    });
}

static int32_t set_model_source_impl(demc_handle* h, const char* hip_source, const double* data, const int64_t* dims, int32_t ndims,
                                     const double* hyper, int32_t nhyper, bool row, bool has_prior) {
    return guarded(h, [&]() -> int32_t {
    if (!h || !hip_source) return DEMC_EINVAL;
    USE_DEVICE(h);
    if (ndims < 1 || ndims > 8 || !dims || dims[0] < 1 || !data) return fail(h, DEMC_EINVAL, "user model: dims[0] = number of observations, data required");
    long long n_data = 1;
    for (int i = 0; i < ndims; ++i) n_data *= dims[i];
    if (n_data < 1) return fail(h, DEMC_EINVAL, "user model: empty data");
    HIPCHK(hipStreamSynchronize(h->stream));
    for (double** p : {&h->data, &h->Ainv, &h->Ypad, &h->Xf, &h->sx, &h->xbar, &h->user_hyper})
        if (*p) { hipFree(*p); *p = nullptr; }
    if (h->user_dims) { hipFree(h->user_dims); h->user_dims = nullptr; }
    if (h->user_module) { hipModuleUnload(h->user_module); h->user_module = nullptr; h->user_kernel = nullptr; }
    h->family = -1; h->d = 0; h->n_acc = 0; h->dpad = 0; h->n_tiles = 0; h->c0 = h->c1 = h->c2 = 0; h->data2_off = 0;
    h->user_row = row; h->user_has_prior = row && has_prior; h->user_ndims = ndims;
    // compile: kernarg struct + declarations + user source + kernel, for gfx950
    const std::string src = std::string(row && has_prior ? "#define DEMC_USER_HAS_PRIOR 1\n" : "") + kUserStruct +
                            (row ? kUserRowPrologue : kUserPrologue) + hip_source + (row ? kUserRowKernel : kUserKernel);
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "demc_user_model.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
        return fail(h, DEMC_EHIP, "hiprtcCreateProgram failed");
    const char* opts[] = {"--offload-arch=" DEMC_ARCH_STR, "-O3", "-std=c++17"};
    const hiprtcResult cr = hiprtcCompileProgram(prog, 3, opts);
    if (cr != HIPRTC_SUCCESS) {
        size_t ls = 0;
        hiprtcGetProgramLogSize(prog, &ls);
        std::string log(ls, '\0');
        if (ls) hiprtcGetProgramLog(prog, &log[0]);
        hiprtcDestroyProgram(&prog);
        return fail(h, DEMC_EINVAL, std::string("user model does not compile: ") + hiprtcGetErrorString(cr) + "\n" + log);
    }
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    std::vector<char> code(cs);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    HIPCHK(hipModuleLoadData(&h->user_module, code.data()));
    HIPCHK(hipModuleGetFunction(&h->user_kernel, h->user_module, row ? "k_user_row" : "k_user_loglike"));
    h->N = dims[0];
    ALLOC(h->data, (size_t)n_data);
    HIPCHK(hipMemcpy(h->data, data, sizeof(double) * (size_t)n_data, hipMemcpyHostToDevice));
    {
        long long dm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < ndims; ++i) dm[i] = dims[i];
        ALLOC(h->user_dims, (size_t)8);
        HIPCHK(hipMemcpy(h->user_dims, dm, sizeof dm, hipMemcpyHostToDevice));
    }
    h->user_nhyper = nhyper > 0 ? nhyper : 0;
    if (h->user_nhyper) {
        if (!hyper) return fail(h, DEMC_EINVAL, "user model: nhyper > 0 without hyper");
        ALLOC(h->user_hyper, (size_t)h->user_nhyper);
        HIPCHK(hipMemcpy(h->user_hyper, hyper, sizeof(double) * (size_t)h->user_nhyper, hipMemcpyHostToDevice));
    }
    h->family = FAM_USER;
    return size_k1_lds(h);
    });
}

int32_t demc_set_model_source(demc_handle* h, const char* hip_source, const double* data, const int64_t* dims, int32_t ndims,
                              const double* hyper, int32_t nhyper) {
    return set_model_source_impl(h, hip_source, data, dims, ndims, hyper, nhyper, false, false);
}

int32_t demc_set_model_source_row(demc_handle* h, const char* hip_source, const double* data, const int64_t* dims, int32_t ndims,
                                  const double* hyper, int32_t nhyper, int32_t flags) {
    if (flags & ~DEMC_USER_HAS_PRIOR) return h ? fail(h, DEMC_EINVAL, "unknown flag") : DEMC_EINVAL;
    return set_model_source_impl(h, hip_source, data, dims, ndims, hyper, nhyper, true, (flags & DEMC_USER_HAS_PRIOR) != 0);
}

static int upload_dimtab(demc_handle* h) {
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->dimtab, h->h_tab.data(), (size_t)h->c.D * sizeof(DimTab), hipMemcpyHostToDevice));
    // run-length form: consecutive scalars with byte-identical entries share a segment
    DimSeg segs[kMaxDimSeg];
    std::memset(segs, 0, sizeof segs);
    int n = 0;
    for (int j = 0; j < h->c.D; ++j) {
        if (j == 0 || std::memcmp(&h->h_tab[(size_t)j], &h->h_tab[(size_t)j - 1], sizeof(DimTab)) != 0) {
            if (n == kMaxDimSeg) { n = -1; break; }
            segs[n].start = j;
            segs[n].t = h->h_tab[(size_t)j];
            ++n;
        }
    }
    h->n_seg = n > 0 ? n : 0;
    h->seg_plain = 0;
    for (int q = 0; q < kMaxDimSeg; ++q) {
        h->seg_start[q] = q < h->n_seg ? segs[q].start : 0;
        const int kd = segs[q].t.kind;
        if (q < h->n_seg && (kd == PR_FLAT || kd == PR_NORMAL || kd == PR_NORMAL_REF)) h->seg_plain |= 1u << q;
    }
    HIPCHK(hipMemcpy(h->dimseg, segs, sizeof segs, hipMemcpyHostToDevice));
    // the lean resident kernel reads the table in its run-length form and knows no Normal(a, theta[ref]) prior: priors and
    // bounds arrive AFTER demc_set_model in every caller, so its plan is taken again whenever the table changes
    if (h->family >= 0) plan_lean(h);
    return DEMC_OK;
}

int32_t demc_set_priors(demc_handle* h, const int32_t* kind, const double* a, const double* b, const int32_t* ref) {
    return guarded(h, [&]() -> int32_t {
    if (!h || !kind) return DEMC_EINVAL;
    USE_DEVICE(h);
    const size_t D = (size_t)h->c.D;
    for (size_t j = 0; j < D; ++j) {
        if (kind[j] < 0 || kind[j] > DEMC_PRIOR_CAUCHY) return fail(h, DEMC_EUNSUPPORTED, "prior kind not registered");
        if (kind[j] == DEMC_PRIOR_NORMAL_REF && (!ref || ref[j] < 0 || ref[j] >= (int)D))
            return fail(h, DEMC_EINVAL, "prior ref out of range");
    }
    for (size_t j = 0; j < D; ++j) {
        DimTab& t = h->h_tab[j];
        const double aj = a ? a[j] : 0.0, bj = b ? b[j] : 1.0;
        t.kind = kind[j];
        t.ref = ref ? ref[j] : 0;
        t.a = aj;
        // reciprocal scale for the location-scale families: no FP64 division per scalar in the kernels
        const bool recip = kind[j] == PR_NORMAL || kind[j] == PR_HALFCAUCHY || kind[j] == PR_GAMMA ||
                           kind[j] == PR_EXPONENTIAL || kind[j] == PR_LOGNORMAL || kind[j] == PR_CAUCHY;
        t.b = recip ? 1.0 / bj : bj;
        t.c = prior_const(kind[j], aj, bj);
    }
    return upload_dimtab(h);
    });
}

int32_t demc_set_bounds(demc_handle* h, const double* lo, const double* hi) {
    return guarded(h, [&]() -> int32_t {
    if (!h || !lo || !hi) return DEMC_EINVAL;
    USE_DEVICE(h);
    const size_t D = (size_t)h->c.D;
    for (size_t j = 0; j < D; ++j) {
        h->h_tab[j].lo = lo[j];
        h->h_tab[j].hi = hi[j];
    }
    return upload_dimtab(h);
    });
}

int32_t demc_set_blocks(demc_handle* h, const uint8_t* masks, int32_t n_blocks) {
    return guarded(h, [&]() -> int32_t {
    if (!h || n_blocks < 0 || (n_blocks > 0 && !masks)) return DEMC_EINVAL;
    USE_DEVICE(h);
    if (h->masks) { hipFree(h->masks); h->masks = nullptr; }
    h->c.n_blocks = n_blocks;
    if (n_blocks > 0) {
        ALLOC(h->masks, (size_t)n_blocks * h->c.D);
        HIPCHK(hipMemcpy(h->masks, masks, (size_t)n_blocks * h->c.D, hipMemcpyHostToDevice));
    }
    h->mask_runs.assign((size_t)n_blocks, demc_handle::MaskRuns());
    for (int b = 0; b < n_blocks; ++b) {  // run-length form of each mask (kernarg table of the long-row kernel)
        auto& mr = h->mask_runs[(size_t)b];
        const uint8_t* mk = masks + (size_t)b * h->c.D;
        for (int j = 0; j < h->c.D; ++j)
            if (j == 0 || (mk[j] != 0) != (mk[j - 1] != 0)) {
                if (mr.n == kMaxMaskRun) { mr.n = -1; break; }
                mr.start[mr.n] = j;
                if (mk[j]) mr.in |= 1u << mr.n;
                ++mr.n;
            }
        if (mr.n < 0) mr = demc_handle::MaskRuns();
    }
    return plan_snap2(h);
    });
}

int32_t demc_set_state(demc_handle* h, const double* theta, const double* weight, const int64_t* id) {
    return guarded(h, [&]() -> int32_t {
    if (!h || !theta) return DEMC_EINVAL;
    USE_DEVICE(h);
    const size_t P = (size_t)h->P, D = (size_t)h->c.D;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->theta, theta, P * D * sizeof(double), hipMemcpyHostToDevice));
    if (id) HIPCHK(hipMemcpy(h->id, id, P * sizeof(long long), hipMemcpyHostToDevice));
    if (weight)
        HIPCHK(hipMemcpy(h->weight, weight, P * sizeof(double), hipMemcpyHostToDevice));
    else {
        if (h->family < 0) return fail(h, DEMC_EINVAL, "demc_set_model before demc_set_state(weight = NULL)");
        int rc = evaluate_rows(h, h->theta, h->weight);
        if (rc != DEMC_OK) return rc;
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipGetLastError());
    }
    return DEMC_OK;
    });
}

int32_t demc_get_state(demc_handle* h, double* theta, double* weight, int64_t* id) {
    return guarded(h, [&]() -> int32_t {
    if (!h) return DEMC_EINVAL;
    USE_DEVICE(h);
    const size_t P = (size_t)h->P, D = (size_t)h->c.D;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (theta) HIPCHK(hipMemcpy(theta, h->theta, P * D * sizeof(double), hipMemcpyDeviceToHost));
    if (weight) HIPCHK(hipMemcpy(weight, h->weight, P * sizeof(double), hipMemcpyDeviceToHost));
    if (id) HIPCHK(hipMemcpy(id, h->id, P * sizeof(long long), hipMemcpyDeviceToHost));
    return DEMC_OK;
    });
}

int32_t demc_set_history_rows(demc_handle* h, int64_t row0, int64_t nrows, const double* rows) {
    return guarded(h, [&]() -> int32_t {
    if (!h || !rows) return DEMC_EINVAL;
    USE_DEVICE(h);
    if (!h->hist) return fail(h, DEMC_EINVAL, "history is not stored on this handle");
    if (row0 < 0 || nrows < 0 || row0 + nrows > h->c.n_rows) return fail(h, DEMC_EINVAL, "history rows out of range");
    const size_t D = (size_t)h->c.D, ld = (size_t)h->hist_ld, P = (size_t)h->P;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (ld == D) {
        HIPCHK(hipMemcpy(h->hist + (size_t)row0 * P * D, rows, (size_t)nrows * P * D * sizeof(double), hipMemcpyHostToDevice));
        return DEMC_OK;
    }
    // padded cells: the caller's dense rows go through a device staging buffer, a few rows at a time
    const size_t chunk = std::max<size_t>(1, ((size_t)64 << 20) / (P * D * sizeof(double)));
    double* stage = nullptr;
    ALLOC(stage, std::min<size_t>(chunk, (size_t)nrows) * P * D);
    int rc = DEMC_OK;
    for (size_t r = 0; r < (size_t)nrows && rc == DEMC_OK; r += chunk) {
        const size_t n = std::min(chunk, (size_t)nrows - r);
        if (hipMemcpy(stage, rows + r * P * D, n * P * D * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { rc = fail(h, DEMC_EHIP, "history upload"); break; }
        hipLaunchKernelGGL(k_hist_repack, dim3(1024), dim3(256), 0, h->stream, h->hist + ((size_t)row0 + r) * P * ld, stage, (long long)(n * P), (int)D,
                           (int)ld, (int)D);
        if (hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(h, DEMC_EHIP, "history repack");
    }
    hipFree(stage);
    return rc;
    });
}

int32_t demc_get_history(demc_handle* h, int64_t row0, int64_t row1, double* th, uint8_t* acc, double* lp, int64_t* idh) {
    return guarded(h, [&]() -> int32_t {
    if (!h) return DEMC_EINVAL;
    USE_DEVICE(h);
    if (!h->hist) return fail(h, DEMC_EINVAL, "history is not stored on this handle");
    if (row0 < 0 || row1 < row0 || row1 > h->c.n_rows) return fail(h, DEMC_EINVAL, "history rows out of range");
    const size_t P = (size_t)h->P, D = (size_t)h->c.D, n = (size_t)(row1 - row0);
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t ld = (size_t)h->hist_ld;
    if (th && ld == D) HIPCHK(hipMemcpy(th, h->hist + (size_t)row0 * P * D, n * P * D * sizeof(double), hipMemcpyDeviceToHost));
    else if (th && n > 0) {  // padded cells: packed into a dense staging buffer on the device, a few rows at a time
        const size_t chunk = std::max<size_t>(1, ((size_t)64 << 20) / (P * D * sizeof(double)));
        double* stage = nullptr;
        ALLOC(stage, std::min(chunk, n) * P * D);
        int rc = DEMC_OK;
        for (size_t r = 0; r < n && rc == DEMC_OK; r += chunk) {
            const size_t m = std::min(chunk, n - r);
            hipLaunchKernelGGL(k_hist_repack, dim3(1024), dim3(256), 0, h->stream, stage, h->hist + ((size_t)row0 + r) * P * ld, (long long)(m * P), (int)D,
                               (int)D, (int)ld);
            if (hipStreamSynchronize(h->stream) != hipSuccess) { rc = fail(h, DEMC_EHIP, "history repack"); break; }
            if (hipMemcpy(th + r * P * D, stage, m * P * D * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(h, DEMC_EHIP, "history download");
        }
        hipFree(stage);
        if (rc != DEMC_OK) return rc;
    }
    if (acc) HIPCHK(hipMemcpy(acc, h->acc_hist + (size_t)row0 * P, n * P, hipMemcpyDeviceToHost));
    if (lp) HIPCHK(hipMemcpy(lp, h->lp_hist + (size_t)row0 * P, n * P * sizeof(double), hipMemcpyDeviceToHost));
    if (idh) {
        std::vector<int> tmp(n * P);
        HIPCHK(hipMemcpy(tmp.data(), h->id_hist + (size_t)row0 * P, n * P * sizeof(int), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n * P; ++i) idh[i] = tmp[i];
    }
    return DEMC_OK;
    });
}

int32_t demc_export_chains(demc_handle* h, int64_t row0, int64_t row1, int32_t layout, double* out) {
    return guarded(h, [&]() -> int32_t {
    if (!h || !out) return DEMC_EINVAL;
    USE_DEVICE(h);
    if (!h->hist) return fail(h, DEMC_EINVAL, "history is not stored on this handle");
    if (row0 < 0 || row1 < row0 || row1 > h->c.n_rows || (layout != 0 && layout != 1)) return fail(h, DEMC_EINVAL, "bad rows / layout");
    if (h->c.n_groups_total != h->c.n_groups)
        return fail(h, DEMC_EINVAL, "sharded handle: gather demc_get_history from every rank and re-key on the host");
    const long long n = row1 - row0;
    if (n == 0) return DEMC_OK;
    const size_t elems = (size_t)n * (size_t)h->P * (size_t)(h->c.D + 2);
    double* dev = nullptr;
    ALLOC(dev, elems);
    KParams k = base_params(h);
    const long long blocks = (long long)((elems + 255) / 256);
    hipLaunchKernelGGL(k_export_chains, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, h->stream, k, (long long)row0, n,
                       (int)layout, (long long)h->c.group_offset * h->c.Np, dev);
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out, dev, elems * sizeof(double), hipMemcpyDeviceToHost);
    hipFree(dev);
    if (e != hipSuccess) return fail(h, DEMC_EHIP, std::string("demc_export_chains: ") + hipGetErrorString(e));
    return DEMC_OK;
    });
}

// the alpha coin of iteration `iter` as this handle sees it: the replayed uniform while one is set (main.jl:85)
static bool migration_due_h(demc_handle* h, int64_t iter) {
    if (h->rp_active && h->rp_has_step) {
        const int ngt = h->c.n_groups_total > 0 ? h->c.n_groups_total : h->c.n_groups;
        return h->rp_u_step <= (ngt == 1 ? 0.0 : h->c.alpha);
    }
    return demc_migration_due(&h->c, iter) != 0;
}

#define NCCLCHK(expr)                                                                                         \
    do {                                                                                                      \
        ncclResult_t r_ = (expr);                                                                             \
        if (r_ != ncclSuccess) return fail(h, DEMC_ERCCL, std::string(#expr) + ": " + ncclGetErrorString(r_)); \
    } while (0)

// this shard's rows inside the [n_groups_total][D+3] exchange buffer: rank r's block starts at r * n_groups rows, which is
// where an IN-PLACE ncclAllGather expects the send buffer (recvbuff + rank * sendcount)
static double* own_rows(demc_handle* h) { return h->mig_rows + (size_t)h->c.group_offset * ((size_t)h->c.D + 3); }

// the ONE collective of the path: all-gather of every shard's candidate rows (migration.jl:11-19 split around it), enqueued
// on stream s.  No host synchronisation.
static int gather_enqueue(demc_handle* h, hipStream_t s) {
    const size_t cnt = (size_t)h->c.n_groups * ((size_t)h->c.D + 3);
    NCCLCHK(ncclAllGather(own_rows(h), h->mig_rows, cnt, ncclDouble, h->comm, s));
    h->n_exchanges += 1;
    return DEMC_OK;
}

// migration! of iteration `iter` on a handle that owns a communicator: pack -> all-gather -> apply, stream-ordered
static int exchange_enqueue(demc_handle* h, int64_t iter) {
    migration_enqueue(h, iter, own_rows(h), nullptr, true, false);
    int rc = gather_enqueue(h, h->stream);
    if (rc != DEMC_OK) return rc;
    migration_enqueue(h, iter, nullptr, h->mig_rows, false, true);
    return DEMC_OK;
}

static int step_body(demc_handle* h, int64_t iter0, int32_t n_iters, bool with_migration);

// update! + store_samples! of iterations [iter0, iter0 + n_iters) for a SUBSET of the handle's groups (local indices),
// enqueued only.  The list travels through a ring of pinned host / device copies: the device copy is refilled by a copy ON
// THE HANDLE'S STREAM, i.e. behind every kernel that still reads its previous content.
static int update_subset(demc_handle* h, int64_t iter0, int32_t n_iters, const int32_t* groups, int32_t n) {
    if (n == 0) return DEMC_OK;
    if (h->rp_active && h->rp_n_mig > 0)
        return fail(h, DEMC_EINVAL, "subset updates follow demc_migration_groups, which does not see a replayed migration sub-group");
    const int G = h->c.n_groups;
    for (int i = 0; i < n; ++i)
        if (groups[i] < 0 || groups[i] >= G) return fail(h, DEMC_EINVAL, "group index outside this handle");
    const int slot = h->glist_next;
    h->glist_next = (slot + 1) % demc_handle::kGlistRing;
    int* dev = h->glist_buf[slot];
    if (hipEventSynchronize(h->glist_ev[slot]) != hipSuccess) return fail(h, DEMC_EHIP, "group-list ring");
    std::memcpy(h->glist_pin[slot], groups, (size_t)n * sizeof(int));
    if (hipMemcpyAsync(dev, h->glist_pin[slot], (size_t)n * sizeof(int), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
        hipEventRecord(h->glist_ev[slot], h->stream) != hipSuccess)
        return fail(h, DEMC_EHIP, "copying the group list");
    h->cur_glist = dev;
    h->cur_ng = n;
    const int rc = step_body(h, iter0, n_iters, false);
    h->cur_glist = nullptr;
    h->cur_ng = 0;
    return rc;
}

// Per-group-asynchronous migration behind the boundary: migration of iteration `iter` + the update of
// iterations [iter, iter + run).  The sub-group of an exchange is a pure function of (seed, iter), so the groups it did NOT
// select start their update while the all-gather is in flight on the side stream; only the selected groups wait for it.
static int exchange_overlapped(demc_handle* h, int64_t iter, int run) {
    const demc_config& c = h->c;
    std::vector<int32_t> sel((size_t)c.n_groups_total), mine, rest;
    int32_t n_sel = 0;
    demc_migration_groups(&c, iter, sel.data(), &n_sel);
    std::vector<char> picked((size_t)c.n_groups, 0);
    for (int i = 0; i < n_sel; ++i) {
        const int gl = sel[(size_t)i] - c.group_offset;
        if (gl >= 0 && gl < c.n_groups) picked[(size_t)gl] = 1;
    }
    for (int g = 0; g < c.n_groups; ++g) (picked[(size_t)g] ? mine : rest).push_back(g);
    migration_enqueue(h, iter, own_rows(h), nullptr, true, false);
    HIPCHK(hipEventRecord(h->ev_pack, h->stream));
    HIPCHK(hipStreamWaitEvent(h->side, h->ev_pack, 0));
    int rc = gather_enqueue(h, h->side);  // the one collective, off the main stream
    if (rc != DEMC_OK) return rc;
    HIPCHK(hipEventRecord(h->ev_gath, h->side));
    rc = update_subset(h, iter, run, rest.data(), (int32_t)rest.size());  // overlaps the gather
    if (rc != DEMC_OK) return rc;
    HIPCHK(hipStreamWaitEvent(h->stream, h->ev_gath, 0));
    migration_enqueue(h, iter, nullptr, h->mig_rows, false, true);
    return update_subset(h, iter, run, mine.data(), (int32_t)mine.size());
}

// iterations [iter0, iter0 + n_iters) enqueued on the handle's stream; nothing is drained
// launch order of the groups for the frozen-row sweeps of iterations [iter0, iter0 + n_iters): mutating groups first (the
// group's coin is addressed Philox -- the host draws what the kernel will draw)
static void plan_frozen_order(demc_handle* h, int64_t iter0, int32_t n_iters) {
    const demc_config& c = h->c;
    h->frozen_iters = 0;
    const bool hier = h->family == FAM_HIER_BINOMIAL || h->family == FAM_HIER_GAUSSIAN;
    if (!hier || c.n_blocks < 1 || h->lpp <= 64 || h->cur_glist || c.beta <= 0.0 || h->rp_active || n_iters < 1) return;
    // nothing to plan where launch_phase cannot take the frozen form at all: too few moving particles for two workgroups per CU,
    // recombination, a pool beyond the kernel's, the unfused / per-phase forms, a trace
    const int n_act = c.schedule == DEMC_SCHED_TWO_COLOUR ? c.Np - c.Np / 2 : c.schedule == DEMC_SCHED_SEQUENTIAL ? 1 : c.Np;
    if ((long long)h->geo_groups * n_act < 2LL * h->n_cus || c.kappa != 1.0 || c.Np > 512 || c.fuse != 0 || c.trace || !h->hier_scr) return;
    bool any = false;
    std::vector<char> frozen((size_t)c.n_blocks, 0);
    for (int b = 0; b < c.n_blocks && (size_t)b < h->mask_runs.size(); ++b) {
        const auto& mr = h->mask_runs[(size_t)b];
        int inside = 0;
        for (int r = 0; r < mr.n; ++r)
            if ((mr.in >> r) & 1u) inside += (r + 1 < mr.n ? mr.start[r + 1] : c.D) - mr.start[r];
        frozen[(size_t)b] = mr.n > 0 && inside >= 1;
        any = any || frozen[(size_t)b];
    }
    const size_t need = (size_t)n_iters * c.n_blocks * c.n_groups;
    if (!any || need > ((size_t)1 << 24)) return;
    if (need > h->frozen_order_cap) {
        if (h->frozen_order_d) { if (hipStreamSynchronize(h->stream) != hipSuccess) return; hipFree(h->frozen_order_d); }
        h->frozen_order_d = nullptr; h->frozen_order_cap = 0;
        if (hipMalloc((void**)&h->frozen_order_d, need * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); return; }
        h->frozen_order_cap = need;
    }
    // PINNED staging owned by the handle, guarded by an event: the copy below is asynchronous for the host too (a pageable source
    // made it wait for everything already queued on the stream -- step_body's "nothing is drained" was not true for blocked
    // hierarchical models), and the buffer outlives the call
    if (!h->frozen_order_ev && hipEventCreateWithFlags(&h->frozen_order_ev, hipEventDisableTiming) != hipSuccess) { h->frozen_order_ev = nullptr; return; }
    if (h->frozen_order_pin && hipEventSynchronize(h->frozen_order_ev) != hipSuccess) return;  // (the previous table has left the buffer)
    if (need > h->frozen_order_pin_cap) {
        if (h->frozen_order_pin) hipHostFree(h->frozen_order_pin);
        h->frozen_order_pin = nullptr; h->frozen_order_pin_cap = 0;
        if (hipHostMalloc((void**)&h->frozen_order_pin, need * sizeof(int), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return; }
        h->frozen_order_pin_cap = need;
    }
    int* order = h->frozen_order_pin;
    std::vector<int> rest;
    rest.reserve((size_t)c.n_groups);
    for (int32_t t = 0; t < n_iters; ++t)
        for (int b = 0; b < c.n_blocks; ++b) {
            int* o = order + ((size_t)t * c.n_blocks + b) * c.n_groups;
            int n_front = 0;
            rest.clear();
            for (int g = 0; g < c.n_groups; ++g) {
                bool mut = false;
                if (frozen[(size_t)b]) {
                    const U4 r = draw_block(c.seed, S_GROUP, (uint32_t)b, (uint64_t)(iter0 + t), (uint32_t)(c.group_offset + g), 0u);
                    mut = u53(r.x, r.y) <= c.beta;
                }
                if (mut) o[n_front++] = g; else rest.push_back(g);
            }
            for (size_t i = 0; i < rest.size(); ++i) o[n_front + (int)i] = rest[i];
        }
    // ordered on the handle's stream behind every kernel that still reads the previous call's table
    if (hipMemcpyAsync(h->frozen_order_d, order, need * sizeof(int), hipMemcpyHostToDevice, h->stream) != hipSuccess) return;
    if (hipEventRecord(h->frozen_order_ev, h->stream) != hipSuccess) return;
    h->frozen_iter0 = iter0; h->frozen_iters = n_iters;
}

static int step_body(demc_handle* h, int64_t iter0, int32_t n_iters, bool with_migration) {
    const demc_config& c = h->c;
    const int n_sweeps = c.n_blocks > 0 ? c.n_blocks : 1;  // block_update! main.jl:174-179
    plan_frozen_order(h, iter0, n_iters);
    h->snap2_iter = -1;  // (a by-product snapshot lives inside one call)
    for (int64_t iter = iter0; iter < iter0 + n_iters; ++iter) {
        if (with_migration && migration_due_h(h, iter)) {  // main.jl:85
            if (c.n_groups_total != c.n_groups || (h->comm && h->own_comm)) {  // (a communicator of one rank takes the same path)
                if (h->multi) return fail(h, DEMC_EINVAL, "shard of a multi-GPU set: step the set with demc_multi_step");
                if (!h->comm)
                    return fail(h, DEMC_EINVAL, "sharded handle without a communicator: demc_comm_init, or drive the exchange with "
                                                "demc_migration_pack/apply + demc_update");
                // (history partners: the cells of row t - 1 come from EVERY group, so no subset of the groups may run ahead of the
                // others by an iteration -- the overlapped form, which lets the unselected groups do exactly that, is not taken)
                if (h->comm_overlap && !h->rp_active && c.partner_kind != DEMC_PARTNER_HISTORY) {
                    int run = 1;
                    while (iter + run < iter0 + n_iters && !migration_due_h(h, iter + run)) ++run;
                    int rc = exchange_overlapped(h, iter, run);
                    if (rc != DEMC_OK) return rc;
                    iter += run - 1;
                    continue;
                }
                int rc = exchange_enqueue(h, iter);
                if (rc != DEMC_OK) return rc;
            } else
                migration_enqueue(h, iter, h->mig_rows, h->mig_rows, true, true);
        }
        const bool st_ok = h->st_ok && !h->cur_glist;  // (a subset update never uses the form whose workgroups wait on each other)
        // the default sampler has lean kernels of its own on MvNormal-full and on the per-observation families (same draws, same decisions)
        KParams kp0 = base_params(h);
        kp0.mode = MODE_STEP;
        const bool plain = is_plain(h, kp0);
        const bool obs_lean = h->lean_obs_ok && plain;
        const bool dir_lean = h->lean_direct_ok && plain && !h->cur_glist;  // (its workgroups wait on each other: never on a subset)
        if ((h->res_ok || st_ok || obs_lean || dir_lean) && !h->rp_active) {  // every iteration up to the next migration in one launch
            int run = 1;  // capped so that a single launch stays in the millisecond range whatever the caller asks for
            const int cap = (st_ok || dir_lean) ? 64 : 1024;
            while (run < cap && iter + run < iter0 + n_iters && !(with_migration && migration_due_h(h, iter + run))) ++run;
            int rc;
            if (dir_lean) rc = launch_lean(h, iter, run, true);
            else if (st_ok && plain && h->lean_stream_ok) rc = launch_lean(h, iter, run, true);
            else if (!st_ok && plain && h->lean_ok) rc = launch_lean(h, iter, run, false);
            else if (obs_lean) rc = launch_lean_obs(h, iter, run);
            else rc = st_ok ? launch_stream(h, iter, run) : launch_resident(h, iter, run);
            if (rc != DEMC_OK) return rc;
            iter += run - 1;
            continue;
        }
        if (h->lean_hist_ok && !h->rp_active) {  // DE-MC_Z, the default sampler: the lean body
            KParams kp = base_params(h);
            kp.mode = MODE_STEP;
            const int lh = lean_hist(h, kp);  // 1: the default sampler; 2 with theta_snooker > 0 and no blocks: + snooker updates
            if (lh == 1 || (lh == 2 && c.n_blocks == 0)) {
                int rc = launch_lean_hist(h, iter, lh == 2);
                if (rc != DEMC_OK) return rc;
                continue;
            }
        }
        for (int b = 0; b < n_sweeps; ++b) {
            const unsigned char* mask = c.n_blocks > 0 ? h->masks + (size_t)b * c.D : nullptr;
            const long long row = iter - 1;
            const long long store_row = (b == n_sweeps - 1 && h->hist && row < c.n_rows) ? row : -1;
            int rc = run_sweep(h, iter, (unsigned)b, mask, store_row);
            if (rc != DEMC_OK) return rc;
        }
    }
    HIPCHK(hipGetLastError());
    return DEMC_OK;
}

static int step_checks(demc_handle* h, int64_t iter0, int32_t n_iters) {
    if (h->family < 0) return fail(h, DEMC_EINVAL, "demc_set_model has not been called");
    if (iter0 < 1 || n_iters < 0) return fail(h, DEMC_EINVAL, "iter0 is 1-based (de.iter, main.jl:34)");
    if (h->c.partner_kind == DEMC_PARTNER_HISTORY && iter0 < 2)
        return fail(h, DEMC_EINVAL, "history partners need at least one stored row (n_initial > 0)");
    // resample draws cells of rows 1:(iter-1) (crossover.jl:116-118): every one of them must lie inside the history buffer
    if (h->c.partner_kind == DEMC_PARTNER_HISTORY && n_iters > 0 && iter0 + (int64_t)n_iters - 2 > h->c.n_rows)
        return fail(h, DEMC_EINVAL, "history partners: iteration t reads history rows 1:(t-1), the history holds n_rows rows "
                                    "(the reference sizes it n_iter + n_initial, utilities.jl:34)");
    return DEMC_OK;
}

static int drain(demc_handle* h) {
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipGetLastError());
    if (h->st_err && *h->st_err) {
        *h->st_err = 0u;
        return fail(h, DEMC_EHIP, "streaming-resident kernel: a hand-over between the workgroups of a group timed out");
    }
    if (h->comm) {  // a collective that failed after it was enqueued shows up here
        ncclResult_t ar = ncclSuccess;
        if (ncclCommGetAsyncError(h->comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress)
            return fail(h, DEMC_ERCCL, std::string("communicator: ") + ncclGetErrorString(ar));
    }
    return DEMC_OK;
}

static int32_t step_impl(demc_handle* h, int64_t iter0, int32_t n_iters, bool with_migration, bool do_drain = true) {
    return guarded(h, [&]() -> int32_t {
    if (!h) return DEMC_EINVAL;
    USE_DEVICE(h);
    int rc = step_checks(h, iter0, n_iters);
    if (rc != DEMC_OK) return rc;
    rc = step_body(h, iter0, n_iters, with_migration);
    if (rc != DEMC_OK || !do_drain) return rc;
    return drain(h);
    });
}

int32_t demc_step(demc_handle* h, int64_t iter0, int32_t n_iters) { return step_impl(h, iter0, n_iters, true); }
int32_t demc_update(demc_handle* h, int64_t iter0, int32_t n_iters) { return step_impl(h, iter0, n_iters, false); }
int32_t demc_step_async(demc_handle* h, int64_t iter0, int32_t n_iters) { return step_impl(h, iter0, n_iters, true, false); }
int32_t demc_synchronize(demc_handle* h) {
    return guarded(h, [&]() -> int32_t {
    if (!h) return DEMC_EINVAL;
    USE_DEVICE(h);
    return drain(h);
    });
}

int32_t demc_update_groups_async(demc_handle* h, int64_t iter0, int32_t n_iters, const int32_t* groups, int32_t n) {
    return guarded(h, [&]() -> int32_t {
    if (!h || n < 0 || (n > 0 && !groups)) return DEMC_EINVAL;
    USE_DEVICE(h);
    int rc = step_checks(h, iter0, n_iters);
    if (rc != DEMC_OK) return rc;
    if (h->c.partner_kind == DEMC_PARTNER_HISTORY && n_iters > 1)
        return fail(h, DEMC_EINVAL, "history partners: a subset of the groups advances one iteration at a time (the cells of a history "
                                    "row come from every group: update the other groups before the next iteration)");
    return update_subset(h, iter0, n_iters, groups, n);
    });
}

int32_t demc_migration_groups(const demc_config* cfg, int64_t iter, int32_t* sel, int32_t* n_sel) {
    if (!cfg || !sel || !n_sel) return DEMC_EINVAL;
    // select_groups (migration.jl:31-35) exactly as k_mig_apply draws it: N = rand(2:n_groups), ordered sample without
    // replacement by a partial Fisher-Yates shuffle, words from the STEP stream
    const int ng = cfg->n_groups_total > 0 ? cfg->n_groups_total : cfg->n_groups;
    *n_sel = 0;
    if (ng < 2) return DEMC_OK;
    try {
        std::vector<int> perm((size_t)ng);
        for (int i = 0; i < ng; ++i) perm[(size_t)i] = i;
        const U4 r0 = draw_block(cfg->seed, S_STEP, 0, (uint64_t)iter, 0, 0);
        const int ns = 2 + (int)mulhi32(r0.z, (uint32_t)(ng - 1));
        U4 r = r0;
        for (int i = 0; i < ns; ++i) {
            if ((i & 3) == 0) r = draw_block(cfg->seed, S_STEP, 0, (uint64_t)iter, 0, 1 + (uint32_t)(i >> 2));
            const uint32_t w = (i & 3) == 0 ? r.x : (i & 3) == 1 ? r.y : (i & 3) == 2 ? r.z : r.w;
            const int j = i + (int)mulhi32(w, (uint32_t)(ng - i));
            std::swap(perm[(size_t)i], perm[(size_t)j]);
        }
        for (int i = 0; i < ns; ++i) sel[i] = perm[(size_t)i];
        *n_sel = ns;
    } catch (...) {
        return DEMC_ENOMEM;
    }
    return DEMC_OK;
}

int32_t demc_migration_due(const demc_config* cfg, int64_t iter) {
    if (!cfg) return 0;
    const int ngt = cfg->n_groups_total > 0 ? cfg->n_groups_total : cfg->n_groups;
    const double alpha = ngt == 1 ? 0.0 : cfg->alpha;
    const U4 r = draw_block(cfg->seed, S_STEP, 0, (uint64_t)iter, 0, 0);
    return u53(r.x, r.y) <= alpha ? 1 : 0;
}

int32_t demc_migration_pack(demc_handle* h, int64_t iter, double* dev_rows) {
    return guarded(h, [&]() -> int32_t {
    if (!h) return DEMC_EINVAL;
    USE_DEVICE(h);
    migration_enqueue(h, iter, dev_rows ? dev_rows : h->mig_rows, nullptr, true, false);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipGetLastError());
    return DEMC_OK;
    });
}

int32_t demc_migration_apply(demc_handle* h, int64_t iter, const double* dev_all_rows) {
    return guarded(h, [&]() -> int32_t {
    if (!h) return DEMC_EINVAL;
    USE_DEVICE(h);
    if (!dev_all_rows && h->c.n_groups_total != h->c.n_groups)
        return fail(h, DEMC_EINVAL, "sharded handle needs the all-gathered rows");
    migration_enqueue(h, iter, nullptr, dev_all_rows ? dev_all_rows : h->mig_rows, false, true);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipGetLastError());
    return DEMC_OK;
    });
}

int32_t demc_migration_pack_async(demc_handle* h, int64_t iter, double* dev_rows) {
    return guarded(h, [&]() -> int32_t {
    if (!h || !dev_rows) return DEMC_EINVAL;
    USE_DEVICE(h);
    migration_enqueue(h, iter, dev_rows, nullptr, true, false);
    HIPCHK(hipGetLastError());
    return DEMC_OK;
    });
}

int32_t demc_migration_apply_async(demc_handle* h, int64_t iter, const double* dev_all_rows) {
    return guarded(h, [&]() -> int32_t {
    if (!h || !dev_all_rows) return DEMC_EINVAL;
    USE_DEVICE(h);
    migration_enqueue(h, iter, nullptr, dev_all_rows, false, true);
    HIPCHK(hipGetLastError());
    return DEMC_OK;
    });
}

// ---- the communicator behind the boundary (SURVEY 8b "demc_create_multi / DEMC_ERCCL", 8e) ----
int32_t demc_comm_unique_id(void* id_out, int32_t nbytes) {
    return guarded(nullptr, [&]() -> int32_t {
    if (!id_out || nbytes < (int32_t)sizeof(ncclUniqueId)) return DEMC_EINVAL;
    static_assert(sizeof(ncclUniqueId) == DEMC_COMM_ID_BYTES, "DEMC_COMM_ID_BYTES must be sizeof(ncclUniqueId)");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return DEMC_ERCCL;
    std::memcpy(id_out, &id, sizeof id);
    return DEMC_OK;
    });
}

static int comm_side_objects(demc_handle* h) {
    if (!h->side) HIPCHK(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
    if (!h->ev_pack) HIPCHK(hipEventCreateWithFlags(&h->ev_pack, hipEventDisableTiming));
    if (!h->ev_gath) HIPCHK(hipEventCreateWithFlags(&h->ev_gath, hipEventDisableTiming));
    return DEMC_OK;
}

static int comm_shape_check(demc_handle* h, int rank, int world) {
    const demc_config& c = h->c;
    if (world < 1 || rank < 0 || rank >= world) return fail(h, DEMC_EINVAL, "need 0 <= rank < world");
    // equal shards in rank order: what the in-place all-gather (and select_groups over GLOBAL group indices) assume
    if ((long long)c.n_groups * world != c.n_groups_total || c.group_offset != rank * c.n_groups)
        return fail(h, DEMC_EINVAL, "communicator shape: n_groups_total must be world * n_groups and group_offset rank * n_groups");
    return DEMC_OK;
}

int32_t demc_comm_init(demc_handle* h, const void* unique_id, int32_t rank, int32_t world) {
    return guarded(h, [&]() -> int32_t {
    if (!h || !unique_id) return DEMC_EINVAL;
    USE_DEVICE(h);
    if (h->comm || h->multi) return fail(h, DEMC_EINVAL, "the handle already has a communicator");
    int rc = comm_shape_check(h, rank, world);
    if (rc != DEMC_OK) return rc;
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    NCCLCHK(ncclCommInitRank(&h->comm, world, id, rank));
    h->own_comm = true; h->comm_rank = rank; h->comm_world = world;
    return comm_side_objects(h);
    });
}

int32_t demc_comm_destroy(demc_handle* h) {
    return guarded(h, [&]() -> int32_t {
    if (!h) return DEMC_EINVAL;
    if (h->multi) return fail(h, DEMC_EINVAL, "the communicator of a shard belongs to its set (demc_destroy_multi)");
    USE_DEVICE(h);
    if (h->stream) HIPCHK(hipStreamSynchronize(h->stream));
    if (h->side) HIPCHK(hipStreamSynchronize(h->side));
    if (h->comm && h->own_comm) NCCLCHK(ncclCommDestroy(h->comm));
    h->comm = nullptr; h->own_comm = false; h->comm_rank = 0; h->comm_world = 1; h->comm_overlap = false;
    return DEMC_OK;
    });
}

int32_t demc_comm_set_overlap(demc_handle* h, int32_t on) {
    return guarded(h, [&]() -> int32_t {
    if (!h) return DEMC_EINVAL;
    if (on && (!h->comm || h->multi)) return fail(h, DEMC_EINVAL, "needs a communicator of the handle's own (demc_comm_init)");
    h->comm_overlap = on != 0;
    return DEMC_OK;
    });
}

static int32_t exchange_impl(demc_handle* h, int64_t iter, bool do_drain) {
    return guarded(h, [&]() -> int32_t {
    if (!h) return DEMC_EINVAL;
    USE_DEVICE(h);
    if (!h->comm) return fail(h, DEMC_EINVAL, "the handle has no communicator (demc_comm_init)");
    int rc = exchange_enqueue(h, iter);
    if (rc != DEMC_OK || !do_drain) return rc;
    return drain(h);
    });
}
int32_t demc_migration_exchange(demc_handle* h, int64_t iter) { return exchange_impl(h, iter, true); }
int32_t demc_migration_exchange_async(demc_handle* h, int64_t iter) { return exchange_impl(h, iter, false); }

int32_t demc_comm_allreduce(demc_handle* h, double* host_inout, int32_t n, int32_t op) {
    return guarded(h, [&]() -> int32_t {
    if (!h || n < 0 || (n > 0 && !host_inout) || op < 0 || op > 2) return DEMC_EINVAL;
    USE_DEVICE(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    if (!h->comm || h->comm_world == 1) return DEMC_OK;  // one rank: the values are the result
    const size_t m = n > 0 ? (size_t)n : 1;              // n == 0: a barrier (one dummy element goes round)
    if (h->red_cap < m) {
        if (h->red_dev) { hipFree(h->red_dev); h->red_dev = nullptr; h->red_cap = 0; }
        ALLOC(h->red_dev, m);
        h->red_cap = m;
    }
    double dummy = 0.0;
    HIPCHK(hipMemcpyAsync(h->red_dev, n > 0 ? host_inout : &dummy, m * sizeof(double), hipMemcpyHostToDevice, h->stream));
    const ncclRedOp_t rop = op == 0 ? ncclSum : op == 1 ? ncclMax : ncclMin;
    NCCLCHK(ncclAllReduce(h->red_dev, h->red_dev, m, ncclDouble, rop, h->comm, h->stream));
    HIPCHK(hipMemcpyAsync(n > 0 ? host_inout : &dummy, h->red_dev, m * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    return drain(h);
    });
}

int32_t demc_comm_stats(demc_handle* h, int64_t* out3) {
    if (!h || !out3) return DEMC_EINVAL;
    out3[0] = h->comm ? h->comm_world : 1;
    out3[1] = h->comm ? h->comm_rank : 0;
    out3[2] = h->n_exchanges;
    return DEMC_OK;
}

// ---- single-process multi-GPU set: one host thread (a Julia task) drives every shard ----
}  // extern "C"  (the struct below is C++)

struct demc_multi {
    std::vector<demc_handle*> shard;
    bool rccl = false;              // distinct devices: one communicator per shard from ncclCommInitAll
    std::vector<hipEvent_t> packed, copied;  // shared-device transport: rows packed / peers' rows copied
    bool copied_valid = false;
    std::string err;
};

namespace {
int mfail(demc_multi* m, int code, const std::string& msg) noexcept {
    if (m) {
        try { m->err = msg; } catch (...) {}
    }
    return code;
}
// first failing shard's message becomes the set's
int mshard_fail(demc_multi* m, int r, int code) {
    return mfail(m, code, "shard " + std::to_string(r) + ": " + m->shard[(size_t)r]->err);
}
}  // namespace

extern "C" {

const char* demc_multi_last_error(demc_multi* m) { return m ? m->err.c_str() : "null set"; }

int32_t demc_destroy_multi(demc_multi* m) {
    return guarded(nullptr, [&]() -> int32_t {
    if (!m) return DEMC_OK;
    for (demc_handle* h : m->shard) {
        if (!h) continue;
        hipSetDevice(h->c.device_id);
        if (h->stream) hipStreamSynchronize(h->stream);
        if (h->comm) { ncclCommDestroy(h->comm); h->comm = nullptr; }
        h->multi = nullptr;
    }
    for (size_t r = m->shard.size(); r-- > 0;) {  // (last first: shards that share a device borrow an earlier shard's stream)
        if (m->shard[r]) hipSetDevice(m->shard[r]->c.device_id);
        if (r < m->packed.size() && m->packed[r]) hipEventDestroy(m->packed[r]);
        if (r < m->copied.size() && m->copied[r]) hipEventDestroy(m->copied[r]);
        if (m->shard[r]) demc_destroy(m->shard[r]);
    }
    delete m;
    return DEMC_OK;
    });
}

int32_t demc_create_multi(const demc_config* cfg, int32_t n_shards, const int32_t* device_ids, demc_multi** out) {
    if (!cfg || !out || n_shards < 1) return DEMC_EINVAL;
    *out = nullptr;
    demc_multi* m = new (std::nothrow) demc_multi();
    if (!m) return DEMC_ENOMEM;
    *out = m;  // returned even on failure so that demc_multi_last_error() can be read; the caller destroys it
    return guarded(nullptr, [&]() -> int32_t {
    if (cfg->n_groups % n_shards != 0) return mfail(m, DEMC_EINVAL, "n_groups (of the whole population) must divide by n_shards");
    if (cfg->group_offset != 0 || (cfg->n_groups_total != 0 && cfg->n_groups_total != cfg->n_groups))
        return mfail(m, DEMC_EINVAL, "demc_create_multi takes the configuration of the WHOLE population (group_offset 0)");
    // DE-MC_Z (resample, crossover.jl:113-124) draws its partner cells from the history of ALL particles; a shard holds the
    // history of its own groups only, so a sharded set would sample another pool than the single handle it promises to
    // reproduce bit for bit: refused (one shard is fine).  The one-process-per-GPU road documents the shard-local pool instead
    // (include/demc.h).
    if (cfg->partner_kind == DEMC_PARTNER_HISTORY && n_shards > 1)
        return mfail(m, DEMC_EUNSUPPORTED, "demc_create_multi: history partners (resample) draw from the history of all particles; "
                                           "a sharded set cannot reproduce that pool -- use one shard, or partner_kind current");
    const int G = cfg->n_groups / n_shards;
    std::vector<int> devs((size_t)n_shards);
    bool distinct = true;
    for (int r = 0; r < n_shards; ++r) {
        devs[(size_t)r] = device_ids ? device_ids[r] : r;
        for (int q = 0; q < r; ++q) distinct = distinct && devs[(size_t)q] != devs[(size_t)r];
    }
    m->shard.assign((size_t)n_shards, nullptr);
    for (int r = 0; r < n_shards; ++r) {
        demc_config c = *cfg;
        c.n_groups = G; c.group_offset = r * G; c.n_groups_total = cfg->n_groups; c.device_id = devs[(size_t)r];
        // one population on several GPUs: every shard takes the lane geometry of the unsharded run, so that the set
        // reproduces a single handle bit for bit (demc_config.geometry_groups)
        if (c.geometry_groups == 0) c.geometry_groups = cfg->n_groups;
        const int rc = demc_create(&c, &m->shard[(size_t)r]);
        if (rc != DEMC_OK) return mfail(m, rc, "shard " + std::to_string(r) + ": " + (m->shard[(size_t)r] ? m->shard[(size_t)r]->err : "demc_create"));
        m->shard[(size_t)r]->multi = m;
    }
    struct Seal { demc_multi* m; ~Seal() { for (demc_handle* h : m->shard) if (h) h->multi_sealed = true; } } seal{m};
    if (n_shards > 1 && distinct) {
        std::vector<ncclComm_t> comms((size_t)n_shards);
        const ncclResult_t nr = ncclCommInitAll(comms.data(), n_shards, devs.data());
        if (nr != ncclSuccess) return mfail(m, DEMC_ERCCL, std::string("ncclCommInitAll: ") + ncclGetErrorString(nr));
        for (int r = 0; r < n_shards; ++r) {
            demc_handle* h = m->shard[(size_t)r];
            h->comm = comms[(size_t)r]; h->own_comm = false; h->comm_rank = r; h->comm_world = n_shards;
        }
        m->rccl = true;
    } else if (n_shards > 1) {
        // shards that share a device (several shards per GPU): the rows change hands by device-to-device copies ordered by
        // events -- RCCL refuses two ranks on one device
        m->packed.assign((size_t)n_shards, nullptr);
        m->copied.assign((size_t)n_shards, nullptr);
        // Shards on ONE device run on one stream (the first such shard's): the streaming-resident kernels size their grid
        // for the whole chip and their workgroups spin on each other, so two of them in flight at once on a device would not
        // be co-resident (a 2 x 40-group set: 2 x 160 workgroups on 256 CUs).  In stream order the set is what one handle
        // of all groups does, group range by group range.
        for (int r = 1; r < n_shards; ++r)
            for (int q = 0; q < r; ++q)
                if (devs[(size_t)q] == devs[(size_t)r]) {
                    const int rc = demc_set_stream(m->shard[(size_t)r], (void*)m->shard[(size_t)q]->stream);
                    if (rc != DEMC_OK) return mshard_fail(m, r, rc);
                    break;
                }
        for (int r = 0; r < n_shards; ++r) {
            if (hipSetDevice(devs[(size_t)r]) != hipSuccess ||
                hipEventCreateWithFlags(&m->packed[(size_t)r], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&m->copied[(size_t)r], hipEventDisableTiming) != hipSuccess)
                return mfail(m, DEMC_EHIP, "creating the exchange events");
        }
    }
    return DEMC_OK;
    });
}

int32_t demc_multi_size(demc_multi* m) { return m ? (int32_t)m->shard.size() : 0; }
demc_handle* demc_multi_shard(demc_multi* m, int32_t r) {
    return (m && r >= 0 && (size_t)r < m->shard.size()) ? m->shard[(size_t)r] : nullptr;
}

// migration! of iteration `iter` over the whole set, enqueued on every shard's stream
static int multi_exchange(demc_multi* m, int64_t iter) {
    const int n = (int)m->shard.size();
    for (int r = 0; r < n; ++r) {
        demc_handle* h = m->shard[(size_t)r];
        if (hipSetDevice(h->c.device_id) != hipSuccess) return mfail(m, DEMC_EHIP, "hipSetDevice");
        if (!m->rccl && m->copied_valid)  // nobody may still be copying the rows this shard is about to overwrite
            for (int q = 0; q < n; ++q)
                if (q != r && hipStreamWaitEvent(h->stream, m->copied[(size_t)q], 0) != hipSuccess) return mfail(m, DEMC_EHIP, "hipStreamWaitEvent");
        migration_enqueue(h, iter, own_rows(h), nullptr, true, false);
        if (!m->rccl && hipEventRecord(m->packed[(size_t)r], h->stream) != hipSuccess) return mfail(m, DEMC_EHIP, "hipEventRecord");
    }
    if (m->rccl) {
        // one host thread, several devices: the collective calls of all ranks go out as one group
        ncclResult_t nr = ncclGroupStart();
        for (int r = 0; r < n && nr == ncclSuccess; ++r) {
            demc_handle* h = m->shard[(size_t)r];
            const size_t cnt = (size_t)h->c.n_groups * ((size_t)h->c.D + 3);
            nr = ncclAllGather(own_rows(h), h->mig_rows, cnt, ncclDouble, h->comm, h->stream);
            h->n_exchanges += 1;
        }
        const ncclResult_t ne = ncclGroupEnd();
        if (nr == ncclSuccess) nr = ne;
        if (nr != ncclSuccess) return mfail(m, DEMC_ERCCL, std::string("ncclAllGather: ") + ncclGetErrorString(nr));
    } else {
        for (int r = 0; r < n; ++r) {
            demc_handle* h = m->shard[(size_t)r];
            if (hipSetDevice(h->c.device_id) != hipSuccess) return mfail(m, DEMC_EHIP, "hipSetDevice");
            const size_t W = (size_t)h->c.D + 3, bytes = (size_t)h->c.n_groups * W * sizeof(double);
            for (int q = 0; q < n; ++q) {
                if (q == r) continue;
                demc_handle* src = m->shard[(size_t)q];
                if (hipStreamWaitEvent(h->stream, m->packed[(size_t)q], 0) != hipSuccess ||
                    hipMemcpyAsync(h->mig_rows + (size_t)src->c.group_offset * W, own_rows(src), bytes, hipMemcpyDeviceToDevice, h->stream) != hipSuccess)
                    return mfail(m, DEMC_EHIP, "copying a peer's migration rows");
            }
            if (hipEventRecord(m->copied[(size_t)r], h->stream) != hipSuccess) return mfail(m, DEMC_EHIP, "hipEventRecord");
            h->n_exchanges += 1;
        }
        m->copied_valid = true;
    }
    for (int r = 0; r < n; ++r) {
        demc_handle* h = m->shard[(size_t)r];
        if (hipSetDevice(h->c.device_id) != hipSuccess) return mfail(m, DEMC_EHIP, "hipSetDevice");
        migration_enqueue(h, iter, nullptr, h->mig_rows, false, true);
    }
    return DEMC_OK;
}

int32_t demc_multi_step(demc_multi* m, int64_t iter0, int32_t n_iters) {
    return guarded(nullptr, [&]() -> int32_t {
    if (!m || m->shard.empty()) return DEMC_EINVAL;
    const int n = (int)m->shard.size();
    for (int r = 0; r < n; ++r) {
        const int rc = step_checks(m->shard[(size_t)r], iter0, n_iters);
        if (rc != DEMC_OK) return mshard_fail(m, r, rc);
    }
    demc_handle* h0 = m->shard[0];
    for (int64_t it = iter0; it < iter0 + n_iters;) {
        // the alpha coin is a pure function of (seed, iteration): every shard sees the same runs (main.jl:85)
        const bool due = n > 1 && migration_due_h(h0, it);
        int run = 1;
        while (it + run < iter0 + n_iters && !(n > 1 && migration_due_h(h0, it + run))) ++run;
        if (due) {
            const int rc = multi_exchange(m, it);
            if (rc != DEMC_OK) return rc;
        }
        for (int r = 0; r < n; ++r) {  // every shard's update enqueued before any is waited for: the GPUs run side by side
            demc_handle* h = m->shard[(size_t)r];
            if (hipSetDevice(h->c.device_id) != hipSuccess) return mfail(m, DEMC_EHIP, "hipSetDevice");
            // (a single shard keeps its on-device migration inside the update)
            const int rc = step_body(h, it, run, n == 1);
            if (rc != DEMC_OK) return mshard_fail(m, r, rc);
        }
        it += run;
    }
    for (int r = 0; r < n; ++r) {
        demc_handle* h = m->shard[(size_t)r];
        if (hipSetDevice(h->c.device_id) != hipSuccess) return mfail(m, DEMC_EHIP, "hipSetDevice");
        const int rc = drain(h);
        if (rc != DEMC_OK) return mshard_fail(m, r, rc);
    }
    return DEMC_OK;
    });
}

int32_t demc_apply_migration(demc_handle* h, const int32_t* src_slot, const int32_t* dst_slot, int32_t n) {
    return guarded(h, [&]() -> int32_t {
    if (!h || n < 0 || (n > 0 && (!src_slot || !dst_slot))) return DEMC_EINVAL;
    USE_DEVICE(h);
    if (n == 0) return DEMC_OK;
    std::vector<char> seen((size_t)h->P, 0);
    for (int k = 0; k < n; ++k) {
        if (src_slot[k] < 0 || src_slot[k] >= h->P || dst_slot[k] < 0 || dst_slot[k] >= h->P)
            return fail(h, DEMC_EINVAL, "migration slot out of range");
        if (seen[(size_t)dst_slot[k]]) return fail(h, DEMC_EINVAL, "migration destination slots must be distinct");
        seen[(size_t)dst_slot[k]] = 1;
    }
    const size_t W = (size_t)h->c.D + 2;
    int* d_slots = nullptr;
    double* d_stage = nullptr;
    HIPCHK(hipMalloc(&d_slots, 2 * (size_t)n * sizeof(int)));
    if (hipMalloc(&d_stage, (size_t)n * W * sizeof(double)) != hipSuccess) {
        hipFree(d_slots);
        return fail(h, DEMC_ENOMEM, "out of device memory for the migration staging rows");
    }
    int rc = DEMC_OK;
    hipError_t e = hipMemcpyAsync(d_slots, src_slot, (size_t)n * sizeof(int), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_slots + n, dst_slot, (size_t)n * sizeof(int), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        KParams k = base_params(h);
        const unsigned wgs = (unsigned)(((size_t)n * W + 255) / 256);
        hipLaunchKernelGGL(k_slot_moves, dim3(wgs), dim3(256), 0, h->stream, k, d_slots, d_stage, n, 0);
        hipLaunchKernelGGL(k_slot_moves, dim3(wgs), dim3(256), 0, h->stream, k, d_slots + n, d_stage, n, 1);
        e = hipStreamSynchronize(h->stream);
        if (e == hipSuccess) e = hipGetLastError();
    }
    if (e != hipSuccess) rc = fail(h, DEMC_EHIP, hipGetErrorString(e));
    hipFree(d_slots);
    hipFree(d_stage);
    return rc;
    });
}

int32_t demc_get_weights(demc_handle* h, double* weight) {
    if (!h || !weight) return DEMC_EINVAL;
    return demc_get_state(h, nullptr, weight, nullptr);
}

int32_t demc_logpost(demc_handle* h, const double* theta, int64_t n, double* out) {
    return guarded(h, [&]() -> int32_t {
    if (!h || !theta || !out || n < 0) return DEMC_EINVAL;
    USE_DEVICE(h);
    if (h->family < 0) return fail(h, DEMC_EINVAL, "demc_set_model has not been called");
    const size_t P = (size_t)h->P, D = (size_t)h->c.D;
    std::vector<double> w(P);
    for (int64_t off = 0; off < n; off += (int64_t)P) {
        const size_t m = (size_t)((n - off < (int64_t)P) ? n - off : (int64_t)P);
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemset(h->scratch_theta, 0, P * D * sizeof(double)));
        HIPCHK(hipMemcpy(h->scratch_theta, theta + (size_t)off * D, m * D * sizeof(double), hipMemcpyHostToDevice));
        int rc = evaluate_rows(h, h->scratch_theta, h->scratch_w);
        if (rc != DEMC_OK) return rc;
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy(w.data(), h->scratch_w, P * sizeof(double), hipMemcpyDeviceToHost));
        std::memcpy(out + off, w.data(), m * sizeof(double));
    }
    return DEMC_OK;
    });
}

int32_t demc_set_replay(demc_handle* h, const demc_replay* r) {
    return guarded(h, [&]() -> int32_t {
    if (!h) return DEMC_EINVAL;
    USE_DEVICE(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    free_replay(h);
    if (!r) return DEMC_OK;
    const demc_config& c = h->c;
    const size_t P = (size_t)h->P, D = (size_t)c.D, G = (size_t)c.n_groups;
    if (r->partner)
        for (size_t i = 0; i < 3 * P; ++i)
            if (r->partner[i] >= c.Np) return fail(h, DEMC_EINVAL, "replay: partner row outside the group");
    if (r->mig_particle)
        for (size_t g = 0; g < G; ++g)
            if (r->mig_particle[g] >= c.Np) return fail(h, DEMC_EINVAL, "replay: migration particle outside the group");
    if (r->n_mig_groups < 0 || r->n_mig_groups > c.n_groups_total || (r->n_mig_groups > 0 && !r->mig_groups))
        return fail(h, DEMC_EINVAL, "replay: bad migration sub-group");
    {
        std::vector<char> seen((size_t)c.n_groups_total, 0);
        for (int i = 0; i < r->n_mig_groups; ++i) {
            const int g = r->mig_groups[i];
            if (g < 0 || g >= c.n_groups_total || seen[(size_t)g]) return fail(h, DEMC_EINVAL, "replay: migration groups must be distinct and in range");
            seen[(size_t)g] = 1;
        }
    }
    auto up = [&](auto** dst, const auto* src, size_t n) -> int {
        if (!src || n == 0) return DEMC_OK;
        int rc = dev_alloc(h, dst, n);
        if (rc != DEMC_OK) return rc;
        HIPCHK(hipMemcpy(*dst, src, n * sizeof(**dst), hipMemcpyHostToDevice));
        return DEMC_OK;
    };
    int rc = DEMC_OK;
    if (rc == DEMC_OK) rc = up(&h->rp_group, r->u_group, G);
    if (rc == DEMC_OK) rc = up(&h->rp_part, r->u_part, 5 * P);
    if (rc == DEMC_OK) rc = up(&h->rp_partner, reinterpret_cast<const long long*>(r->partner), 3 * P);
    if (rc == DEMC_OK) rc = up(&h->rp_noise, r->u_noise, P * D);
    if (rc == DEMC_OK) rc = up(&h->rp_znoise, r->z_noise, P * D);
    if (rc == DEMC_OK) rc = up(&h->rp_recomb, r->u_recomb, P * D);
    if (rc == DEMC_OK) rc = up(&h->rp_mig_particle, reinterpret_cast<const long long*>(r->mig_particle), G);
    if (rc == DEMC_OK) rc = up(&h->rp_mig_groups, r->mig_groups, (size_t)r->n_mig_groups);
    if (rc != DEMC_OK) {
        free_replay(h);
        return rc;
    }
    h->rp_n_mig = r->n_mig_groups;
    if (r->u_step && r->u_step[0] == r->u_step[0]) {
        h->rp_has_step = true;
        h->rp_u_step = r->u_step[0];
    }
    h->rp_active = true;
    return DEMC_OK;
    });
}

int32_t demc_get_trace(demc_handle* h, double* proposal, double* w_prop, double* log_adj, int32_t* idx, uint8_t* accepted) {
    return guarded(h, [&]() -> int32_t {
    if (!h) return DEMC_EINVAL;
    USE_DEVICE(h);
    const size_t P = (size_t)h->P, D = (size_t)h->c.D;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (proposal) HIPCHK(hipMemcpy(proposal, h->prop, P * D * sizeof(double), hipMemcpyDeviceToHost));
    if (w_prop) HIPCHK(hipMemcpy(w_prop, h->tr_w, P * sizeof(double), hipMemcpyDeviceToHost));
    if (log_adj) HIPCHK(hipMemcpy(log_adj, h->prop_adj, P * sizeof(double), hipMemcpyDeviceToHost));
    if (idx) HIPCHK(hipMemcpy(idx, h->tr_idx, P * 4 * sizeof(int), hipMemcpyDeviceToHost));
    if (accepted) HIPCHK(hipMemcpy(accepted, h->tr_acc, P, hipMemcpyDeviceToHost));
    return DEMC_OK;
    });
}

int32_t demc_last_kernels(demc_handle* h, char* out, int32_t nbytes) {
    return guarded(h, [&]() -> int32_t {
    if (!h || !out || nbytes < 1) return DEMC_EINVAL;
    const demc_handle::LastPlan& L = h->last;
    static const char* const tails[4] = {"TAIL_NONE", "TAIL_PREP", "TAIL_PREP_MFMA", "TAIL_OBS"};
    char buf[256];
    buf[0] = '\0';
    const char* tf[3] = {"false", "true", "2"};  // (the LEAN level: 0 general, 1 plain, 2 plain + snooker)
    switch (L.k1) {
        case 0:
            std::snprintf(buf, sizeof buf, "k_propose<%d,%s,%s,false,%s>", L.wg, tf[L.tile != 0], tails[L.tail & 3], tf[L.plain]);
            break;
        case 1: std::snprintf(buf, sizeof buf, "k_longrow<%d>", L.wg); break;
        case 5: std::snprintf(buf, sizeof buf, "k_frozen_sweep<%d%s>", L.wg, L.big ? ",big" : ""); break;
        case 6: std::snprintf(buf, sizeof buf, "k_res_obs<%d>", L.wg); break;
        case 2:
            std::snprintf(buf, sizeof buf, "k_propose<%d,true,%s,true,%s>", L.wg, tails[L.tail & 3], tf[L.plain]);
            break;
        case 3:
            std::snprintf(buf, sizeof buf, "k_propose<%d,true,%s,true,%s,true>", L.wg, tails[L.tail & 3], tf[L.plain]);
            break;
        case 4:
            if (L.hist && L.iso) std::snprintf(buf, sizeof buf, "k_res_mvn<%d,%s,%d,%d,iso>", L.wg, tf[L.stream != 0], L.dt, L.hist);
            else if (L.hist) std::snprintf(buf, sizeof buf, "k_res_mvn<%d,%s,%d,%d>", L.wg, tf[L.stream != 0], L.dt, L.hist);
            else if (L.big) std::snprintf(buf, sizeof buf, "k_res_mvn<%d,%s,%d,direct>", L.wg, tf[L.stream != 0], L.dt);
            else std::snprintf(buf, sizeof buf, "k_res_mvn<%d,%s,%d>", L.wg, tf[L.stream != 0], L.dt);
            break;
        default: break;
    }
    std::string s = buf;
    if (L.k2 == 1) s += " + k_cross_mfma<" + std::to_string(L.ks) + ",4>";
    else if (L.k2 == 2) s += " + k_obs_loglike";
    else if (L.k2 == 8) s += " + k_lba_wave";
    else if (L.k2 == 3) s += " + k_hier_loglike";
    else if (L.k2 == 4) s += " + k_user_loglike";
    else if (L.k2 == 5) s += " + k_direct_mvn<" + std::to_string(L.ks) + ">";
    else if (L.k2 == 6) s += " + k_user_row";
    else if (L.k2 == 7) s += " + k_lba_loglike<512>";
    if (L.k3) s += " + k_accept_store";
    std::snprintf(out, (size_t)nbytes, "%s", s.c_str());
    return DEMC_OK;
    });
}

int32_t demc_timing_enable(demc_handle* h, int32_t on) {
    return guarded(h, [&]() -> int32_t {
    if (!h) return DEMC_EINVAL;
    USE_DEVICE(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    drain_events(h);
    h->timing = on != 0;
    return DEMC_OK;
    });
}

int32_t demc_timing_read(demc_handle* h, double* out10, int32_t reset) {
    return guarded(h, [&]() -> int32_t {
    if (!h || !out10) return DEMC_EINVAL;
    USE_DEVICE(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    drain_events(h);
    for (int i = 0; i < 5; ++i) {
        out10[i] = h->t_ms[i];
        out10[5 + i] = (double)h->t_n[i];
    }
    if (reset)
        for (int i = 0; i < 5; ++i) { h->t_ms[i] = 0; h->t_n[i] = 0; }
    return DEMC_OK;
    });
}

int32_t demc_timing_clock(demc_handle* h, double* out4) {
    return guarded(h, [&]() -> int32_t {
    if (!h || !out4) return DEMC_EINVAL;
    USE_DEVICE(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    out4[0] = out4[1] = out4[2] = out4[3] = 0.0;
    if (!h->clk_dev || h->clk_n == 0) return DEMC_OK;  // no DIRECT likelihood launch ran with timing enabled
    std::vector<unsigned long long> t(3 * h->clk_n);
    HIPCHK(hipMemcpy(t.data(), h->clk_dev, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    // per CU (the shader-clock counters of different CUs are not aligned: differencing across CUs gave 1.8 - 13.7 "GHz"): the
    // workgroup that finished first there and the one that finished last, by the 100 MHz reference; the shader-clock ticks between
    // the two over the reference ticks between the two is the clock the CU held over that span of the launch.  A CU whose span is
    // under 20 us (a launch of a single round of workgroups) says nothing.
    std::vector<size_t> order;
    order.reserve(h->clk_n);
    for (size_t i = 0; i < h->clk_n; ++i)
        if (t[3 * i + 1] != 0) order.push_back(i);  // (a workgroup that reported)
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return t[3 * a + 2] < t[3 * b + 2]; });
    std::vector<double> mhz;
    for (size_t a = 0; a < order.size();) {
        size_t b = a, lo = order[a], hi = order[a];
        for (; b < order.size() && t[3 * order[b] + 2] == t[3 * order[a] + 2]; ++b) {
            const size_t i = order[b];
            if (t[3 * i + 1] < t[3 * lo + 1]) lo = i;
            if (t[3 * i + 1] > t[3 * hi + 1]) hi = i;
        }
        if (t[3 * lo + 1] != 0 && t[3 * hi + 1] - t[3 * lo + 1] >= 2000 && t[3 * hi] > t[3 * lo])
            mhz.push_back(100.0 * (double)(t[3 * hi] - t[3 * lo]) / (double)(t[3 * hi + 1] - t[3 * lo + 1]));
        a = b;
    }
    if (mhz.empty()) return DEMC_OK;
    std::sort(mhz.begin(), mhz.end());
    out4[0] = mhz[mhz.size() / 2];
    out4[1] = mhz.front();
    out4[2] = mhz.back();
    out4[3] = (double)mhz.size();
    return DEMC_OK;
    });
}

}  // extern "C"
