// demc_kernels.hpp -- hand-written gfx950 kernels of the DE-MCMC hot path.
//
//   K1  k_propose<WG,TILE,TAIL,RES,PLAIN>  crossover!/snooker_update!/mutation!/recombination!/reset!/in_bounds + prior
//                        (crossover.jl:30-99,154-257,301-352; mutation.jl:13-25; utilities.jl:70-78)
//       (fused tails)    MvNormal preparation y = A^-1 (theta' - xbar), a = (theta' - xbar).y; optionally the whole
//                        accept/store when the likelihood needs no separate pass over the data (SUFFSTAT, small N,
//                        hierarchical families; two_colour) -- then also as a RESIDENT kernel that runs every
//                        iteration up to the next migration with the group held in LDS
//   K2  k_cross_mfma     S_p = sum_i y_p . x~_i over all observations on v_mfma_f64_16x16x4_f64
//       k_obs_loglike    thread-per-proposal streaming likelihoods (Gaussian, Binomial, LBA, LNR, rastrigin)
//       k_hier_loglike   workgroup-per-proposal likelihoods whose cost is O(D) (hierarchical families, unfused form)
//   K3  k_accept_store   compute_posterior! finalisation + mh_update!/maximize!/minimize! + store_samples!
//                        (utilities.jl:92-99,161-180,201-226)
//   M   k_mig_pack / k_mig_apply   select_particle / select_groups + shift_particles! (migration.jl:31-91)
//
// Wave = 64 lanes everywhere.  A particle is handled by a power-of-two sub-group of LPP lanes
// (LPP = lanes per particle, chosen on the host from D and the population size) so that rows of theta are read and written
// as contiguous segments; every lane owns dims {2k, 2k+1} for k = sl, sl+LPP, ... which is also
// the granularity of one Philox block (two 53-bit uniforms).
#pragma once
#include <type_traits>
#include "demc_device.hpp"

namespace demc {

enum Family : int {
    FAM_GAUSSIAN = 0, FAM_MVN_ISO = 1, FAM_MVN_FULL = 2, FAM_BINOMIAL = 3, FAM_HIER_BINOMIAL = 4,
    FAM_HIER_GAUSSIAN = 5, FAM_LBA = 6, FAM_LNR = 7, FAM_RASTRIGIN = 8, FAM_USER = 100
};
enum Mode : int { MODE_STEP = 0, MODE_IDENT = 1 };  // IDENT: theta' = theta, always accepted (init / logpost)

// Per-scalar constants, packed so that one dimension costs three 16-byte loads: bounds (de.bounds flattened) and
// the prior entry (model.prior_loglike as data).  b = 1/scale for Normal / half-Cauchy; c = the x-independent
// part of the log-density (prior_const()).
struct DimTab {
    double lo, hi;
    double a, b;
    double c;
    int kind, ref;
};

// Run-length form of the DimTab array: models give whole blocks of scalars the same bounds and prior (cfg4: 10 000 subject
// effects ~ Normal(0, sd); cfg3: 32 means ~ Normal(0,1)), so the table is a handful of segments.  The kernels keep them in LDS
// and look a scalar's entry up by its segment instead of streaming 48 bytes per scalar from L2/HBM -- at D = 10 002 that
// stream was larger than the particle rows themselves.
struct DimSeg {
    int start, pad;  // first scalar of the segment
    DimTab t;
    double pad2;
};
constexpr int kMaxDimSeg = 16;
constexpr int kMaxMaskRun = 16;

// Everything a sweep needs, passed by value as the kernarg.
struct KParams {
    // geometry
    int n_groups, Np, D, group_offset;
    int a_lo, n_act;                  // active particles of each group in this phase: [a_lo, a_lo+n_act)
    int pool_lo, pool_n, exclude_self;  // partner pool of each group
    int lpp;                          // lanes per particle in K1 (256 = one workgroup per particle, very large D)
    int lpp3;                         // lanes per particle in K3
    int mode;
    // sampler
    long long iter, burnin;
    unsigned sweep;
    unsigned long long seed;
    double beta, eps, sigma, kappa, theta_snooker;
    int proposal_kind, partner_kind, update_kind, fitness_kind;
    // state
    double* theta;        // [P][D]
    double* weight;       // [P]
    long long* id;        // [P]
    double* prop;         // [P][D] proposals, by slot
    double* prop_prior;   // [P]
    double* prop_adj;     // [P]
    unsigned char* prop_oob;  // [P]
    int* tr_idx;          // [P][4]
    double* tr_w;         // [P]
    unsigned char* tr_acc;  // [P]
    const DimTab* dimtab;       // [D] bounds + prior of every scalar, packed
    const DimSeg* dimseg;       // [n_seg] the same table run-length encoded (n_seg = 0: too many segments, use dimtab)
    int n_seg;
    int ainv_lds;               // K1: A^-1 [d][d] is staged in LDS (else read from L2: wide data)
    const int* glist;           // update of a SUBSET of the handle's groups (demc_update_groups_async): launch group i is local
                                // group glist[i], n_groups counts the subset; null: all groups, in order
    const unsigned char* mask;  // [D] or null
    // history (slot keyed)
    double* hist;             // [rows][P][hist_ld]: a cell holds the D scalars of a row, cells hist_ld >= D doubles apart
    unsigned char* acc_hist;  // [rows][P]
    double* lp_hist;          // [rows][P]
    int* id_hist;             // [rows][P]
    int hist_ld;              // doubles between consecutive history cells (D, or D padded to whole cache lines: demc_create)
    long long P;              // local particles
    long long store_row;      // >= 0: K3 stores this history row
    int tile_in_lds;          // K1: stage the group tile in LDS
    int n_split;              // K1: workgroups per group
    // K1 resident form (one workgroup keeps its group in LDS over both colour phases of several iterations)
    int n_iters, n_sweeps;    // iterations to run from `iter`, and block sweeps per iteration; mask = all block masks [n_sweeps][D]
    long long n_rows;         // history rows allocated (store while iter-1 < n_rows)
    int fuse_prep;            // K1 computes y = A^-1 theta', a = theta'.y (MvNormal families)
    int prep_mfma;            // ... on the matrix cores (full Sigma, d <= 32, 16 lanes per particle)
    int tile_rows;            // K1: rows of the LDS tile = the partner pool (+ the workgroup's own rows if outside it)
    int own_in_pool;          // K1: the moving particles are themselves pool rows (synchronous schedule)
    int plan;                 // K1: per-particle scalars (coins, indices, gammas) computed once per workgroup, 4 lanes each
    int scr_doubles;          // K1: size of the theta' scratch region in LDS (doubles)
    int fuse_obs;             // K1 sums the per-observation terms itself (small N, scalar-data families)
    int fuse_accept;          // K1 finishes the update (cheap likelihoods, two_colour)
    int write_prop;           // K1 writes proposals to HBM (needed by K2/K3 or by the trace)
    int trace;                // keep the per-slot diagnostic trace
    const double* Ainv;       // [d][d] or null (ISO)
    const double* sx;         // [d] sum_i (x_i - xbar), non-null in SUFFSTAT mode
    const double* xbar;       // [d] data mean: the MvNormal families work with theta' - xbar
    double* Ypad;             // [P][dpad]
    int dpad;
    // model
    int family;
    long long N;              // observations (or subjects)
    int d;                    // data dimension
    int n_acc;                // accumulators (LBA/LNR)
    int n_partials;
    double* partial;          // [n_partials][P]
    double* aux;              // [P] per-proposal scalar (quadratic term)
    const double* data;       // family specific
    const double* data2;
    double c0, c1, c2;        // family constants
    int direct;               // MvNormal families, DIRECT likelihood: Ypad holds the whitened proposal m = L^-1 (theta' - xbar),
                              // data the whitened observations z_i, and the summed statistic is sum_i |z_i - m|^2
    // replay (demc_set_replay, test mode): device copies of the caller's draws, null = addressed Philox as usual
    const double* rp_group;         // [n_groups]
    const double* rp_part;          // [P][5] snooker coin, select_base, gamma_1, gamma_2, accept
    const long long* rp_partner;    // [P][3] rows inside the group
    const double* rp_noise;         // [P][D]
    const double* rp_znoise;        // [P][D]
    const double* rp_recomb;        // [P][D]
    const int* rp_mig_groups;       // [rp_n_mig] global groups
    const long long* rp_mig_particle;  // [n_groups]
    int rp_n_mig;
    // streaming-resident form (STREAM instances of k_propose): the MvNormal observation stream inside the resident kernel
    int st_C;                       // workgroups per group: each streams 1/C of the observation tiles for ALL moving particles
    int st_nact_max;                // moving particles of the larger colour phase
    int st_rows;                    // rows of the theta' scratch (st_nact_max rounded up to whole passes)
    int st_x_lds;                   // this workgroup's chunk of Xf is held in LDS for the whole launch
    int st_chunk_tiles;             // observation tiles per workgroup chunk
    int n_tiles;                    // observation tiles of 16 in Xf
    const double* Xf;               // [n_tiles+1][dpad/4][64] fragment-ordered data (demc_set_model)
    unsigned long long* st_gran;    // [2][n_groups][st_C][st_nact_max][2] hand-over granules {epoch:32 | half of a double:32}
    unsigned* st_err;               // set to 1 if a hand-over timed out (never, with co-resident workgroups)
    // DE-MC_Z inside burn-in on long rows (k_longrow, synchronous schedule): random_gamma's base particle (crossover.jl:156-164)
    // is a row of the CURRENT population, which another workgroup of the same launch may be writing -- so the base rows and the
    // weights select_base reads come from a SNAPSHOT of the sweep's start (null: the live arrays)
    const double* base_theta;       // [P][D]
    const double* base_weight;      // [P]
    // ... and a frozen sweep over the whole population, which streams every row anyway, leaves the rows and weights it ENDS with
    // in these (null: it does not): the next sweep's snapshot without a copy (k_frozen_sweep, launch_phase)
    double* snap_theta;             // [P][D]
    double* snap_weight;            // [P]
    // run-length tables IN the kernarg (scalar loads, no memory behind them) for wave-uniform look-ups (demc_longrow.hpp):
    // first scalars of the table segments; first scalars of the runs of the sweep's block mask, bit r of mrun_in = run r
    // lies inside the block (no blocks: one run, inside); n_mrun = 0: more runs than kMaxMaskRun.  Bit q of seg_plain:
    // segment q has a flat, Normal or Normal(a, theta[ref]) prior.
    int seg_start[kMaxDimSeg];
    int mrun_start[kMaxMaskRun];
    int n_mrun;
    unsigned mrun_in, seg_plain;
};
// a replayed uniform replaces the drawn one unless it is NaN
__device__ inline double replayed(const double* tab, size_t i, double drawn) {
    if (!tab) return drawn;
    const double v = tab[i];
    return v == v ? v : drawn;
}

typedef double d4 __attribute__((ext_vector_type(4)));  // C/D fragment of v_mfma_f64_16x16x4_f64

// q-th active particle of the phase -> (group, particle-in-group, local slot); 32-bit on purpose (P < 2^31)
__device__ inline int slot_of(const KParams& p, int q, int& g, int& pl) {
    g = q / p.n_act;
    pl = p.a_lo + (q - g * p.n_act);
    if (p.glist) g = p.glist[g];
    return g * p.Np + pl;
}
__device__ inline int slot_of(const KParams& p, int q) {
    int g, pl;
    return slot_of(p, q, g, pl);
}

// ---- cross-lane data movement inside a sub-group, on the DPP path (VALU speed) instead of ds_bpermute (LDS round trip).
// hip's __shfl* always lower to ds_bpermute_b32 (~100 cycles of latency each); a sub-group of up to 16 lanes lives inside one
// DPP row, where quad_perm / row_half_mirror / row_mirror give an all-reduce in four VALU-rate steps.
template <int CTRL>
__device__ inline int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ inline double dpp_mov(double v) {
    return __hiloint2double(dpp_i32<CTRL>(__double2hiint(v)), dpp_i32<CTRL>(__double2loint(v)));
}
template <int CTRL>
__device__ inline int dpp_mov(int v) {
    return dpp_i32<CTRL>(v);
}
constexpr int kDppXor1 = 0xB1;         // quad_perm:[1,0,3,2]
constexpr int kDppXor2 = 0x4E;         // quad_perm:[2,3,0,1]
constexpr int kDppHalfMirror = 0x141;  // row_half_mirror: lane i <-> 7-i inside each 8
constexpr int kDppMirror = 0x140;      // row_mirror: lane i <-> 15-i inside each 16

// all-reduce over the lpp lanes of a sub-group (lpp = power of two <= 64, sub-groups aligned)
template <typename T>
__device__ inline T subgroup_sum(T v, int lpp) {
    if (lpp >= 2) v += dpp_mov<kDppXor1>(v);
    if (lpp >= 4) v += dpp_mov<kDppXor2>(v);
    if (lpp >= 8) v += dpp_mov<kDppHalfMirror>(v);
    if (lpp >= 16) v += dpp_mov<kDppMirror>(v);
    if (lpp >= 32) v += __shfl_xor(v, 16);
    if (lpp >= 64) v += __shfl_xor(v, 32);
    return v;
}
// A lane holds four values v[0..3], one for each of four ROWS, and every row is to be summed over the 16 lanes of the lane's
// group (the layout of a v_mfma_f64_16x16x4_f64 result: rows (lane >> 4) + 4 r, column lane & 15).  Folded as a transposition
// instead of four separate 16-lane sums: lanes exchange the half of their rows they give up (xor 1: rows {0,1} <-> {2,3};
// xor 2: one row each), then the four quads of the group add up (row_ror 4, 8) -- 5 exchanges and 5 additions instead of 16
// and 16.  Every lane returns the sum of row fold4_row(lane); a fixed tree (the same bits wherever it runs).
__device__ inline int fold4_row(int lane) { return 2 * (lane & 1) + ((lane >> 1) & 1); }
__device__ inline double fold4_over16(double v0, double v1, double v2, double v3, int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    const double k0 = b0 ? v2 : v0, k1 = b0 ? v3 : v1, s0 = b0 ? v0 : v2, s1 = b0 ? v1 : v3;
    const double r0 = k0 + dpp_mov<kDppXor1>(s0), r1 = k1 + dpp_mov<kDppXor1>(s1);
    double u = (b1 ? r1 : r0) + dpp_mov<kDppXor2>(b1 ? r0 : r1);
    u += dpp_mov<0x124>(u);  // row_ror:4
    u += dpp_mov<0x128>(u);  // row_ror:8
    return u;
}
// lpp == blockDim (256 or 512): one particle spans the whole workgroup (very large D); the sum crosses the waves through
// LDS and is formed as a fixed tree over pairs of waves.  Must be called by every thread of the workgroup (the particle --
// hence the control flow -- is workgroup-uniform then).
template <typename T>
__device__ inline T group_sum(T v, int lpp, T* s8) {
    if (lpp <= 64) return subgroup_sum(v, lpp);
    v = subgroup_sum(v, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s8[threadIdx.x >> 6] = v;
    __syncthreads();
    const T lo = (s8[0] + s8[1]) + (s8[2] + s8[3]);
    return blockDim.x > 256 ? lo + ((s8[4] + s8[5]) + (s8[6] + s8[7])) : lo;
}
// value held by lane B of the sub-group (B < lpp): row_newbcast inside a 16-lane row, quad_perm inside a quad
template <int B>
__device__ inline uint32_t subgroup_bcast(uint32_t v, int lpp, int sub_base) {
    if (lpp == 16) return (uint32_t)dpp_i32<0x150 + B>((int)v);
    if (lpp == 4 && B < 4) return (uint32_t)dpp_i32<(B & 3) * 0x55>((int)v);  // quad_perm:[B,B,B,B]
    return (uint32_t)__shfl((int)v, sub_base + B);
}

// ------------------------------------------------------------------------------------------------
// Log-likelihood from the accumulated statistics of one proposal (shared by the fused K1 tail and K3).
//   s   = sum of the partial sums (MvNormal: cross term S = sum_i y.x_i; Gaussian: sum z^2; else the log-likelihood)
//   aux = quadratic term mu.y (MvNormal);  sg = the proposal's sigma where the family has one
// ------------------------------------------------------------------------------------------------
__device__ inline double loglike_from_stats(const KParams& p, double s, double aux, double sg) {
    switch (p.family) {
        case FAM_MVN_FULL:  // c0 = -N/2 (d log2pi + logdet), c1 = sum_i x_i' A^-1 x_i
            if (p.direct) return p.c0 - 0.5 * s;  // s = sum_i |L^-1 (x_i - mu)|^2, summed term by term
            return p.c0 - 0.5 * (p.c1 - 2.0 * s + (double)p.N * aux);
        case FAM_MVN_ISO: {  // c1 = sum_i |x_i|^2
            const double nd = (double)p.N * (double)p.d;
            if (p.direct) return -0.5 * nd * kLog2Pi - nd * log(sg) - 0.5 * s / (sg * sg);
            return -0.5 * nd * kLog2Pi - nd * log(sg) - 0.5 * (p.c1 - 2.0 * s + (double)p.N * aux) / (sg * sg);
        }
        case FAM_GAUSSIAN:
            return -0.5 * (s + (double)p.N * kLog2Pi) - (double)p.N * log(sg);
        default:
            return s;
    }
}

// LBA log-likelihood of the trials i0, i0+stride, ... < i1 for one proposal (thread per proposal: every lane of the wave is
// at the same trial, so choice and decision time arrive by scalar loads).  Trials go in batches of kLbaBatch: their (choice,
// rt) pairs are loaded together -- scalar loads share the wait counter with the LDS table reads, so a load per trial
// made every trial wait for both -- and the batch contributes ONE log, of the product of its floored densities (each
// >= 1e-10 and of order one or below for any plausible proposal; a product that nevertheless leaves the double range is
// redone with a log per trial; a -Inf trial contributes a factor 0 and log 0 = -Inf).
constexpr int kLbaBatch = 8;
template <int NA, int RS = kPhiRow>
__device__ __forceinline__ double lba_range_sum(const KParams& p, const double* th, long long i0, long long i1, int stride, const double* tab) {
    const int na = NA > 0 ? NA : p.n_acc;
    double nu[8], nuS[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        nu[a] = a < na ? th[a] : 0.0;
        nuS[a] = kPhiS * nu[a];
    }
    const double A = th[na], kk = th[na + 1], tau = th[na + 2], b = A + kk, inv_A = 1.0 / A;
    const double kS = kPhiS * kk, bS = kPhiS * b, inv_SA = inv_A * (1.0 / kPhiS);
    double pneg = 1.0;
#pragma unroll
    for (int a = 0; a < (NA > 0 ? NA : 8); ++a)
        if (a < na) {
            double q, Ph;
            phiS_Phi_table<RS>(tab, -nuS[a], q, Ph);
            pneg *= Ph;
        }
    const double inv_norm = 1.0 / (1.0 - pneg);
    double acc = 0.0;
    long long i = i0;
    for (; i + (long long)(kLbaBatch - 1) * stride < i1; i += (long long)kLbaBatch * stride) {
        double cc[kLbaBatch], rr[kLbaBatch];
#pragma unroll
        for (int j = 0; j < kLbaBatch; ++j) {
            cc[j] = p.data[i + (long long)j * stride];
            rr[j] = p.data2[i + (long long)j * stride];
        }
        double prod = 1.0;
#pragma unroll
        for (int j = 0; j < kLbaBatch; ++j)
            prod *= lba_trial<NA, RS>(tab, na, nu, nuS, kS, bS, tau, inv_A, inv_SA, inv_norm, (int)cc[j], rr[j]);
        if (__builtin_expect(!(prod < 1e300), 0)) {
            // The density scales like 1/b and b = A + k has no lower bound inside the bounds (0, Inf): eight huge factors could
            // leave the double range and log(+Inf) would be accepted for ever.  Cold path: the batch again, a log per trial.
            double s = 0.0;
#pragma unroll 1
            for (int j = 0; j < kLbaBatch; ++j)
                s += log(lba_trial<NA, RS>(tab, na, nu, nuS, kS, bS, tau, inv_A, inv_SA, inv_norm, (int)p.data[i + (long long)j * stride],
                                       p.data2[i + (long long)j * stride]));
            acc += s;
        } else
            acc += log(prod);
    }
    double prod = 1.0;
    for (; i < i1; i += stride)
        prod *= lba_trial<NA, RS>(tab, na, nu, nuS, kS, bS, tau, inv_A, inv_SA, inv_norm, (int)p.data[i], p.data2[i]);
    return acc + log(prod);
}

// LNR log-likelihood of the trials i0, i0+stride, ... < i1 for one proposal (lognormal_race_tests.jl:9-12); tab = kLogPhiTable
__device__ __forceinline__ double lnr_range_sum(const KParams& p, const double* th, long long i0, long long i1, int stride, const double* tab) {
    const int na = p.n_acc;
    double nu[8];
    for (int a = 0; a < 8; ++a) nu[a] = a < na ? th[a] : 0.0;
    const double tau = th[na], sg = p.c0, lsg = log(sg), isg = 1.0 / sg;
    double acc = 0.0;
    for (long long i = i0; i < i1; i += stride) {
        const int c = (int)p.data[i];
        const double t = p.data2[i] - tau;
        double ll = 0.0;
        if (!(t > 0.0))
            ll = -INFINITY;
        else {
            const double lt = log(t);
            for (int a = 0; a < na; ++a) {
                const double z = (lt - nu[a]) * isg;
                ll += (a + 1 == c) ? (-(z * z + kLog2Pi) / 2.0 - lsg - lt) : log_Phi_neg_table(tab, z);
            }
        }
        acc += ll;
    }
    return acc;
}

// Sum over observations i = i0, i0+stride, ... < i1 of the per-observation statistic of one proposal `th`
// (Gaussian: z^2; Binomial: the log-density; hierarchical families: one term per subject; rastrigin: the objective on i == 0).
// The race models (LBA, LNR) are never summed by the proposal kernel's own lanes (thousands of trials, a table in LDS):
// they live in lba_range_sum / lnr_range_sum, inlined into k_obs_loglike alone -- as cases of this function they made it too
// large to inline anywhere, and every caller paid a call with its kernarg spilled to scratch.
__device__ inline double obs_range_sum(const KParams& p, const double* th, long long i0, long long i1, int stride) {
    double acc = 0.0;
    switch (p.family) {
        case FAM_GAUSSIAN: {  // sum_i ((x_i - mu)/sigma)^2   Gaussian_Example.jl:26-28
            const double mu = th[0], isg = 1.0 / th[1];
            for (long long i = i0; i < i1; i += stride) {
                const double z = (p.data[i] - mu) * isg;
                acc = fma(z, z, acc);
            }
        } break;
        case FAM_BINOMIAL: {  // binomial_tests.jl:15-17; data=[n], data2=[k], then the log binomial coefficients
            const double pr = th[0];
            const double lp = log(pr), l1p = log1p(-pr);
            const double* lgc = p.data2 + p.N;
            for (long long i = i0; i < i1; i += stride) {
                const double n = p.data[i], k = p.data2[i];
                const double t1 = (k == 0.0) ? 0.0 : k * lp;
                const double t2 = (n - k == 0.0) ? 0.0 : (n - k) * l1p;
                acc += lgc[i] + t1 + t2;
            }
        } break;
        case FAM_HIER_BINOMIAL: {  // k_s ~ Binomial(n, logistic(mu_b0 + b0_s)) (BASELINE cfg4); "observation" = subject
            const double mu0 = th[0], n = p.c0;
            const double* lgc = p.data + p.N;
            for (long long s = i0; s < i1; s += stride) {
                const double eta = mu0 + th[2 + s], k = p.data[s];
                // k log p + (n-k) log(1-p), p = logistic(eta): softplus(eta) = eta + softplus(-eta), so ONE softplus per subject
                acc += lgc[s] - n * softplus_fast(-eta) - (n - k) * eta;
            }
        } break;
        case FAM_HIER_GAUSSIAN: {  // Hierarchical_Example.jl:36-44
            const double mu0 = th[0], sg = th[2 + p.N];
            const int n = p.d;
            const double lsg = log(sg), isg = 1.0 / sg;
            for (long long s = i0; s < i1; s += stride) {
                const double mu = mu0 + th[2 + s];
                double l = 0.0;
                for (int i = 0; i < n; ++i) {
                    const double z = (p.data[s * n + i] - mu) * isg;
                    l += -(z * z + kLog2Pi) / 2.0 - lsg;
                }
                acc += l;
            }
        } break;
        case FAM_RASTRIGIN: {  // optimization_tests.jl:15-23
            if (i0 == 0) {
                acc = 10.0 * p.D;
                for (int j = 0; j < p.D; ++j) acc += th[j] * th[j] - 10.0 * cos(2.0 * kPi * th[j]);
            }
        } break;
        default:
            break;
    }
    return acc;
}

// the same on the two fields it reads (for a caller that holds its parameters in another address space)
__device__ inline int decide_mh(int mode, int update_kind, double u, double wp, double w, double adj) {
    if (mode == MODE_IDENT) return 1;
    if (update_kind == 1) return wp > w;
    if (update_kind == 2) return wp < w;
    const double e = exp(wp - w + adj);
    return (e >= 1.0) || (u <= e);
}
// mh_update! / maximize! / minimize! decision for one particle (utilities.jl:55-58, 201-226)
// u = the particle's accept uniform (Philox block 3 of its PART stream)
template <bool PLAIN = false>
__device__ inline int decide(const KParams& p, double u, double wp, double w, double adj) {
    if (p.mode == MODE_IDENT) return 1;
    if (!PLAIN) {
        if (p.update_kind == 1) return wp > w;
        if (p.update_kind == 2) return wp < w;
    }
    const double e = exp(wp - w + adj);  // min(1, NaN) = NaN in Julia -> `rand() <= NaN` is false -> reject
    return (e >= 1.0) || (u <= e);
}

// ------------------------------------------------------------------------------------------------
// K1: proposals.  grid = n_groups x n_split workgroups; workgroup (g, sp) proposes for a slice of the
// active particles of group g.  TILE = true: the partner pool of the phase (plus the workgroup's own moving rows) is
// staged in LDS and every partner / base / current row is read from there (partner rows never re-read HBM); TILE = false
// (tile too large, or history partners): rows come from theta / hist in HBM-L2.  Either way a phase-start snapshot: in the
// unfused path theta is only written by K3, and in the fused path (two_colour) a phase writes active rows only
// while partners come from the other colour.
//
// Instruction economy (at D = 32 the pass is bound by VALU issue and latency, not by bandwidth):
//   - everything that is one value per PARTICLE (Philox blocks for snooker/base, indices, gammas, accept) is drawn
//     ONCE, four lanes per particle, in the plan stage and read back from LDS by the particle's lanes; without a plan
//     lane b of the sub-group computes block b and the words are handed out on the DPP network;
//   - one noise block per lane and dim pair; per-dimension constants come packed in one 48-byte DimTab;
//   - no FP64 division in the per-dimension path (reciprocal scales are precomputed on the host).
//
// Fused tails (one kernel instance each, template TAIL):
//   TAIL_PREP / TAIL_PREP_MFMA (MvNormal families): y = A^-1 theta~ and a = theta~.y from an LDS copy of theta' and A^-1
//               (on the matrix cores for d <= 32); SUFFSTAT mode also forms S = y . sum_i x~_i here.
//   TAIL_OBS    (few observations, or one term per scalar): the particle's lanes sum the per-observation terms.
//   fuse_accept (a tail that leaves nothing for K2, two_colour schedule): finishes the whole update -- prior +
//               loglike, Metropolis accept, theta/weight write-back and the history row -- in the same launch.
// ------------------------------------------------------------------------------------------------
template <int B>
__device__ inline U4 bcast_u4(const U4& v, int lpp, int sub_base) {
    U4 r;
    r.x = subgroup_bcast<B>(v.x, lpp, sub_base);
    r.y = subgroup_bcast<B>(v.y, lpp, sub_base);
    r.z = subgroup_bcast<B>(v.z, lpp, sub_base);
    r.w = subgroup_bcast<B>(v.w, lpp, sub_base);
    return r;
}

// x-dependent part of one scalar's log-prior; DimTab.b holds 1/scale for Normal / half-Cauchy
__device__ inline double prior_term(const DimTab& t, double x, double inv_sref, double log_sref) {
    switch (t.kind) {
        case PR_NORMAL: {
            const double z = (x - t.a) * t.b;
            return t.c - 0.5 * (z * z);
        }
        case PR_NORMAL_REF: {
            const double z = (x - t.a) * inv_sref;
            return -(z * z + kLog2Pi) / 2.0 - log_sref;
        }
        case PR_HALFCAUCHY: {
            if (x < 0.0) return -INFINITY;
            const double z = (x - t.a) * t.b;
            return t.c - log1p(z * z);
        }
        case PR_UNIFORM:
            return (x >= t.a && x <= t.b) ? t.c : -INFINITY;
        case PR_BETA: {
            if (x < 0.0 || x > 1.0) return -INFINITY;
            const double t1 = (t.a == 1.0) ? 0.0 : (t.a - 1.0) * log(x);
            const double t2 = (t.b == 1.0) ? 0.0 : (t.b - 1.0) * log1p(-x);
            return t1 + t2 + t.c;
        }
        case PR_GAMMA:  // a = shape, b = 1/scale
            return x > 0.0 ? (t.a - 1.0) * log(x) - x * t.b + t.c : -INFINITY;
        case PR_EXPONENTIAL:  // b = 1/scale
            return x >= 0.0 ? t.c - x * t.b : -INFINITY;
        case PR_LOGNORMAL: {  // b = 1/sigma
            if (!(x > 0.0)) return -INFINITY;
            const double lx = log(x), z = (lx - t.a) * t.b;
            return t.c - lx - 0.5 * (z * z);
        }
        case PR_CAUCHY: {  // b = 1/scale
            const double z = (x - t.a) * t.b;
            return t.c - log1p(z * z);
        }
        default:
            return 0.0;
    }
}

// Rare paths kept OUT OF LINE: inlined into the unrolled per-scalar code they made the colour phase ~10 000 instructions of
// straight-line code (eight copies each of log / sqrt / sincospi and of the whole prior switch) -- more than the instruction
// cache holds, so every phase re-fetched its own code.  The common path (crossover proposal, Normal prior) is a few hundred.
__device__ __attribute__((noinline)) double prior_term_outofline(const DimTab* t, double x) { return prior_term(*t, x, 0.0, 0.0); }
__device__ __attribute__((noinline)) double prior_term_ref_outofline(const DimTab* t, double x, double inv_sref, double log_sref) {
    return prior_term(*t, x, inv_sref, log_sref);
}
__device__ __attribute__((noinline)) U4 draw_block_outofline(uint64_t seed, uint32_t stream, uint32_t sweep, uint64_t iter, uint32_t entity,
                                                            uint32_t block) {
    return draw_block(seed, stream, sweep, iter, entity, block);
}
// mutation noise of one dim pair (mutation.jl:15-18): Box-Muller on the pair's two 32-bit uniforms
__device__ inline double2 box_muller(uint32_t w0, uint32_t w1) {
    const double rad = sqrt(-2.0 * log(1.0 - u32unit(w0)));
    double sn, cs;
    sincospi(2.0 * u32unit(w1), &sn, &cs);
    return make_double2(rad * cs, rad * sn);
}
__device__ __attribute__((noinline)) double2 box_muller_outofline(uint32_t w0, uint32_t w1) { return box_muller(w0, w1); }

// In-kernel stamps (diagnostic build only: make STAMPS=1; tools/k1_stamps.py reads them back through the trace).
// Thread 0 of every workgroup (as many as fit in the P-long trace array) stores the s_memtime delta since kernel start into tr_w[24*blockIdx.x + i]; the
// product build compiles them away.
#ifndef DEMC_STAMP_PASS
#define DEMC_STAMP_PASS 1  // which pass of the workgroup the per-pass stamps sample (0 = the cold first pass)
#endif
#ifdef DEMC_STAMPS_TIMELINE  // (k_longrow: start and end of every particle instead of the stamps -- tools/k1_stamps.py)
#define DEMC_STAMP_SLOTS 0
#else
#define DEMC_STAMP_SLOTS 24
#endif
#ifdef DEMC_STAMPS
#define DEMC_STAMP(i)                                                                                              \
    do {                                                                                                           \
        if (DEMC_STAMP_SLOTS && threadIdx.x == 0 && ((long long)blockIdx.x + 1) * 24 <= p.P)                      \
            p.tr_w[blockIdx.x * 24 + (i)] = (double)(__builtin_amdgcn_s_memtime() - t_start__);                   \
    } while (0)
#define DEMC_STAMP_AT(i, thr, val)                                                                                 \
    do {                                                                                                           \
        if (DEMC_STAMP_SLOTS && threadIdx.x == (thr) && ((long long)blockIdx.x + 1) * 24 <= p.P) p.tr_w[blockIdx.x * 24 + (i)] = (double)(val); \
    } while (0)
#define DEMC_STAMP_NOW() (__builtin_amdgcn_s_memtime() - t_start__)
#define DEMC_STAMP_INIT() unsigned long long t_start__ = __builtin_amdgcn_s_memtime()
#define DEMC_STAMP_RESET() t_start__ = __builtin_amdgcn_s_memtime()  // resident form: stamps are relative to the step's start
#else
#define DEMC_STAMP(i) \
    do {              \
    } while (0)
#define DEMC_STAMP_AT(i, thr, val) \
    do {                           \
    } while (0)
#define DEMC_STAMP_NOW() 0
#define DEMC_STAMP_INIT() \
    do {                  \
    } while (0)
#define DEMC_STAMP_RESET() \
    do {                   \
    } while (0)
#endif

// One LDS-DMA instruction: every active lane moves 16 bytes from its own global address to (wave-uniform LDS address) +
// lane*16.  Written as assembly rather than __builtin_amdgcn_global_load_lds so that the compiler does not track it:
// with the builtin, the first LDS access after the copy (which "may alias" its destination) gets an s_waitcnt vmcnt(0),
// i.e. the prologue would stall on the copy it is meant to overlap.  The kernel waits for the copy itself (vmcnt(0) +
// barrier) before the first read of the tile; compiler-inserted vmcnt waits in between can only be stricter than needed,
// because a wave's memory operations retire in issue order.
__device__ inline void lds_dma16(const double* g, double* l) {
    const uint32_t lds_addr = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) void*)l);
    uint32_t saved_m0;  // M0 carries the LDS address of the copy; put back what the compiler may have had in it
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(saved_m0)
                 : "v"(g), "s"(lds_addr)
                 : "memory");
}

// max over the 64 lanes of a wave, returned to all of them: butterfly inside the 16-lane rows on the DPP network, then
// the four row results through scalar registers
__device__ inline double wave_max(double v) {
    v = fmax(v, dpp_mov<kDppXor1>(v));
    v = fmax(v, dpp_mov<kDppXor2>(v));
    v = fmax(v, dpp_mov<kDppHalfMirror>(v));
    v = fmax(v, dpp_mov<kDppMirror>(v));
    const int lo = __double2loint(v), hi = __double2hiint(v);
    double r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        r[k] = __hiloint2double(__builtin_amdgcn_readlane(hi, 16 * k), __builtin_amdgcn_readlane(lo, 16 * k));
    return fmax(fmax(r[0], r[1]), fmax(r[2], r[3]));
}

// Cumulative weights of up to 64 R values held one per lane and register (index = lane + 64 r; entries beyond n must be
// 0.0), in place, in the FIXED three-level order the oracle uses (the CPU checker restates it as cdf_fixed_order): sequential
// left-to-right prefix sums inside quads, a sequential prefix over the four quad totals of a chunk of 16, a sequential prefix
// over the chunk totals, cdf[i] = off[chunk] + (o[quad] + c[i]).  A chunk is one DPP row: three rounds of
// c = row_shr:1(c) * (lane & 3 ? 1 : 0) + e leave the quad prefixes in every lane (one fma each: x * 1 + e rounds once, and
// lane k is recomputed to the same value once it is final), the three quad totals a lane may need come by row_newbcast, the
// chunk totals travel through v_readlane.  Round 3 had fifteen dependent row_shr rounds per chunk (two levels); every level
// adds non-negative numbers left to right, so the result is monotone in either form.  No LDS round trips.
template <int R>
__device__ inline void wave_cdf(double (&e)[R], int n) {
    const int lane = threadIdx.x & 63;
    const double qm = (lane & 3) ? 1.0 : 0.0;
    double pre[R];
#pragma unroll
    for (int r = 0; r < R; ++r) pre[r] = e[r];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double sh = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(pre[r]), 0x111, 0xf, 0xf, true),
                                               __builtin_amdgcn_update_dpp(0, __double2loint(pre[r]), 0x111, 0xf, 0xf, true));
            pre[r] = fma(sh, qm, e[r]);  // row_shr:1 feeds +0.0 into lane 0 of the row; qm cuts the chain at every quad
        }
    const int quad = (lane >> 2) & 3;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double t0 = dpp_mov<0x153>(pre[r]), t1 = dpp_mov<0x157>(pre[r]), t2 = dpp_mov<0x15B>(pre[r]);  // row_newbcast:3 / 7 / 11
        const double s01 = t0 + t1, s012 = s01 + t2;
        const double o = quad == 0 ? 0.0 : quad == 1 ? t0 : quad == 2 ? s01 : s012;
        pre[r] = o + pre[r];
    }
    double off = 0.0, offv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        offv[r] = 0.0;
#pragma unroll
        for (int row = 0; row < 4; ++row) {
            if (16 * (4 * r + row) < n) {  // (uniform)
                const double t = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(pre[r]), 16 * row + 15),
                                                  __builtin_amdgcn_readlane(__double2loint(pre[r]), 16 * row + 15));
                if ((lane >> 4) == row) offv[r] = off;
                off = off + t;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) e[r] = offv[r] + pre[r];
}
// The same order for ONE chunk of 16 held by one lane (the LDS forms: pools of more than 256, k_mig_pack): v[k] becomes the
// chunk-local prefix o[quad] + c[k]; entries beyond the pool must be 0.0.  Returns the chunk's total.
__device__ inline double chunk16_prefix(double (&v)[16]) {
    double o = 0.0, last = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double c = v[4 * q];
        v[4 * q] = last = o + c;
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            c = c + v[4 * q + k];
            v[4 * q + k] = last = o + c;
        }
        o = o + c;
    }
    return last;
}

// Orders the LDS operations of ONE wave (which the hardware executes in issue order) against compiler reordering: enough
// for data handed between lanes of the same wave.
__device__ inline void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's global-memory counter, which
// would make every barrier of the prologue wait for the tile copy (LDS-DMA) the prologue is supposed to overlap with.
__device__ inline void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---- streaming-resident form: cross term S_p = sum_i y_p . x~_i of the workgroup's observation chunk on the matrix cores.
// One wave: MT particle tiles of 16 (A fragments from the y rows in LDS, kept in registers) x the observation tiles
// [t_lo, t_hi) of `xsrc` (fragment order [tile][kstep][64 lanes], LDS copy or global Xf), accumulated inside the MFMA
// accumulators; the 16 columns of each result are folded on the DPP network and lane (l & 15) == 0 of every row leaves the
// per-particle sums in out[particle].  Same operand layout as k_cross_mfma below.
typedef __attribute__((address_space(3))) const double* lds_cptr;
typedef __attribute__((address_space(3))) double* lds_ptr;
typedef __attribute__((address_space(1))) const double* glb_cptr;
// XP = lds_cptr or glb_cptr: the address space is part of the type so that the B loads are ds_read / global_load (a generic
// pointer makes them flat loads, which wait on both memory counters and serialise the prefetch against the MFMAs).
// `zt` = index (relative to xsrc) of an all-zero tile that absorbs the odd tail of the two-tile ping-pong.
// The tile loop of cross_tiles for KS = 2, MT = 2 out of an LDS copy (BASELINE cfg2: d = 8, 32 moving particles per group),
// as ONE asm statement.  Written with the builtin, inside the resident kernels (256 VGPRs live), the compiler keeps the
// loop-carried accumulators in VGPRs while selecting the AGPR form of the instruction: every trip copies 16 registers into
// AGPRs, runs its 8 MFMAs, idles until the last has drained (s_nop 15) and copies them back -- 108 cycles per MFMA measured
// against 64 for the bare chain (tools/mfma_f64_chain.hip, tools/cross_stage_bench.hip).  Here the accumulators are operands
// of the one statement, so they stay in AGPRs from the first tile to the last.  n >= 1 tiles from LDS byte address va
// (the lane's element of tile 0; a tile is 2 k-steps x 64 lanes x 8 bytes); two register sets in ping-pong, one tile always
// in flight, every load in bounds; four-tile trips first, then at most one two-tile trip, then a tail of one or two tiles.  Per accumulator the products are added in the order of the builtin loop: same bits.
// Wait states (cdna_hip_programming.md 5.7): two before the first MFMA reads operands the compiler has just written (s_nop 1);
// MFMA -> MFMA taking D whole as C: none; D -> any other reader: 18 for this 16-pass instruction (s_nop 15 + s_nop 7 closing
// the string).  Every ds_read of the statement has landed (lgkmcnt(0)) before its last MFMA group issues.
#define DEMC_MFMA4(bx, by)                                              \
    "v_mfma_f64_16x16x4_f64 %[c0], %[a00], %[" #bx "], %[c0]\n\t"        \
    "v_mfma_f64_16x16x4_f64 %[c1], %[a10], %[" #bx "], %[c1]\n\t"        \
    "v_mfma_f64_16x16x4_f64 %[c0], %[a01], %[" #by "], %[c0]\n\t"        \
    "v_mfma_f64_16x16x4_f64 %[c1], %[a11], %[" #by "], %[c1]\n\t"
__device__ __forceinline__ void cross_loop_lds_2x2(d4& c0, d4& c1, double a00, double a01, double a10, double a11, unsigned va, int n) {
    double p0, p1, q0, q1;    // the two register sets: k-steps 0 and 1 of a tile
    int nq = (n - 1) >> 2;    // four-tile trips (one vector instruction per sixteen MFMAs); they leave one to four tiles
    const int r = n - 4 * nq;
    int np = (r - 1) >> 1;    // then two-tile trips (none or one); they leave one or two tiles for the tail
    const int two = r - 2 * np - 1;  // 1: the tail has two tiles
    asm volatile(
        "s_nop 1\n\t"
        "ds_read_b64 %[p0], %[va]\n\t"
        "ds_read_b64 %[p1], %[va] offset:512\n\t"
        "s_cmp_eq_u32 %[nq], 0\n\t"
        "s_cbranch_scc1 6f\n"
        "5:\n\t"
        "ds_read_b64 %[q0], %[va] offset:1024\n\t"
        "ds_read_b64 %[q1], %[va] offset:1536\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        DEMC_MFMA4(p0, p1)
        "ds_read_b64 %[p0], %[va] offset:2048\n\t"
        "ds_read_b64 %[p1], %[va] offset:2560\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        DEMC_MFMA4(q0, q1)
        "ds_read_b64 %[q0], %[va] offset:3072\n\t"
        "ds_read_b64 %[q1], %[va] offset:3584\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        DEMC_MFMA4(p0, p1)
        "ds_read_b64 %[p0], %[va] offset:4096\n\t"
        "ds_read_b64 %[p1], %[va] offset:4608\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        DEMC_MFMA4(q0, q1)
        "v_add_u32 %[va], 0x1000, %[va]\n\t"
        "s_sub_u32 %[nq], %[nq], 1\n\t"
        "s_cmp_lg_u32 %[nq], 0\n\t"
        "s_cbranch_scc1 5b\n"
        "6:\n\t"
        "s_cmp_eq_u32 %[np], 0\n\t"
        "s_cbranch_scc1 2f\n"
        "1:\n\t"
        "ds_read_b64 %[q0], %[va] offset:1024\n\t"
        "ds_read_b64 %[q1], %[va] offset:1536\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        DEMC_MFMA4(p0, p1)
        "ds_read_b64 %[p0], %[va] offset:2048\n\t"
        "ds_read_b64 %[p1], %[va] offset:2560\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        DEMC_MFMA4(q0, q1)
        "v_add_u32 %[va], 0x800, %[va]\n\t"
        "s_sub_u32 %[np], %[np], 1\n\t"
        "s_cmp_lg_u32 %[np], 0\n\t"
        "s_cbranch_scc1 1b\n"
        "2:\n\t"
        "s_cmp_eq_u32 %[two], 0\n\t"
        "s_cbranch_scc1 3f\n\t"
        "ds_read_b64 %[q0], %[va] offset:1024\n\t"
        "ds_read_b64 %[q1], %[va] offset:1536\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        DEMC_MFMA4(p0, p1)
        "s_waitcnt lgkmcnt(0)\n\t"
        DEMC_MFMA4(q0, q1)
        "s_branch 4f\n"
        "3:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        DEMC_MFMA4(p0, p1)
        "4:\n\t"
        "s_nop 15\n\t"
        "s_nop 7"
        : [c0] "+a"(c0), [c1] "+a"(c1), [p0] "=&v"(p0), [p1] "=&v"(p1), [q0] "=&v"(q0), [q1] "=&v"(q1), [va] "+v"(va), [np] "+s"(np), [nq] "+s"(nq)
        : [a00] "v"(a00), [a01] "v"(a01), [a10] "v"(a10), [a11] "v"(a11), [two] "s"(two)
        : "scc", "memory");
}
#undef DEMC_MFMA4
// ASMLOOP: take the one-statement tile loop where it exists (KS = 2, MT = 2, LDS copy).  Only k_res_mvn asks for it: inside
// the general k_propose<512,...,STREAM> (164 KB of code, 128 VGPRs + 128 AGPRs once the statement pins accumulators to
// AGPRs) the dispatch died with HSA_STATUS_ERROR_INVALID_ISA even on shapes that never reach the statement.  Narrowed down, not
// understood: the same kernel runs with its accumulators forced into AGPRs by an empty "+a" statement, and with a statement
// of LDS reads only; it dies as soon as a statement holds a v_mfma_f64_16x16x4_f64 (aligned operands, no branches, identical
// kernel descriptor apart from the scratch size, identical opcode set).  That kernel keeps the builtin loop it is tested with.
template <int KS, int MT, typename XP, bool ASMLOOP = false>
__device__ inline void cross_tiles(lds_cptr ybuf, int dpad, int n_act, int ptile0, XP xsrc, int ksx, int t_lo, int t_hi, int zt,
                                   lds_ptr out, int lane) {
    double a[MT][KS];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int q = (ptile0 + mt) * 16 + (lane & 15);
        lds_cptr yrow = ybuf + (q < n_act ? q : 0) * dpad + (lane >> 4);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const double v = yrow[4 * ks];
            a[mt][ks] = q < n_act ? v : 0.0;
        }
    }
    d4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (d4){0.0, 0.0, 0.0, 0.0};
    const int tstride = ksx * 64;
    XP xb = xsrc + lane;
    if constexpr (ASMLOOP && KS == 2 && MT == 2 && std::is_same<XP, lds_cptr>::value) {
        const int tl = __builtin_amdgcn_readfirstlane(t_lo), th = __builtin_amdgcn_readfirstlane(t_hi);  // wave-uniform by construction
        if (th > tl) cross_loop_lds_2x2(acc[0], acc[1], a[0][0], a[0][1], a[1][0], a[1][1], (unsigned)(size_t)(xb + tl * tstride), th - tl);
    } else {
    double b0[KS], b1[KS];
    {
        XP x = xb + (t_lo < t_hi ? t_lo : zt) * tstride;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) b0[ks] = x[ks * 64];
    }
    // two register sets in ping-pong: the loads of tile t+1 are in flight while tile t feeds the matrix core
    for (int t = t_lo; t < t_hi; t += 2) {
        {
            XP x = xb + (t + 1 < t_hi ? t + 1 : zt) * tstride;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) b1[ks] = x[ks * 64];
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt][ks], b0[ks], acc[mt], 0, 0, 0);
        {
            XP x = xb + (t + 2 < t_hi ? t + 2 : zt) * tstride;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) b0[ks] = x[ks * 64];
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt][ks], b1[ks], acc[mt], 0, 0, 0);
    }
    }
    // four rows per lane, each summed over its 16-lane group: fold4_over16 (a transposition, not four separate sums)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const double u = fold4_over16(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3], lane);
        const int q = (ptile0 + mt) * 16 + (lane >> 4) + 4 * fold4_row(lane);
        if ((lane & 12) == 0 && q < n_act) out[q] = u;
    }
}
// particle tiles in groups of MT (registers: KS x MT fragments), as few accumulators as the tiles need
template <int KS, typename XP, bool ASMLOOP = false>
__device__ inline void cross_ks(lds_cptr ybuf, int dpad, int n_act, XP xsrc, int t_lo, int t_hi, int zt, lds_ptr out, int lane) {
    const int n_pt = (n_act + 15) >> 4;
    constexpr int MTMAX = KS >= 16 ? 2 : 4;
    int p0 = 0;
    for (; p0 + MTMAX <= n_pt; p0 += MTMAX) cross_tiles<KS, MTMAX, XP, ASMLOOP>(ybuf, dpad, n_act, p0, xsrc, KS, t_lo, t_hi, zt, out, lane);
    const int rest = n_pt - p0;
    if (MTMAX == 4 && rest == 3) cross_tiles<KS, 4, XP, ASMLOOP>(ybuf, dpad, n_act, p0, xsrc, KS, t_lo, t_hi, zt, out, lane);
    else if (MTMAX == 4 && rest == 2) cross_tiles<KS, 2, XP, ASMLOOP>(ybuf, dpad, n_act, p0, xsrc, KS, t_lo, t_hi, zt, out, lane);
    else if (rest >= 1) cross_tiles<KS, 1, XP, ASMLOOP>(ybuf, dpad, n_act, p0, xsrc, KS, t_lo, t_hi, zt, out, lane);
}
// dispatch on the number of k-steps (dpad / 4)
template <typename XP>
__device__ inline void cross_stage(lds_cptr ybuf, int dpad, int n_act, XP xsrc, int t_lo, int t_hi, int zt, lds_ptr out, int lane) {
    switch (dpad >> 2) {
        case 1: cross_ks<1, XP>(ybuf, dpad, n_act, xsrc, t_lo, t_hi, zt, out, lane); break;
#ifdef DEMC_ASMLOOP_GENERAL  // (diagnostic build: the one-statement tile loop inside the general kernel, see cross_tiles)
        case 2: cross_ks<2, XP, true>(ybuf, dpad, n_act, xsrc, t_lo, t_hi, zt, out, lane); break;
#else
        case 2: cross_ks<2, XP>(ybuf, dpad, n_act, xsrc, t_lo, t_hi, zt, out, lane); break;
#endif
        case 4: cross_ks<4, XP>(ybuf, dpad, n_act, xsrc, t_lo, t_hi, zt, out, lane); break;
#ifndef DEMC_ASMLOOP_SMALL  // (diagnostic build: the same kernel without its two largest cases -- under the reach of s_branch)
        case 8: cross_ks<8, XP>(ybuf, dpad, n_act, xsrc, t_lo, t_hi, zt, out, lane); break;
        default: cross_ks<16, XP>(ybuf, dpad, n_act, xsrc, t_lo, t_hi, zt, out, lane); break;
#else
        default: break;
#endif
    }
}

typedef __attribute__((address_space(1))) unsigned long long gu64;
// One hand-over granule: {epoch, 32 bits of payload}, written by ONE write-through (sc1) 8-byte store, so the data is its
// own flag (cdna_hip_programming.md section 6, Guideline 16, form R2): no release fence, no separate flag word.
__device__ inline void store_granule(unsigned long long* g, unsigned epoch, unsigned value) {
    __hip_atomic_store((gu64*)g, ((unsigned long long)epoch << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline unsigned long long load_granule(const unsigned long long* g) {
    return __hip_atomic_load((gu64*)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// TAIL selects the fused tail compiled into the instance (the host passes the matching flags in KParams): registers
// are allocated for the worst path of a kernel, so the tails a model cannot take are kept out of its instance.
enum K1Tail : int { TAIL_NONE = 0, TAIL_PREP = 1, TAIL_PREP_MFMA = 2, TAIL_OBS = 3 };
// RES = resident form: ONE workgroup of WG threads owns a group, copies all of its rows and weights into LDS once, and then
// runs both colour phases of p.n_iters iterations (x p.n_sweeps block sweeps) without leaving the kernel: accepted rows are
// written to the LDS copy and through to HBM, a workgroup barrier separates the phases.  Groups never interact inside
// update! (main.jl:135-167), so nothing else is needed between two migrations.  Requires the fused accept tail,
// two_colour, current-population partners.  RES = false is the one-phase-per-launch form with n_split workgroups per group.
// PLAIN = the default sampler and nothing else (random_gamma, no snooker, kappa = 1, no block masks, posterior + Metropolis,
// no trace): the branches of everything else are compiled out of the instance (6 % at cfg3 SUFFSTAT).
// STREAM = streaming-resident form (MvNormal families, STREAMING likelihood, small populations): the resident kernel with
// the observation stream inside.  p.st_C workgroups per group each hold the whole group in LDS and make every proposal of
// the phase (redundantly -- it is cheap), but stream only their own 1/st_C of the observation tiles through the matrix
// cores for all of the phase's proposals; the per-chunk cross terms are handed over between the st_C workgroups of the group
// (8-byte write-through granules that carry their own epoch tag, polled with sc1 loads), summed by every one of them in the
// same fixed order, and every one then makes the same accept decisions and keeps its LDS copy of the group current;
// workgroup 0 of the group writes HBM (state, weights, history).  Every proposal still visits every observation, as the
// reference's loglike does; what disappears is the K1 -> K2 -> K3 launch chain per colour phase.
template <int WG, bool TILE, int TAIL, bool RES, int LEAN, bool STREAM = false>
__global__ __launch_bounds__(WG, (WG == 256 && !STREAM) ? 2 : 1) void k_propose(KParams p0) {
    // LEAN 0: every option of the sampler.  LEAN 1 ("plain"): the default sampler and nothing else.  LEAN 2: the default sampler
    // plus snooker updates and block updates (theta_snooker > 0, blocking_on: what the reference's multivariate, hierarchical,
    // LBA and blocking examples / tests ask of sample()) -- replay, recombination, the optimiser's updates, the other proposal
    // kinds and the trace compiled out as in LEAN 1.
    constexpr bool PLAIN = LEAN != 0;  // none of the general options
    constexpr bool SNK = LEAN != 1;    // snooker branches compiled in
    constexpr bool MSK = LEAN != 1;    // block masks (reset!) compiled in
    constexpr bool FUSE_PREP = TAIL == TAIL_PREP || TAIL == TAIL_PREP_MFMA;
    constexpr bool PREP_MFMA = TAIL == TAIL_PREP_MFMA;
    constexpr bool FUSE_OBS = TAIL == TAIL_OBS;
    static_assert(!RES || TILE, "the resident form keeps the group in LDS");
    static_assert(!STREAM || (RES && FUSE_PREP), "the streaming-resident form is a resident MvNormal instance");
    extern __shared__ double lds[];
    DEMC_STAMP_INIT();
    __shared__ double s_total;
    __shared__ double s_gsum[8];
    __shared__ int s_gsumi[8];
    __shared__ DimSeg s_seg[kMaxDimSeg];
    for (int i = threadIdx.x; i < p0.n_seg * (int)(sizeof(DimSeg) / sizeof(double)); i += WG)
        reinterpret_cast<double*>(s_seg)[i] = reinterpret_cast<const double*>(p0.dimseg)[i];  // visible after the barriers below
    KParams p = p0;  // RES: the per-phase fields (iter, sweep, mask, store_row, active range, pool) are rewritten every step
    const int tid = threadIdx.x;
    // Workgroups are dealt to the 8 XCDs round-robin by blockIdx, and each XCD has its own L2.  The n_split workgroups of
    // a group copy the same partner pool, so they are given blockIdx values 8 apart: same XCD, dispatched back to back,
    // and the second copy is served by that XCD's L2 instead of HBM.  (Plain order when the groups do not divide by 8.)
    int g, sp, c_idx = 0;  // c_idx: which of the st_C workgroups of the group (STREAM)
    if (STREAM) {
        // the st_C workgroups of a group get blockIdx values 8 apart -> one XCD, one L2 for their hand-over granules
        if ((p.n_groups & 7) == 0) {
            const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
            c_idx = j % p.st_C;
            g = (j / p.st_C) * 8 + xcd;
        } else {
            g = blockIdx.x / p.st_C;
            c_idx = blockIdx.x % p.st_C;
        }
        sp = 0;
    } else if (!RES && (p.n_groups & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        sp = j % p.n_split;
        g = (j / p.n_split) * 8 + xcd;
    } else {
        g = blockIdx.x / p.n_split;
        sp = blockIdx.x % p.n_split;
    }
    const int gi = g;  // position in this launch (indexes launch-private buffers); g = the local group it stands for
    if (p.glist) g = p.glist[g];
    const int g_glob = p.group_offset + g;
    const int D = p.D, Np = p.Np, d = p.d;
    const bool even = (D & 1) == 0;
    double* grows = p.theta + (size_t)g * Np * D;
    double* gw = p.weight + (size_t)g * Np;
    const int lpp = p.lpp;
    const int ppp = WG / lpp;  // particles per pass
    const int wave = tid >> 6, lane = tid & 63;
    const int dd = d * d;
    // A^-1 is staged in LDS when it fits next to everything else (host: p.ainv_lds); for wide data (d in the hundreds) the
    // preparation reads it from L2 instead
    const bool ld_ainv = FUSE_PREP && p.Ainv && p.ainv_lds;
    // LDS carve-up (host computes the same sizes): tile | weights (RES) | cdf + chunk totals | A^-1 | theta' scratch | plan
    const int plan_cap = RES ? Np - Np / 2 : (p.n_act + p.n_split - 1) / p.n_split;  // particles a workgroup moves per phase
    double* tile = lds;
    double* w_s = tile + (TILE ? (size_t)p.tile_rows * D : 0);
    double* cdf = w_s + (RES ? Np : 0);
    double* ainv_s = cdf + Np + ((Np + 15) >> 4);
    // STREAM keeps theta' of EVERY moving particle until the cross terms are in: one scratch row per particle, with room for
    // the scalars the accept needs later (aux, S, prior, out-of-bounds flag, snooker adjustment, accept uniform)
    const int scr_stride = STREAM ? D + 6 : D + 2;
    double* xb_s = ainv_s + (ld_ainv ? (size_t)dd : 0);
    double* scr = xb_s + (FUSE_PREP ? d : 0);
    const bool use_scr = FUSE_PREP || FUSE_OBS;  // theta' of the pass kept in LDS for the fused tails
    double* plan_d = scr + p.scr_doubles;  // [plan_cap][4]: g1, g2, accept uniform, select_base uniform
    int* plan_i = reinterpret_cast<int*>(plan_d + 4 * (size_t)plan_cap);  // [plan_cap][4]: snooker?, three row indices
    // STREAM: y rows [st_rows][dpad] | per-wave partial sums [WG/64][st_nact_max] | granule payloads [st_C][st_nact_max] (as
    // 2 x u32) | this workgroup's chunk of Xf (if it fits)
    double* ybuf = reinterpret_cast<double*>(plan_i + 4 * (size_t)plan_cap);
    double* part_l = ybuf + (STREAM ? (size_t)p.st_rows * p.dpad : 0);
    unsigned* part_c = reinterpret_cast<unsigned*>(part_l + (STREAM ? (size_t)(WG / 64) * p.st_nact_max : 0));
    // (16-byte aligned: the chunk is copied by 16-byte LDS-DMA pieces; the host leaves the slack)
    double* xs = reinterpret_cast<double*>((reinterpret_cast<size_t>(part_c + (STREAM ? 2 * (size_t)p.st_C * p.st_nact_max : 0)) + 15) & ~(size_t)15);
    // observation tiles of this workgroup: [xt_lo, xt_hi) of Xf
    const int xt_lo = STREAM ? c_idx * p.st_chunk_tiles : 0;
    const int xt_hi = STREAM ? (xt_lo + p.st_chunk_tiles < p.n_tiles ? xt_lo + p.st_chunk_tiles : p.n_tiles) : 0;

    if (RES) {  // once: every row and weight of the group, A^-1 and xbar
        if (even) {
            const int n16 = (Np * D) >> 1;
            for (int c0 = wave * 64; c0 < n16; c0 += WG)
                if (c0 + lane < n16) lds_dma16(grows + 2 * (size_t)(c0 + lane), tile + 2 * (size_t)c0);
        } else
            for (int i = tid; i < Np * D; i += WG) tile[i] = grows[i];
        for (int i = tid; i < Np; i += WG) w_s[i] = gw[i];
        if (ld_ainv)
            for (int i = tid; i < dd; i += WG) ainv_s[i] = p.Ainv[i];
        if (FUSE_PREP)
            for (int i = tid; i < d; i += WG) xb_s[i] = p.xbar[i];
        if (STREAM) {
            for (int i = tid; i < p.st_rows * p.dpad; i += WG) ybuf[i] = 0.0;  // the k-step padding beyond d stays zero
            if (p.st_x_lds && xt_lo < xt_hi) {  // Xf is contiguous per tile: a linear copy, 16 B per lane, by LDS-DMA
                const double* src = p.Xf + (size_t)xt_lo * (p.dpad >> 2) * 64;
                const int n16 = ((xt_hi - xt_lo) * (p.dpad >> 2) * 64) >> 1;
                for (int c0 = wave * 64; c0 < n16; c0 += WG)
                    if (c0 + lane < n16) lds_dma16(src + 2 * (size_t)(c0 + lane), xs + 2 * (size_t)c0);
            }
            if (p.st_x_lds)  // tile st_chunk_tiles of the LDS copy: all zero, absorbs the odd tail of the tile ping-pong
                for (int i = tid; i < (p.dpad >> 2) * 64; i += WG) xs[(size_t)p.st_chunk_tiles * (p.dpad >> 2) * 64 + i] = 0.0;
        }
        if (even || (STREAM && p.st_x_lds)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const bool wr_hbm = !STREAM || c_idx == 0;  // STREAM: workgroup 0 of the group owns the HBM copy
    double bfrag[2][8];  // A^-1 fragments of the MFMA preparation, loaded once (below)

    const long long n_steps = RES ? (long long)p0.n_iters * p0.n_sweeps * 2 : 1;
    for (long long step = 0; step < n_steps; ++step) {
    if (RES) {  // step -> (iteration, block sweep, colour phase)
        DEMC_STAMP_RESET();
        const int ph = (int)(step & 1);
        const long long s2 = step >> 1;
        const int bsw = (int)(s2 % p0.n_sweeps);
        p.iter = p0.iter + s2 / p0.n_sweeps;
        p.sweep = (unsigned)bsw;
        p.mask = p0.mask ? p0.mask + (size_t)bsw * D : nullptr;  // block_update! main.jl:174-179
        p.store_row = (bsw == p0.n_sweeps - 1 && p0.hist && p.iter - 1 < p0.n_rows) ? p.iter - 1 : -1;
        const int half = Np / 2;  // two_colour: the halves take turns, partners and base come from the resting half
        p.a_lo = ph ? half : 0;
        p.n_act = ph ? Np - half : half;
        p.pool_lo = ph ? 0 : half;
        p.pool_n = ph ? half : Np - half;
    }
    const int per_split = (p.n_act + p.n_split - 1) / p.n_split;
    const int q_lo = sp * per_split;
    const int q_hi = (q_lo + per_split < p.n_act) ? q_lo + per_split : p.n_act;

    // select_base (crossover.jl:282-289) over the partner POOL: the whole group in the synchronous schedule, the fixed
    // half in two_colour -- so nothing a moving particle reads (partners, base row, base weights) can change during
    // the phase, which is what makes the fused accept tail race-free across workgroups.
    const int n_cdf = p.pool_n;
    // (no-tile form with a snapshot of the sweep's start -- KParams::base_weight / base_theta: DE-MC_Z inside burn-in, where the base
    // particle is a row of the current population that another workgroup of the synchronous launch may be writing)
    const double* pw = (RES ? w_s : (!TILE && p.base_weight) ? p.base_weight + (size_t)g * Np : gw) + p.pool_lo;
    // Order of the prologue: a wave's memory results return in issue order, so the few global reads the prologue itself
    // consumes (pool weights, A^-1, xbar) are issued FIRST, all at once, and the bulk tile copy after them; the copy
    // then stays in flight under the softmax prefix sums and the plan stage, none of which touch global memory.
    const bool maybe_base = p.mode == MODE_STEP && p.proposal_kind == 0 && p.iter <= p.burnin;  // crossover.jl:164
    double wv[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, av[4], xv = 0.0;
    if (maybe_base && wave == 0) {  // the softmax below is wave 0's job alone
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (lane + 64 * k < n_cdf) wv[k] = pw[lane + 64 * k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) av[k] = (!RES && ld_ainv && tid + WG * k < dd) ? p.Ainv[tid + WG * k] : 0.0;
    if (!RES && FUSE_PREP && tid < d) xv = p.xbar[tid];

    bool is_mut = false;  // the group's coin, drawn while those loads are in flight
    if (p.mode == MODE_STEP) {
        const U4 r = draw_block(p.seed, S_GROUP, p.sweep, (uint64_t)p.iter, (uint32_t)g_glob, 0);
        double u_coin = u53(r.x, r.y);
        if (!PLAIN) u_coin = replayed(p.rp_group, (size_t)g, u_coin);
        is_mut = u_coin <= p.beta;  // mutate_or_crossover! main.jl:199-207
    }
    const bool de_any = (p.mode == MODE_STEP) && !is_mut;
    const bool use_base = de_any && maybe_base;
    const bool hist_partners = !TILE && p.partner_kind == 1;
    DEMC_STAMP(12);  // group coin drawn

    if (!RES && FUSE_PREP) {
        if (ld_ainv) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (tid + WG * k < dd) ainv_s[tid + WG * k] = av[k];
            for (int i = tid + 4 * WG; i < dd; i += WG) ainv_s[i] = p.Ainv[i];
        }
        if (tid < d) xb_s[tid] = xv;
        for (int i = tid + WG; i < d; i += WG) xb_s[i] = p.xbar[i];
    }
    DEMC_STAMP(14);  // A^-1, xbar parked in LDS; pool weights in wave 0's registers
    // The tile holds what this workgroup can read: the partner pool (all partner / base rows come from it) and, when
    // the moving particles are not pool rows (two_colour), its own slice of them -- two linear pieces of theta.
    double* own = tile + (size_t)p.pool_n * D;
    if (TILE && !RES) {
        const double* src[2] = {grows + (size_t)p.pool_lo * D, grows + (size_t)(p.a_lo + q_lo) * D};
        double* dst[2] = {tile, own};
        const int cnt[2] = {p.pool_n * D, p.own_in_pool ? 0 : (q_hi - q_lo) * D};
#pragma unroll
        for (int piece = 0; piece < 2; ++piece) {
            if (even) {
                // LDS-DMA (global_load_lds_dwordx4): 16 B per lane straight into LDS, no VGPR round trip; the destination
                // of one wave-instruction is a wave-uniform LDS base + lane*16, i.e. exactly a linear copy.
                const int n16 = cnt[piece] >> 1;  // 16-byte pieces
                for (int c0 = wave * 64; c0 < n16; c0 += WG) {
                    if (c0 + lane < n16) lds_dma16(src[piece] + 2 * (size_t)(c0 + lane), dst[piece] + 2 * (size_t)c0);
                }
            } else
                for (int i = tid; i < cnt[piece]; i += WG) dst[piece][i] = src[piece][i];
        }
    }
    DEMC_STAMP(15);  // tile copy issued
    // indexed by the row's position in its group; only pool rows (and, through pt below, own rows) are ever touched
    const double* rows = RES ? (const double*)tile : TILE ? (const double*)tile - (ptrdiff_t)p.pool_lo * D : grows;
    const double* rows_base = (!TILE && p.base_theta) ? p.base_theta + (size_t)g * Np * D : rows;  // where a base row is read from
    if (use_base && wave == 0) {
        // stabilised: e_j = exp(w_j - max w); cumulative weights in the fixed three-level order of the oracle (wave_cdf):
        // sequential inside quads, over the quad totals of a chunk of 16, over the chunk totals.
        // One wave does all of it: its LDS operations execute in order, so the steps need no workgroup barrier (the
        // other three waves are computing their share of the plan meanwhile).
        double m = fmax(fmax(wv[0], wv[1]), fmax(wv[2], wv[3]));
        for (int i = lane + 256; i < n_cdf; i += 64) m = fmax(m, pw[i]);  // pools beyond 256 (these reads queue behind the tile)
        m = wave_max(m);
        if (n_cdf <= 256) {  // the whole pool sits in this wave's registers: prefix sums on the DPP network, one LDS write
            double e[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) e[k] = (lane + 64 * k < n_cdf) ? exp(wv[k] - m) : 0.0;
            if (n_cdf <= 64) {
                double e1[1] = {e[0]};
                wave_cdf<1>(e1, n_cdf);
                e[0] = e1[0];
            } else if (n_cdf <= 128) {
                double e2[2] = {e[0], e[1]};
                wave_cdf<2>(e2, n_cdf);
                e[0] = e2[0]; e[1] = e2[1];
            } else
                wave_cdf<4>(e, n_cdf);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (lane + 64 * k < n_cdf) cdf[lane + 64 * k] = e[k];
            wave_lds_sync();
            if (lane == 0) s_total = cdf[n_cdf - 1];
        } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (lane + 64 * k < n_cdf) cdf[lane + 64 * k] = exp(wv[k] - m);
        for (int i = lane + 256; i < n_cdf; i += 64) cdf[i] = exp(pw[i] - m);
        wave_lds_sync();
        const int n_chunk = (n_cdf + 15) >> 4;
        double* ctot = cdf + Np;
        for (int c = lane; c < n_chunk; c += 64) {
            // 16 reads up front, then the dependent adds in registers (padding with 0.0 leaves a running sum unchanged)
            double v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = (c * 16 + k < n_cdf) ? cdf[c * 16 + k] : 0.0;
            const double pre = chunk16_prefix(v);
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (c * 16 + k < n_cdf) cdf[c * 16 + k] = v[k];
            ctot[c] = pre;
        }
        wave_lds_sync();
        if (lane == 0) {  // chunk offsets, left to right, eight totals per round trip
            double off = 0.0;
            for (int c0 = 0; c0 < n_chunk; c0 += 8) {
                double t[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] = (c0 + k < n_chunk) ? ctot[c0 + k] : 0.0;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (c0 + k < n_chunk) {
                        ctot[c0 + k] = off;
                        off = off + t[k];
                    }
            }
        }
        wave_lds_sync();
        for (int i = lane; i < n_cdf; i += 64) cdf[i] = ctot[i >> 4] + cdf[i];
        wave_lds_sync();
        if (lane == 0) s_total = cdf[n_cdf - 1];
        }
    }
    DEMC_STAMP(0);  // wave 0: softmax prefix sums done (tile still in flight)

    const int n_pass = (q_hi - q_lo + ppp - 1) / ppp;
    const int nblk = hist_partners ? 6 : 4;
    // ---- plan stage.  Everything about a crossover proposal that is one value per PARTICLE -- the snooker coin, the
    // partner / base indices, the gammas, the accept uniform -- is computed here once, four lanes per particle (lane b of
    // the quad evaluates Philox block b; quad_perm hands the blocks round), and parked in LDS.  The passes below, where
    // lpp lanes share a particle, then read it back instead of recomputing it lpp times over.  The base pick needs the
    // prefix sums and follows after a barrier.
    const bool planned = TILE && p.plan && de_any && lpp >= 4;
    const int n_loc = q_hi - q_lo;
    // While wave 0 is busy with the prefix sums the other three waves draw the plan for all of the workgroup's particles.
    const int plan_t0 = use_base ? 64 : 0, plan_q = (WG - plan_t0) >> 2;
    if (planned && tid >= plan_t0) {
        for (int base = 0; base < n_loc; base += plan_q) {
            const int ql = base + ((tid - plan_t0) >> 2), blk = tid & 3;
            const bool ok = ql < n_loc;
            const int pl = p.a_lo + q_lo + (ok ? ql : 0);
            const uint32_t eslot = (uint32_t)g_glob * (uint32_t)Np + (uint32_t)pl;
            const U4 mine = draw_block(p.seed, S_PART, p.sweep, (uint64_t)p.iter, eslot, (uint32_t)blk);
            const U4 r0 = bcast_u4<0>(mine, 4, 0), ri = bcast_u4<1>(mine, 4, 0), rg = bcast_u4<2>(mine, 4, 0),
                     ra = bcast_u4<3>(mine, 4, 0);
            const double u_snk = u53(r0.x, r0.y), u_base = u53(r0.z, r0.w);
            const double u_g1 = u53(rg.x, rg.y), u_g2 = u53(rg.z, rg.w);
            const bool snooker = SNK && u_snk <= p.theta_snooker;  // crossover.jl:31
            int i0, i1, i2 = -1;
            double g1, g2 = 0.0;
            if (snooker) {
                uint32_t a, b, c;  // snooker_update! draws 3 from the whole pool (crossover.jl:241)
                pick_triple(ri.x, ri.y, ri.z, (uint32_t)p.pool_n, a, b, c);
                i0 = (int)a + p.pool_lo; i1 = (int)b + p.pool_lo; i2 = (int)c + p.pool_lo;
                g1 = 1.2 + (2.2 - 1.2) * u_g1;  // crossover.jl:249
            } else {
                uint32_t a, b;
                if (p.exclude_self) {  // setdiff(group, [Pt]) crossover.jl:158
                    pick_pair(ri.x, ri.y, (uint32_t)p.pool_n - 1, a, b);
                    const uint32_t t = (uint32_t)(pl - p.pool_lo);
                    a += (a >= t); b += (b >= t);
                } else
                    pick_pair(ri.x, ri.y, (uint32_t)p.pool_n, a, b);
                i0 = (int)a + p.pool_lo; i1 = (int)b + p.pool_lo;
                if (PLAIN || p.proposal_kind == 0) {
                    g1 = 0.5 + (1.0 - 0.5) * u_g1;  // crossover.jl:162
                    if (use_base) g2 = 0.5 + (1.0 - 0.5) * u_g2;
                } else if (p.proposal_kind == 1)
                    g1 = 2.38;  // crossover.jl:191
                else
                    g1 = 2.38 / sqrt(2.0 * (double)D);  // crossover.jl:218
            }
            if (ok && blk == 0) {
                plan_i[4 * ql + 0] = snooker ? 1 : 0;
                plan_i[4 * ql + 1] = i0;
                plan_i[4 * ql + 2] = i1;
                plan_i[4 * ql + 3] = i2;
                plan_d[4 * ql + 0] = g1;
                plan_d[4 * ql + 1] = g2;
                plan_d[4 * ql + 2] = u53(ra.x, ra.y);
                plan_d[4 * ql + 3] = u_base;
            }
        }
    }
    if (planned) {
        if (use_base) {
            lds_barrier();  // prefix sums (wave 0) and plan records visible to everyone
            DEMC_STAMP(11);
            const double total = cdf[n_cdf - 1];
            for (int ql = tid; ql < n_loc; ql += WG)
                if (plan_i[4 * ql + 0] == 0) {  // select_base for the crossover proposals (crossover.jl:282-289)
                    const double u_base = plan_d[4 * ql + 3];
                    int b2;
                    if (!(total > 0.0) || !(total < INFINITY)) {
                        b2 = (int)(u_base * n_cdf);
                        b2 = b2 < n_cdf ? b2 : n_cdf - 1;
                    } else {  // first i with cdf[i] >= t, else last (cdf is monotone): binary search
                        const double t = u_base * total;
                        int lo = 0, hi = n_cdf - 1;
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            if (cdf[mid] >= t) hi = mid; else lo = mid + 1;
                        }
                        b2 = lo;
                    }
                    plan_i[4 * ql + 3] = b2 + p.pool_lo;
                }
        }
    }
    if (TILE && even && !RES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's LDS-DMA pieces have landed
    __syncthreads();
    DEMC_STAMP(1);  // tile and plan visible

    const int sub = tid / lpp, sl = tid % lpp;
    const int sub_base = (tid & 63) & ~(lpp - 1);  // lane 0 of this sub-group inside its wave
    const double eps = p.eps, eps2 = p.eps - (-p.eps);
    // MFMA preparation (TAIL_PREP_MFMA): y[R particles x d] = theta~[R x d] . A^-1[d x d], R = 64/lpp particles per wave,
    // as one 16x16x4 tile product per 16 columns and k-step.  B fragments (A^-1, lane: k = 4ks + (lane>>4),
    // col = 16nt + (lane&15)) are loaded once per workgroup.
    if (PREP_MFMA && step == 0) {
        const int kq = (tid & 63) >> 4, col = tid & 15;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int k = 4 * ks + kq, c = 16 * nt + col;
                bfrag[nt][ks] = (k < d && c < d) ? ainv_s[k * d + c] : 0.0;
            }
    }

    // ---- fused tail: compute_posterior! + mh_update! + store_samples! for one particle (all lanes of its sub-group) ----
    auto finish_particle = [&](bool valid, int pl, size_t slot, const double* pt, int srow, double prior, int oob, double adj,
                               double aux, double S, double u_acc, int kind, int i0, int i1, int i2) {
        const double w = RES ? w_s[pl] : gw[pl];
        const double sg = (p.family == FAM_MVN_ISO) ? scr[srow * scr_stride + d]
                          : (p.family == FAM_GAUSSIAN) ? scr[srow * scr_stride + 1] : 0.0;
        double wp;
        if (!PLAIN && p.fitness_kind == 1)
            wp = oob ? (p.update_kind == 1 ? -INFINITY : INFINITY) : loglike_from_stats(p, S, aux, sg);
        else
            wp = oob ? -INFINITY : prior + loglike_from_stats(p, S, aux, sg);
        const int acc = decide<PLAIN>(p, u_acc, wp, w, adj);  // every lane of the sub-group holds the same inputs
        if (sl == 0 && valid) {
            if (acc) {
                if (wr_hbm) p.weight[slot] = wp;
                if (RES) w_s[pl] = wp;
            }
            if (!PLAIN && p.trace && wr_hbm) {
                if (!STREAM) {
                    p.tr_idx[slot * 4 + 0] = kind; p.tr_idx[slot * 4 + 1] = i0;
                    p.tr_idx[slot * 4 + 2] = i1; p.tr_idx[slot * 4 + 3] = i2;
                }
                p.tr_w[slot] = wp; p.tr_acc[slot] = (unsigned char)acc; p.prop_adj[slot] = adj;
            }
            if (p.store_row >= 0 && wr_hbm) {
                const size_t hrow = (size_t)p.store_row * p.P + slot;
                if ((PLAIN || p.update_kind == 0) && p.mode == MODE_STEP) {  // utilities.jl:207-208
                    p.acc_hist[hrow] = (unsigned char)acc;
                    p.lp_hist[hrow] = acc ? wp : w;
                }
                p.id_hist[hrow] = (int)p.id[slot];
            }
        }
        if (valid) {
            double* trow = p.theta + slot * D;
            double* lrow = tile + (size_t)pl * D;  // RES: the group's copy in LDS moves with it
            double* hrow = (p.store_row >= 0 && wr_hbm) ? p.hist + ((size_t)p.store_row * p.P + slot) * p.hist_ld : nullptr;
            const double* th = scr + srow * scr_stride;
            if (acc || hrow)
                for (int k = sl; 2 * k < D; k += lpp) {
                    const int j0 = 2 * k;
                    const bool has1 = j0 + 1 < D;
                    const double v0 = acc ? th[j0] : pt[j0];
                    const double v1 = has1 ? (acc ? th[j0 + 1] : pt[j0 + 1]) : 0.0;
                    if (even) {
                        if (acc && wr_hbm) *reinterpret_cast<double2*>(trow + j0) = make_double2(v0, v1);  // utilities.jl:204
                        if (RES && acc) *reinterpret_cast<double2*>(lrow + j0) = make_double2(v0, v1);
                        if (hrow) *reinterpret_cast<double2*>(hrow + j0) = make_double2(v0, v1);  // utilities.jl:170-180
                    } else {
                        if (acc && wr_hbm) { trow[j0] = v0; if (has1) trow[j0 + 1] = v1; }
                        if (RES && acc) { lrow[j0] = v0; if (has1) lrow[j0 + 1] = v1; }
                        if (hrow) { hrow[j0] = v0; if (has1) hrow[j0 + 1] = v1; }
                    }
                }
        }
    };

    DEMC_STAMP(13);  // A^-1 fragments in registers
    for (int pass = 0; pass < n_pass; ++pass) {
        const int q = q_lo + pass * ppp + sub;
        const bool valid = q < q_hi;
        const int pl = p.a_lo + (valid ? q : q_lo);
        const size_t slot = (size_t)g * Np + pl;
        const uint32_t eslot = (uint32_t)g_glob * (uint32_t)Np + (uint32_t)pl;
        const double* pt = (TILE && !p.own_in_pool) ? own + (size_t)(valid ? q - q_lo : 0) * D : rows + (size_t)pl * D;
        const int srow = STREAM ? pass * ppp + sub : sub;  // scratch row: per sub-group, or per particle (STREAM)

        // per-particle Philox blocks: lane b of the sub-group evaluates block b (when the sub-group is wide enough)
        if (pass == DEMC_STAMP_PASS || n_pass == 1) DEMC_STAMP(2);  // top of the steady-state pass
        U4 r0 = {0, 0, 0, 0}, ri = r0, rg = r0, ra = r0, h4 = r0, h5 = r0;
        if (p.mode == MODE_STEP && !planned) {
            if (lpp >= nblk && lpp <= 64) {
                const U4 mine = draw_block(p.seed, S_PART, p.sweep, (uint64_t)p.iter, eslot, (uint32_t)(sl < nblk ? sl : 0));
                r0 = bcast_u4<0>(mine, lpp, sub_base);
                ri = bcast_u4<1>(mine, lpp, sub_base);
                rg = bcast_u4<2>(mine, lpp, sub_base);
                ra = bcast_u4<3>(mine, lpp, sub_base);
                if (hist_partners) {
                    h4 = bcast_u4<4>(mine, lpp, sub_base);
                    h5 = bcast_u4<5>(mine, lpp, sub_base);
                }
            } else if (lpp == 4 && hist_partners) {
                // DE-MC_Z on four lanes per particle: six blocks, TWO per lane (block sl, then block 4 + (sl & 1)) handed round the
                // quad -- round 3 had every lane draw all six itself
                const U4 mine = draw_block(p.seed, S_PART, p.sweep, (uint64_t)p.iter, eslot, (uint32_t)sl);
                r0 = bcast_u4<0>(mine, 4, 0);
                ri = bcast_u4<1>(mine, 4, 0);
                rg = bcast_u4<2>(mine, 4, 0);
                ra = bcast_u4<3>(mine, 4, 0);
                const U4 cells = draw_block(p.seed, S_PART, p.sweep, (uint64_t)p.iter, eslot, (uint32_t)(4 + (sl & 1)));
                h4 = bcast_u4<0>(cells, 4, 0);
                h5 = bcast_u4<1>(cells, 4, 0);
            } else {
                r0 = draw_block(p.seed, S_PART, p.sweep, (uint64_t)p.iter, eslot, 0);
                ri = draw_block(p.seed, S_PART, p.sweep, (uint64_t)p.iter, eslot, 1);
                rg = draw_block(p.seed, S_PART, p.sweep, (uint64_t)p.iter, eslot, 2);
                ra = draw_block(p.seed, S_PART, p.sweep, (uint64_t)p.iter, eslot, 3);
                if (hist_partners) {
                    h4 = draw_block(p.seed, S_PART, p.sweep, (uint64_t)p.iter, eslot, 4);
                    h5 = draw_block(p.seed, S_PART, p.sweep, (uint64_t)p.iter, eslot, 5);
                }
            }
        }

        if (pass == DEMC_STAMP_PASS || n_pass == 1) DEMC_STAMP(3);  // particle Philox blocks handed out
        int kind = 3;  // 0 DE, 1 snooker, 2 mutation, 3 identity
        int i0 = -1, i1 = -1, i2 = -1;
        const double *Pa = pt, *Pb2 = pt, *Pc = pt, *Pbase = pt;
        double g1 = 0.0, g2 = 0.0, cm = 0.0, cn = 0.0;
        bool base_on = false;
        double u_acc = u53(ra.x, ra.y);
        if (!PLAIN) u_acc = replayed(p.rp_part, slot * 5 + 4, u_acc);
        const long long* rpi = (!PLAIN && p.rp_partner) ? p.rp_partner + slot * 3 : nullptr;
        if (planned) {
            const int ql = valid ? q - q_lo : 0;
            kind = plan_i[4 * ql + 0];
            i0 = plan_i[4 * ql + 1]; i1 = plan_i[4 * ql + 2]; i2 = plan_i[4 * ql + 3];
            g1 = plan_d[4 * ql + 0]; g2 = plan_d[4 * ql + 1]; u_acc = plan_d[4 * ql + 2];
            Pa = rows + (size_t)i0 * D; Pb2 = rows + (size_t)i1 * D;
            if (SNK && kind == 1) {
                Pc = rows + (size_t)i2 * D;
                // project(Pm,Pd), project(Pn,Pd): dots over all scalars (utilities.jl:239-246)
                double vm = 0.0, vn = 0.0, vd = 0.0;
                for (int k = sl; 2 * k < D; k += lpp)
                    for (int e = 0; e < 2; ++e) {
                        const int j = 2 * k + e;
                        if (j < D) {
                            const double dj = pt[j] - Pa[j];
                            vm += Pb2[j] * dj; vn += Pc[j] * dj; vd += dj * dj;
                        }
                    }
                vm = group_sum(vm, lpp, s_gsum); vn = group_sum(vn, lpp, s_gsum); vd = group_sum(vd, lpp, s_gsum);
                cm = vm / vd; cn = vn / vd;
            } else if (i2 >= 0) {
                Pbase = rows_base + (size_t)i2 * D;
                base_on = true;
            }
        } else if (p.mode == MODE_STEP) {
            if (is_mut)
                kind = 2;
            else {
                double u_snk = u53(r0.x, r0.y), u_base = u53(r0.z, r0.w);
                double u_g1 = u53(rg.x, rg.y), u_g2 = u53(rg.z, rg.w);
                if (!PLAIN && p.rp_part) {
                    u_snk = replayed(p.rp_part, slot * 5 + 0, u_snk);
                    u_base = replayed(p.rp_part, slot * 5 + 1, u_base);
                    u_g1 = replayed(p.rp_part, slot * 5 + 2, u_g1);
                    u_g2 = replayed(p.rp_part, slot * 5 + 3, u_g2);
                }
                const bool snooker = SNK && u_snk <= p.theta_snooker;  // crossover.jl:31
                kind = snooker ? 1 : 0;
                if (hist_partners) {
                    // resample (crossover.jl:113-124): distinct cells of rows 1:(iter-1) x local particles
                    const uint64_t hd0 = ((uint64_t)h4.y << 32) | h4.x, hd1 = ((uint64_t)h4.w << 32) | h4.z,
                                   hd2 = ((uint64_t)h5.y << 32) | h5.x;
                    const uint64_t ub = (uint64_t)(p.iter - 1), M = ub * (uint64_t)p.P;
                    uint64_t a = mulhi64(hd0, M), b = mulhi64(hd1, M - 1), c = 0;
                    if (b >= a) ++b;
                    if (snooker) {
                        c = mulhi64(hd2, M - 2);
                        const uint64_t lo = a < b ? a : b, hi = a < b ? b : a;
                        if (c >= lo) ++c;
                        if (c >= hi) ++c;
                    }
                    // cell x = (row x mod ub, slot x div ub); the 64-bit division is a ~100-instruction routine: while the number of
                    // cells fits 32 bits (250 rows x 65 536 particles do) the 32-bit one serves (wave-uniform branch)
                    auto cell = [&](uint64_t x) -> const double* {
                        uint64_t row, sl_;
                        if ((M >> 32) == 0) {
                            const uint32_t x32 = (uint32_t)x, u32 = (uint32_t)ub, qd = x32 / u32;
                            sl_ = qd; row = x32 - qd * u32;
                        } else {
                            sl_ = x / ub; row = x - sl_ * ub;
                        }
                        return p.hist + (row * (uint64_t)p.P + sl_) * (uint64_t)p.hist_ld;
                    };
                    Pa = cell(a);
                    Pb2 = cell(b);
                    if (snooker) Pc = cell(c);
                    i0 = (int)(a & 0x7fffffff);
                    i1 = (int)(b & 0x7fffffff);
                    i2 = snooker ? (int)(c & 0x7fffffff) : -1;
                } else if (snooker) {
                    uint32_t a, b, c;  // snooker_update! draws 3 from the whole pool (crossover.jl:241)
                    pick_triple(ri.x, ri.y, ri.z, (uint32_t)p.pool_n, a, b, c);
                    a += p.pool_lo; b += p.pool_lo; c += p.pool_lo;
                    if (rpi) {
                        if (rpi[0] >= 0) a = (uint32_t)rpi[0];
                        if (rpi[1] >= 0) b = (uint32_t)rpi[1];
                        if (rpi[2] >= 0) c = (uint32_t)rpi[2];
                    }
                    Pa = rows + (size_t)a * D; Pb2 = rows + (size_t)b * D; Pc = rows + (size_t)c * D;
                    i0 = (int)a; i1 = (int)b; i2 = (int)c;
                } else {
                    uint32_t a, b;
                    if (p.exclude_self) {  // setdiff(group, [Pt]) crossover.jl:158
                        pick_pair(ri.x, ri.y, (uint32_t)p.pool_n - 1, a, b);
                        const uint32_t t = (uint32_t)(pl - p.pool_lo);
                        a += (a >= t); b += (b >= t);
                    } else
                        pick_pair(ri.x, ri.y, (uint32_t)p.pool_n, a, b);
                    a += p.pool_lo; b += p.pool_lo;
                    if (rpi) {
                        if (rpi[0] >= 0) a = (uint32_t)rpi[0];
                        if (rpi[1] >= 0) b = (uint32_t)rpi[1];
                    }
                    Pa = rows + (size_t)a * D; Pb2 = rows + (size_t)b * D;
                    i0 = (int)a; i1 = (int)b;
                }
                if (snooker) {
                    g1 = 1.2 + (2.2 - 1.2) * u_g1;  // crossover.jl:249
                    // project(Pm,Pd), project(Pn,Pd): dots over all scalars (utilities.jl:239-246)
                    double vm = 0.0, vn = 0.0, vd = 0.0;
                    for (int k = sl; 2 * k < D; k += lpp)
                        for (int e = 0; e < 2; ++e) {
                            const int j = 2 * k + e;
                            if (j < D) {
                                const double dj = pt[j] - Pa[j];
                                vm += Pb2[j] * dj; vn += Pc[j] * dj; vd += dj * dj;
                            }
                        }
                    vm = group_sum(vm, lpp, s_gsum); vn = group_sum(vn, lpp, s_gsum); vd = group_sum(vd, lpp, s_gsum);
                    cm = vm / vd; cn = vn / vd;
                } else {
                    if (PLAIN || p.proposal_kind == 0) {
                        g1 = 0.5 + (1.0 - 0.5) * u_g1;  // crossover.jl:162
                        if (use_base) {
                            g2 = 0.5 + (1.0 - 0.5) * u_g2;
                            const double total = s_total;
                            int b;
                            if (!(total > 0.0) || !(total < INFINITY)) {
                                b = (int)(u_base * n_cdf);
                                b = b < n_cdf ? b : n_cdf - 1;
                            } else {  // first i with cdf[i] >= t, else last: cdf is monotone, so that index is the
                                      // number of entries below t -- counted by the sub-group's lanes in parallel
                                const double t = u_base * total;
                                int cnt = 0;
                                for (int i = sl; i < n_cdf; i += lpp) cnt += (cdf[i] < t) ? 1 : 0;
                                cnt = group_sum(cnt, lpp, s_gsumi);
                                b = cnt < n_cdf ? cnt : n_cdf - 1;
                            }
                            b += p.pool_lo;
                            if (rpi && rpi[2] >= 0) b = (int)rpi[2];
                            Pbase = rows_base + (size_t)b * D;
                            i2 = b;
                            base_on = true;
                        }
                    } else if (p.proposal_kind == 1)
                        g1 = 2.38;  // crossover.jl:191
                    else
                        g1 = 2.38 / sqrt(2.0 * (double)D);  // crossover.jl:218
                }
            }
        }

        // crossover / snooker value of scalar j given its uniform (before recombination! / reset!)
        auto cross = [&](int j, double tj, double uu) -> double {
            const double bj = -eps + eps2 * uu;  // b = Uniform(-eps, eps) crossover.jl:166
            if (SNK && kind == 1) {  // (Pt + gamma*(Pr1 - Pr2)) + b  crossover.jl:253
                const double dj = tj - Pa[j];
                const double t1 = dj * cm - dj * cn;
                return (tj + t1 * g1) + bj;
            }
            // ((Pt + g1*(Pm-Pn)) + g2*(Pb-Pt)) + b  crossover.jl:168
            const double t1 = Pa[j] - Pb2[j];
            double t6 = tj + t1 * g1;
            if (base_on) {
                const double t4 = Pbase[j] - tj;
                t6 = t6 + t4 * g2;
            }
            return t6 + bj;
        };
        // both scalars {2k, 2k+1} of one noise block, after recombination! and reset!
        auto value_pair = [&](int k, double& v0, double& v1) {
            const int j0 = 2 * k, j1 = j0 + 1;
            const bool has1 = j1 < D;
            double t0, t1 = 0.0;
            if (even) {
                const double2 t = *reinterpret_cast<const double2*>(pt + j0);
                t0 = t.x; t1 = t.y;
            } else {
                t0 = pt[j0];
                if (has1) t1 = pt[j1];
            }
            v0 = t0; v1 = t1;
            if (kind == 3) return;
            bool keep0 = false, keep1 = false;  // reset! (crossover.jl:336-352): asked for now, with the row loads
            if (MSK && p.mask) {
                keep0 = !p.mask[j0];
                keep1 = has1 && !p.mask[j1];
            }
            // block sweeps (block_update!, main.jl:174-179): a scalar outside the block keeps its value whatever the crossover
            // proposes (reset!), so neither its noise draw nor the partner rows are needed -- in the hyper-parameter sweep of
            // a hierarchical model that is all but a few scalars of the row.  (Mutation ignores the mask, main.jl:205.)
            if (MSK && kind != 2 && keep0 && (keep1 || !has1)) return;
            // (noise block k >> 1 covers the dim pairs 2(k >> 1) and 2(k >> 1) + 1: four scalars per block)
            const U4 nz = draw_block(p.seed, S_NOISE, p.sweep, (uint64_t)p.iter, eslot, (uint32_t)(k >> 1));
            double u0 = u32unit((k & 1) ? nz.z : nz.x), u1 = u32unit((k & 1) ? nz.w : nz.y);
            if (kind == 2) {  // pt + Normal(0, sigma): mutation.jl:15-18 (block mask ignored, main.jl:205)
                const double rad = sqrt(-2.0 * log(1.0 - u0));
                double sn, cs;
                sincospi(2.0 * u1, &sn, &cs);  // Box-Muller angle 2*pi*u1
                double z0 = rad * cs, z1 = rad * sn;
                if (!PLAIN && p.rp_znoise) {
                    z0 = replayed(p.rp_znoise, slot * D + j0, z0);
                    if (has1) z1 = replayed(p.rp_znoise, slot * D + j1, z1);
                }
                v0 = t0 + p.sigma * z0;
                v1 = t1 + p.sigma * z1;
                return;
            }
            if (!PLAIN && p.rp_noise) {
                u0 = replayed(p.rp_noise, slot * D + j0, u0);
                if (has1) u1 = replayed(p.rp_noise, slot * D + j1, u1);
            }
            v0 = cross(j0, t0, u0);
            if (has1) v1 = cross(j1, t1, u1);
            if (!PLAIN && p.kappa != 1.0) {  // recombination! crossover.jl:301-312
                const U4 rc = draw_block(p.seed, S_RECOMB, p.sweep, (uint64_t)p.iter, eslot, (uint32_t)(k >> 1));
                double c0u = u32unit((k & 1) ? rc.z : rc.x), c1u = u32unit((k & 1) ? rc.w : rc.y);
                if (p.rp_recomb) {
                    c0u = replayed(p.rp_recomb, slot * D + j0, c0u);
                    if (has1) c1u = replayed(p.rp_recomb, slot * D + j1, c1u);
                }
                if (c0u <= 1.0 - p.kappa) v0 = t0;
                if (c1u <= 1.0 - p.kappa) v1 = t1;
            }
            if (keep0) v0 = t0;
            if (keep1) v1 = t1;
        };

        if (pass == DEMC_STAMP_PASS || n_pass == 1) DEMC_STAMP(4);  // indices, gammas, base picked
        int oob = 0;
        double prior = 0.0, s1 = 0.0, s2 = 0.0;
        int ref_cached = -1;
        double inv_sref = 0.0, log_sref = 0.0;
        for (int k = sl; 2 * k < D; k += lpp) {
            const int j0 = 2 * k;
            const bool has1 = j0 + 1 < D;
            // the per-scalar constants go out with the row loads of value_pair (one wait for all of them) instead of
            // starting a second round trip once the proposal values exist
            DimTab tab0, tab1;
            if (p.n_seg > 0) {  // segment of each scalar = number of segment starts at or below it
                int g0 = 0, g1 = 0;
                const int j1 = has1 ? j0 + 1 : j0;
                for (int i = 1; i < p.n_seg; ++i) {
                    const int st = s_seg[i].start;
                    g0 += (j0 >= st) ? 1 : 0;
                    g1 += (j1 >= st) ? 1 : 0;
                }
                tab0 = s_seg[g0].t;
                tab1 = s_seg[g1].t;
            } else {
                tab0 = p.dimtab[j0];
                tab1 = p.dimtab[has1 ? j0 + 1 : j0];
            }
            double v0, v1;
            value_pair(k, v0, v1);
            if (SNK && kind == 1) {  // adjust_loglike norms (crossover.jl:268-273)
                const double a0 = v0 - Pa[j0], b0 = pt[j0] - Pa[j0];
                s1 += a0 * a0; s2 += b0 * b0;
                if (has1) {
                    const double a1 = v1 - Pa[j0 + 1], b1 = pt[j0 + 1] - Pa[j0 + 1];
                    s1 += a1 * a1; s2 += b1 * b1;
                }
            }
            for (int e = 0; e < 2; ++e) {
                if (e == 1 && !has1) break;
                const double v = e ? v1 : v0;
                const DimTab t = e ? tab1 : tab0;
                oob |= !(v >= t.lo && v <= t.hi);  // in_bounds utilities.jl:70-78 (NaN fails)
                if ((PLAIN || p.fitness_kind == 0) && t.kind != PR_FLAT) {
                    if (t.kind == PR_NORMAL_REF && t.ref != ref_cached) {
                        ref_cached = t.ref;
                        double r0v, r1v;
                        value_pair(ref_cached >> 1, r0v, r1v);
                        const double sref = (ref_cached & 1) ? r1v : r0v;
                        inv_sref = 1.0 / sref;
                        log_sref = log(sref);
                    }
                    prior += prior_term(t, v, inv_sref, log_sref);
                }
            }
            if (valid && p.write_prop && wr_hbm) {
                if (even)
                    *reinterpret_cast<double2*>(p.prop + slot * D + j0) = make_double2(v0, v1);
                else {
                    p.prop[slot * D + j0] = v0;
                    if (has1) p.prop[slot * D + j0 + 1] = v1;
                }
            }
            if (use_scr) {
                scr[srow * scr_stride + j0] = v0;
                if (has1) scr[srow * scr_stride + j0 + 1] = v1;
            }
        }
        if (pass == DEMC_STAMP_PASS || n_pass == 1) DEMC_STAMP(5);  // per-dimension loop done
        prior = group_sum(prior, lpp, s_gsum);
        oob = group_sum(oob, lpp, s_gsumi);
        double adj = 0.0;
        if (SNK && kind == 1) {  // (d-1)(log|a| - log|b|): stable form of log(|a|^(d-1)/|b|^(d-1))  crossover.jl:268-273
            s1 = group_sum(s1, lpp, s_gsum);
            s2 = group_sum(s2, lpp, s_gsum);
            adj = (double)(D - 1) * (0.5 * log(s1) - 0.5 * log(s2));
        }

        if (pass == DEMC_STAMP_PASS || n_pass == 1) DEMC_STAMP(6);  // reductions done
        double aux = 0.0, S = 0.0;
        if (FUSE_PREP) {
            // y = A^-1 theta' (FULL) or theta' (ISO) for the data dimensions; each lane owns output columns {2k, 2k+1}.
            // A scratch row is written and read by ONE sub-group, which lives inside one wave: LDS operations of a
            // wave execute in order, so only the compiler must be kept from reordering them.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const double* th = scr + srow * scr_stride;
            // centred proposal mu~ = theta' - xbar (the data were centred the same way at demc_set_model)
            if (PREP_MFMA) {
                // The wave's R = 64/lpp particles are rows 0..R-1 of the A operand (lane: row = lane&15, k = 4ks + lane>>4);
                // the C layout then hands lane l the y of particles (l>>4) + 4r, r = 0..3, at columns l&15 and 16 + (l&15).
                // The two dot products a particle needs (mu~.y and y.sx) are sums over columns = over the 16 lanes of a DPP
                // row; lane 0 of the row parks them in the two spare slots of the particle's scratch row, from where the
                // particle's own lanes pick them up.
                const int lane = tid & 63, kq = lane >> 4, row = lane & 15;
                const int R = 64 / lpp, wrow0 = (tid >> 6) * R;
                const int srow0 = (STREAM ? pass * ppp : 0) + wrow0;  // scratch row of the wave's first particle
                const double* trow = scr + (srow0 + (row < R ? row : 0)) * scr_stride;
                d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
                const bool two_tiles = d > 16;  // columns 16..31 exist
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    if (4 * ks < d) {  // (uniform) k-steps beyond d would multiply zeros
                        const int k = 4 * ks + kq;
                        const double a = (row < R && k < d) ? trow[k] - xb_s[k] : 0.0;
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bfrag[0][ks], acc0, 0, 0, 0);
                        if (two_tiles) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bfrag[1][ks], acc1, 0, 0, 0);
                    }
                }
                const int c0 = row, c1 = row + 16;
                const double xb0 = c0 < d ? xb_s[c0] : 0.0, xb1 = c1 < d ? xb_s[c1] : 0.0;
                const double sx0 = (p.sx && c0 < d) ? p.sx[c0] : 0.0, sx1 = (p.sx && c1 < d) ? p.sx[c1] : 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int pr = kq + 4 * r;  // uniform over the 16 lanes of a DPP row
                    if (pr < R) {
                        double* tr = scr + (srow0 + pr) * scr_stride;
                        const double y0 = c0 < d ? acc0[r] : 0.0, y1 = c1 < d ? acc1[r] : 0.0;
                        double a_ = 0.0, s_ = 0.0;
                        if (c0 < d) a_ = fma(tr[c0] - xb0, y0, a_);
                        if (c1 < d) a_ = fma(tr[c1] - xb1, y1, a_);
                        if (p.sx)
                            s_ = fma(y1, sx1, y0 * sx0);
                        else {
                            const int qr = q_lo + pass * ppp + wrow0 + pr;
                            if (qr < q_hi) {
                                double* yrow = STREAM ? ybuf + (size_t)(qr - q_lo) * p.dpad
                                                      : p.Ypad + ((size_t)g * Np + p.a_lo + qr) * p.dpad;
                                if (c0 < p.dpad) yrow[c0] = y0;
                                if (c1 < p.dpad) yrow[c1] = y1;
                            }
                        }
                        a_ = subgroup_sum(a_, 16);
                        s_ = subgroup_sum(s_, 16);
                        if (row == 0) {
                            tr[D] = a_;
                            tr[D + 1] = s_;
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                aux = th[D];
                S = th[D + 1];
            } else
            for (int k = sl; 2 * k < d; k += lpp) {
                const int c0 = 2 * k, c1 = 2 * k + 1;
                double y0 = 0.0, y1 = 0.0;
                const double* ainv_g = ld_ainv ? (const double*)ainv_s : p.Ainv;
                if (p.Ainv) {
                    if (c1 < d) {
#pragma unroll 4
                        for (int j = 0; j < d; ++j) {
                            const double t = th[j] - xb_s[j];
                            y0 = fma(ainv_g[j * d + c0], t, y0);  // A^-1 is symmetric: column c == row c
                            y1 = fma(ainv_g[j * d + c1], t, y1);
                        }
                    } else
                        for (int j = 0; j < d; ++j) y0 = fma(ainv_g[j * d + c0], th[j] - xb_s[j], y0);
                } else {
                    y0 = th[c0] - xb_s[c0];
                    if (c1 < d) y1 = th[c1] - xb_s[c1];
                }
                aux = fma(th[c0] - xb_s[c0], y0, aux);
                if (c1 < d) aux = fma(th[c1] - xb_s[c1], y1, aux);
                if (p.sx) {
                    S = fma(y0, p.sx[c0], S);
                    if (c1 < d) S = fma(y1, p.sx[c1], S);
                } else if (valid) {
                    double* yrow = STREAM ? ybuf + (size_t)srow * p.dpad : p.Ypad + slot * p.dpad;
                    yrow[c0] = y0;
                    if (c1 < p.dpad) yrow[c1] = (c1 < d) ? y1 : 0.0;
                }
            }
            if (!p.sx && valid && !PREP_MFMA && !STREAM)  // zero the k-step padding beyond d (STREAM: zeroed once, never written)
                for (int c = 2 * ((d + 1) / 2) + sl; c < p.dpad; c += lpp) p.Ypad[slot * p.dpad + c] = 0.0;
            if (!PREP_MFMA) {
                aux = group_sum(aux, lpp, s_gsum);
                S = group_sum(S, lpp, s_gsum);
            }
        }

        if (pass == DEMC_STAMP_PASS || n_pass == 1) DEMC_STAMP(7);  // MvNormal preparation done
        if (FUSE_OBS) {
            // small-N scalar-data families (and the hierarchical ones, whose "observations" are the subjects): the
            // sub-group visits every observation itself (lanes stride over them), reading theta' from its scratch row
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lpp > 64) __syncthreads();  // the row was written by all waves of the workgroup
            S = group_sum(obs_range_sum(p, scr + srow * scr_stride, sl, p.N, lpp), lpp, s_gsum);
        }

        if (!p.fuse_accept) {
            if (sl == 0 && valid) {
                p.prop_prior[slot] = prior;
                p.prop_oob[slot] = oob ? 1 : 0;
                p.prop_adj[slot] = adj;
                if (FUSE_PREP) {
                    p.aux[slot] = aux;
                    if (p.sx) p.partial[slot] = S;
                }
                if (!PLAIN && p.trace) {
                    p.tr_idx[slot * 4 + 0] = kind;
                    p.tr_idx[slot * 4 + 1] = i0;
                    p.tr_idx[slot * 4 + 2] = i1;
                    p.tr_idx[slot * 4 + 3] = i2;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lpp > 64 && use_scr) __syncthreads();  // (workgroup-wide particle) same hand-over as at the end of the pass
            continue;
        }

        if (STREAM) {
            // the cross terms are not in yet: park what the accept needs next to theta' (scratch row of the particle) and
            // go on with the next pass; the second loop below finishes the particles
            if (sl == 0 && valid) {
                double* tr = scr + srow * scr_stride;
                tr[D] = aux; tr[D + 2] = prior; tr[D + 3] = (double)oob; tr[D + 4] = adj; tr[D + 5] = u_acc;
                if (!PLAIN && p.trace && wr_hbm) {
                    p.tr_idx[slot * 4 + 0] = kind; p.tr_idx[slot * 4 + 1] = i0;
                    p.tr_idx[slot * 4 + 2] = i1; p.tr_idx[slot * 4 + 3] = i2;
                }
            }
            continue;
        }
        if (pass == DEMC_STAMP_PASS || n_pass == 1) DEMC_STAMP(8);  // in-kernel observation loop done
        finish_particle(valid, pl, slot, pt, srow, prior, oob, adj, aux, S, u_acc, kind, i0, i1, i2);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // scratch rows are reused by the same sub-group in the next pass
        if (lpp > 64 && use_scr) __syncthreads();
        if (pass == DEMC_STAMP_PASS || n_pass == 1) DEMC_STAMP(9);  // accept + row moves done
    }
    if (STREAM) {
        // ---- the observation stream: cross terms of ALL proposals of the phase against this workgroup's chunk of tiles ----
        __syncthreads();  // y rows of every pass are in LDS
        DEMC_STAMP(16);  // proposals + preparation of every moving particle done
        const int n_act = q_hi - q_lo;  // (q_lo = 0: one workgroup moves the whole colour)
        {
            const int nw = WG / 64;
            const int nt = xt_hi - xt_lo, per_w = (nt + nw - 1) / nw;
            const int t_lo = wave * per_w < nt ? wave * per_w : nt, t_hi = t_lo + per_w < nt ? t_lo + per_w : nt;
            lds_ptr outw = (lds_ptr)(part_l + (size_t)wave * p.st_nact_max);
            if (p.st_x_lds)  // zero tile: the one appended to the LDS copy
                cross_stage<lds_cptr>((lds_cptr)ybuf, p.dpad, n_act, (lds_cptr)xs, t_lo, t_hi, p.st_chunk_tiles, outw, lane);
            else             // zero tile: tile n_tiles of Xf
                cross_stage<glb_cptr>((lds_cptr)ybuf, p.dpad, n_act, (glb_cptr)(p.Xf + (size_t)xt_lo * (p.dpad >> 2) * 64), t_lo, t_hi,
                                      p.n_tiles - xt_lo, outw, lane);
        }
        __syncthreads();
        DEMC_STAMP(17);  // cross terms of this workgroup's chunk done
        // this workgroup's partial per particle (waves in fixed order) -> two granules, tagged with the step's epoch
        const unsigned epoch = (unsigned)(step + 1);
        unsigned long long* gran = p.st_gran + (((size_t)(step & 1) * p.n_groups + gi) * p.st_C) * p.st_nact_max * 2;
        if (tid < n_act) {
            double v = 0.0;
            for (int wv = 0; wv < WG / 64; ++wv) v += part_l[(size_t)wv * p.st_nact_max + tid];
            unsigned long long* mine = gran + ((size_t)c_idx * p.st_nact_max + tid) * 2;
            store_granule(mine, epoch, (unsigned)__double2loint(v));
            store_granule(mine + 1, epoch, (unsigned)__double2hiint(v));
        }
        DEMC_STAMP(18);  // granules stored
        // every workgroup of the group collects all st_C x n_act partials: sweep until every tag carries this epoch
        {
            const int tot = p.st_C * n_act * 2;
            unsigned spins = 0;
            for (;;) {
                int ok = 1;
                for (int e = tid; e < tot; e += WG) {
                    const int cc = e / (2 * n_act), r = e - cc * 2 * n_act;
                    const unsigned long long x = load_granule(gran + (size_t)cc * p.st_nact_max * 2 + r);
                    ok &= (unsigned)(x >> 32) == epoch;
                    part_c[(size_t)cc * p.st_nact_max * 2 + r] = (unsigned)x;
                }
                if (__syncthreads_and(ok)) break;
                if (++spins > (1u << 22)) {  // cannot happen with co-resident workgroups; never spin unbounded
                    if (tid == 0) *p.st_err = 1u;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        DEMC_STAMP(19);  // every chunk's granules collected
        if (tid < n_act) {  // S = sum over the chunks, in chunk order: the same bits in every workgroup of the group
            double S = 0.0;
            for (int cc = 0; cc < p.st_C; ++cc) {
                const unsigned* h2 = part_c + ((size_t)cc * p.st_nact_max + tid) * 2;
                S += __hiloint2double((int)h2[1], (int)h2[0]);
            }
            scr[tid * scr_stride + D + 1] = S;
        }
        __syncthreads();
        for (int pass = 0; pass < n_pass; ++pass) {
            const int q = q_lo + pass * ppp + sub;
            const bool valid = q < q_hi;
            const int pl = p.a_lo + (valid ? q : q_lo);
            const size_t slot = (size_t)g * Np + pl;
            const int srow = pass * ppp + sub;
            const double* tr = scr + (valid ? srow : 0) * scr_stride;
            finish_particle(valid, pl, slot, rows + (size_t)pl * D, valid ? srow : 0, tr[D + 2], (int)tr[D + 3], tr[D + 4], tr[D],
                            tr[D + 1], tr[D + 5], 0, -1, -1, -1);
        }
        DEMC_STAMP(20);  // accept + row moves of every moving particle done
    }
    if (RES) __syncthreads();  // the other colour reads what this phase wrote (rows, weights); scratch and plan are reused
    DEMC_STAMP(10);  // end of the step (of the kernel in the one-phase form)
    }  // step
}

// ------------------------------------------------------------------------------------------------
// The k_propose instances the runtime launches (k1_instance / k1_resident_instance / k1_stream_instance in demc_hip.cpp), in three
// lists: demc_k1_phase.cpp, demc_k1_res.cpp and demc_k1_stream.cpp instantiate one each -- 64 instances of a kernel whose colour
// phase is ~10 000 instructions are most of the library's compile time, and `make -j` builds the units side by side --, and
// demc_hip.cpp declares them extern (DEMC_K1_EXTERN).  <WG, TILE, TAIL, RES, LEAN, STREAM>
// ------------------------------------------------------------------------------------------------
#define DEMC_K1_PHASE_INSTANCES(X) \
    X(256, false, TAIL_NONE, false, 0, false) X(256, false, TAIL_PREP, false, 0, false) \
    X(256, false, TAIL_PREP_MFMA, false, 0, false) X(256, false, TAIL_OBS, false, 0, false) \
    X(256, true, TAIL_NONE, false, 0, false) X(256, true, TAIL_PREP, false, 0, false) \
    X(256, true, TAIL_PREP_MFMA, false, 0, false) X(256, true, TAIL_OBS, false, 0, false) \
    X(256, true, TAIL_NONE, false, 1, false) X(256, true, TAIL_PREP, false, 1, false) \
    X(256, true, TAIL_PREP_MFMA, false, 1, false) X(256, true, TAIL_OBS, false, 1, false) \
    X(512, false, TAIL_NONE, false, 0, false) X(512, false, TAIL_PREP, false, 0, false) \
    X(512, false, TAIL_PREP_MFMA, false, 0, false) X(512, false, TAIL_OBS, false, 0, false) \
    X(256, true, TAIL_NONE, false, 2, false) X(256, true, TAIL_PREP, false, 2, false) \
    X(256, true, TAIL_PREP_MFMA, false, 2, false) X(256, true, TAIL_OBS, false, 2, false) \
    X(256, false, TAIL_NONE, false, 1, false) X(256, false, TAIL_PREP, false, 1, false) \
    X(256, false, TAIL_PREP_MFMA, false, 1, false) X(256, false, TAIL_OBS, false, 1, false) \
    X(256, false, TAIL_NONE, false, 2, false) X(256, false, TAIL_PREP, false, 2, false) \
    X(256, false, TAIL_PREP_MFMA, false, 2, false) X(256, false, TAIL_OBS, false, 2, false)
#define DEMC_K1_RES_INSTANCES(X) \
    X(256, true, TAIL_NONE, true, 0, false) X(256, true, TAIL_PREP, true, 0, false) \
    X(256, true, TAIL_PREP_MFMA, true, 0, false) X(256, true, TAIL_OBS, true, 0, false) \
    X(512, true, TAIL_NONE, true, 0, false) X(512, true, TAIL_PREP, true, 0, false) \
    X(512, true, TAIL_PREP_MFMA, true, 0, false) X(512, true, TAIL_OBS, true, 0, false) \
    X(256, true, TAIL_NONE, true, 1, false) X(256, true, TAIL_PREP, true, 1, false) \
    X(256, true, TAIL_PREP_MFMA, true, 1, false) X(256, true, TAIL_OBS, true, 1, false) \
    X(512, true, TAIL_NONE, true, 1, false) X(512, true, TAIL_PREP, true, 1, false) \
    X(512, true, TAIL_PREP_MFMA, true, 1, false) X(512, true, TAIL_OBS, true, 1, false) \
    X(256, true, TAIL_NONE, true, 2, false) X(256, true, TAIL_PREP, true, 2, false) \
    X(256, true, TAIL_PREP_MFMA, true, 2, false) X(256, true, TAIL_OBS, true, 2, false) \
    X(512, true, TAIL_NONE, true, 2, false) X(512, true, TAIL_PREP, true, 2, false) \
    X(512, true, TAIL_PREP_MFMA, true, 2, false) X(512, true, TAIL_OBS, true, 2, false)
#define DEMC_K1_STREAM_INSTANCES(X) \
    X(256, true, TAIL_PREP, true, 0, true) X(256, true, TAIL_PREP_MFMA, true, 0, true) \
    X(256, true, TAIL_PREP, true, 1, true) X(256, true, TAIL_PREP_MFMA, true, 1, true) \
    X(256, true, TAIL_PREP, true, 2, true) X(256, true, TAIL_PREP_MFMA, true, 2, true) \
    X(512, true, TAIL_PREP, true, 0, true) X(512, true, TAIL_PREP_MFMA, true, 0, true) \
    X(512, true, TAIL_PREP, true, 1, true) X(512, true, TAIL_PREP_MFMA, true, 1, true) \
    X(512, true, TAIL_PREP, true, 2, true) X(512, true, TAIL_PREP_MFMA, true, 2, true)
#ifdef DEMC_K1_EXTERN
#define DEMC_X_(...) extern template __global__ void k_propose<__VA_ARGS__>(KParams);
DEMC_K1_PHASE_INSTANCES(DEMC_X_)
DEMC_K1_RES_INSTANCES(DEMC_X_)
DEMC_K1_STREAM_INSTANCES(DEMC_X_)
#undef DEMC_X_
#endif

// ------------------------------------------------------------------------------------------------
// K2 (MvNormal): streaming cross term on the FP64 matrix cores.
//   S[p] = sum_i sum_k Y[p][k] * X[i][k]
// v_mfma_f64_16x16x4_f64: A = 16 particles x 4 dims (lane l: row l&15, k l>>4), B = 4 dims x 16
// observations (lane l: k l>>4, col l&15), C/D 16x16, 4 doubles per lane (row (l>>4)+4r, col l&15).
// A wave owns MT*16 particles whose Y fragments stay in registers for the whole pass; observations
// stream through in tiles of 16 from a fragment-ordered copy of the data (Xf[tile][kstep][64 lanes],
// one coalesced 512-byte load per k-step) and every tile accumulates into the SAME MT accumulators,
// so the sum over observations happens inside the MFMA accumulate; the 16 columns are folded with
// four shuffles at the end.  blockIdx -> (chunk = b % n_chunks, particle tile = b / n_chunks):
// workgroups that share an XCD (b % 8) stream the same chunk(s) of X, so a chunk stays in that XCD's L2.
// ------------------------------------------------------------------------------------------------
template <int KS, int MT>
__global__ __launch_bounds__(256, 2) void k_cross_mfma(KParams p, const double* __restrict__ Ypad, int dpad, int k0,
                                                       const double* __restrict__ Xf, int n_tiles, int n_chunks,
                                                       int part0) {
    // Xf holds n_tiles+1 tiles of (dpad/4) k-steps x 64 lanes; tile n_tiles is all zero and absorbs the odd tail.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = blockIdx.x % n_chunks;
    const int ptile = blockIdx.x / n_chunks;
    const int n_prop = p.n_groups * p.n_act;
    const int q0 = (ptile * 4 + wave) * (MT * 16);
    const int ksx = dpad >> 2;  // k-steps per tile in Xf (host pads so that k0/4 + KS <= ksx)

    double a[MT][KS];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int q = q0 + mt * 16 + (lane & 15);
        const bool ok = q < n_prop;
        const size_t slot = ok ? (size_t)slot_of(p, q) : 0;
        const double* yrow = Ypad + slot * dpad + k0 + (lane >> 4);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const double v = yrow[4 * ks];
            a[mt][ks] = ok ? v : 0.0;
        }
    }
    d4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (d4){0.0, 0.0, 0.0, 0.0};

    const int per = (n_tiles + n_chunks - 1) / n_chunks;
    const int t0 = chunk * per, t1 = (t0 + per < n_tiles) ? t0 + per : n_tiles;
    const size_t tstride = (size_t)ksx * 64;
    const double* xb = Xf + (size_t)(k0 >> 2) * 64 + lane;
    double b0[KS], b1[KS];
    {
        const double* x = xb + (size_t)(t0 < t1 ? t0 : n_tiles) * tstride;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) b0[ks] = x[ks * 64];
    }
    // two register sets in ping-pong: the loads of tile t+1 are in flight while tile t feeds the matrix core
    for (int t = t0; t < t1; t += 2) {
        {
            const double* x = xb + (size_t)(t + 1 < t1 ? t + 1 : n_tiles) * tstride;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) b1[ks] = x[ks * 64];
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt][ks], b0[ks], acc[mt], 0, 0, 0);
        {
            const double* x = xb + (size_t)(t + 2 < t1 ? t + 2 : n_tiles) * tstride;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) b0[ks] = x[ks * 64];
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt][ks], b1[ks], acc[mt], 0, 0, 0);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double v = acc[mt][r];
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 8);
            const int q = q0 + mt * 16 + (lane >> 4) + 4 * r;
            if ((lane & 15) == 0 && q < n_prop) p.partial[(size_t)(part0 + chunk) * p.P + slot_of(p, q)] = v;
        }
}

// ------------------------------------------------------------------------------------------------
// K2 (MvNormal, DIRECT mode): the residual form the reference's loglike implies (test/multivariate_normal_tests.jl:31-33:
// logpdf(MvNormal(mu, Sigma), x_i) summed over i) with the whitening done once:  sum_i |z_i - m|^2,  z_i = L^-1 (x_i - xbar)
// (demc_set_model),  m = L^-1 (theta' - xbar) (K1's preparation, on the matrix cores).  Unlike the expanded form this one
// does not separate into data-only and proposal-only factors: every (proposal, observation) pair costs 3 d flop on the
// FP64 vector pipe (SURVEY 8d's count), nothing collapses.  Thread per proposal with m in registers; all lanes of a wave
// visit the same observation, so a z row is one wave-uniform (scalar) load and enters the VALU as an SGPR operand; per
// dimension one v_add_f64 and one v_fma_f64.  DP = padded row length (host pads z rows and m with zeros).
// grid = (proposal blocks of 256, observation chunks).
// ------------------------------------------------------------------------------------------------
// `clk` (demc_timing_enable only, else null: a wave-uniform branch on a kernarg, AFTER the loop): every workgroup leaves the
// shader-clock counter (s_memtime), the 100 MHz reference counter (s_memrealtime) and the CU it ran on as it ENDS, so that the host
// can say which clock the vector pipe held under THIS kernel's load (demc_timing_clock: counter differences between the first and
// the last workgroup to finish on the SAME CU -- the counters of different CUs are not aligned) -- the chip lowers its clock under a dense FP64 loop and devices differ
// (MI355X_MICROARCH.md, "DVFS give-back" items 5 and 6), and a VALU-bound rate scales with it.  No stamp BEFORE the loop: the
// compiler treats s_memtime as a memory clobber, and a clobber ahead of the loop turns the wave-uniform z-row loads from scalar
// loads (s_load_dwordx16, SGPR operands) into vector loads -- the first form of this probe cost the kernel a factor of 3.3.
template <int DP>
__global__ __launch_bounds__(256) void k_direct_mvn(KParams p, int n_chunks, unsigned long long* __restrict__ clk) {
    // workgroup -> (block of 256 proposals, observation chunk).  Workgroups are dealt to the eight XCDs in turn (linear id mod 8) and
    // every XCD has an L2 of its own: with the chunk count a multiple of 8, XCD x takes the chunks x, x + 8, ... and all proposal
    // blocks of one chunk before the next, so a chunk of whitened rows is fetched by ONE XCD, once (k_cross_mfma's rule).  With
    // (block, chunk) = (blockIdx.x, blockIdx.y) every XCD held a share of twelve chunks at a time -- 6.4 MB of rows behind a 4 MB
    // L2 -- and the counters saw the data sixteen times per launch (443 MB where the rows are 25.6).
    int pblock = blockIdx.x, chunk = blockIdx.y;
    if ((n_chunks & 7) == 0) {
        const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x, xcd = lin & 7u, j = lin >> 3;
        chunk = (int)((j / gridDim.x) * 8u + xcd);
        pblock = (int)(j % gridDim.x);
    }
    const int q = pblock * 256 + threadIdx.x;
    const int n_prop = p.n_groups * p.n_act;
    const bool ok = q < n_prop;
    const size_t slot = ok ? (size_t)slot_of(p, q) : 0;
    double m[DP];
    const double* mrow = p.Ypad + slot * p.dpad;
#pragma unroll
    for (int k = 0; k < DP; ++k) m[k] = (ok && k < p.d) ? mrow[k] : 0.0;
    const long long per = (p.N + n_chunks - 1) / n_chunks;
    const long long i0 = chunk * per, i1 = (i0 + per < p.N) ? i0 + per : p.N;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};  // four independent chains per lane
    const double* z = p.data + i0 * DP;
    for (long long i = i0; i < i1; ++i, z += DP) {
#pragma unroll
        for (int k = 0; k < DP; k += 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double r = z[k + e] - m[k + e];
                acc[e] = fma(r, r, acc[e]);
            }
        }
    }
    // (The four s_load_dwordx16 of a row sit at the top of the trip with s_waitcnt lgkmcnt(0) right behind them: the wait is exposed
    // per wave and six waves per SIMD cover most of it.  A half-row pipeline written in the source -- each half of the NEXT row asked
    // for as soon as the current half is consumed -- is folded back by the compiler into exactly this shape; profiles/r06/NOTES.md.)
    if (ok) p.partial[(size_t)chunk * p.P + slot] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    if (clk) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u;  // hwreg(HW_REG_XCC_ID, 0, 4)
        const unsigned hwid = __builtin_amdgcn_s_getreg(4 | (31 << 11));       // hwreg(HW_REG_HW_ID): cu_id [11:8], sh_id [12], se_id [15:13]
        if (threadIdx.x == 0) {
            const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
            clk[3 * wg] = t1;
            clk[3 * wg + 1] = r1;
            clk[3 * wg + 2] = (xcc << 8) | ((hwid >> 8) & 0xffu);  // the CU the workgroup ran on
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K2 (scalar-data families): thread-per-proposal streaming likelihoods.  All lanes of a wave visit the same observation,
// so data loads are wave-uniform (scalar) and the loop is pure FP64 VALU.  grid = (proposal blocks, chunks).
// ------------------------------------------------------------------------------------------------
#ifndef DEMC_DEVICE_HELPERS_ONLY  // (a translation unit that only wants the device helpers: demc_longrow.cpp)
__global__ __launch_bounds__(256) void k_obs_loglike(KParams p, int n_chunks) {
    // LDS copy of the family's table: Phi / phi polynomials (LBA), log Phi(-z) polynomials (LNR)
    constexpr int kTabPhi = kPhiIntervals * kPhiRow, kTabErfcx = kLogPhiRows * kLogPhiRow;  // (LNR: the log Phi(-z) table)
    __shared__ double s_tab[kTabPhi > kTabErfcx ? kTabPhi : kTabErfcx];
    if (p.family == FAM_LBA) {
        for (int i = threadIdx.x; i < kTabPhi; i += 256) s_tab[i] = kPhiTable[i];
        __syncthreads();
    } else if (p.family == FAM_LNR) {
        for (int i = threadIdx.x; i < kTabErfcx; i += 256) s_tab[i] = kLogPhiTable[i];
        __syncthreads();
    }
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int n_prop = p.n_groups * p.n_act;
    const int chunk = blockIdx.y;
    if (q >= n_prop) return;
    const size_t slot = (size_t)slot_of(p, q);
    const double* th = p.prop + slot * p.D;
    const long long per = (p.N + n_chunks - 1) / n_chunks;
    const long long i0 = chunk * per, i1 = (i0 + per < p.N) ? i0 + per : p.N;
    double acc;
    if (p.family == FAM_LBA)  // Run_LBA.jl:33-37
        acc = p.n_acc == 3 ? lba_range_sum<3>(p, th, i0, i1, 1, s_tab)
              : p.n_acc == 2 ? lba_range_sum<2>(p, th, i0, i1, 1, s_tab) : lba_range_sum<0>(p, th, i0, i1, 1, s_tab);
    else if (p.family == FAM_LNR)
        acc = lnr_range_sum(p, th, i0, i1, 1, s_tab);
    else
        acc = obs_range_sum(p, th, i0, i1, 1);
    p.partial[(size_t)chunk * p.P + slot] = acc;
}
#endif

// ------------------------------------------------------------------------------------------------
// K2 (LBA, Examples/Run_LBA.jl:33-37): a WAVE per proposal, lanes across trials (round 5).
// k_obs_loglike's lanes are proposals at one trial: in a population that has not converged every lane reads another row of the
// Phi table and the LDS pipe, not the vector pipe, sets the pace (LDS busy 0.88, VALU 0.62 on cfg5's prior-drawn row).  Here the
// lanes of a wave are 64 TRIALS of one proposal, and demc_set_model has sorted the trials by (choice, decision time): the table
// argument S (b / t - nu_a) moves by a fraction of a row across the wave, whatever the proposal -- the reads are broadcasts -- and
// the winner is the same accumulator in all lanes but at the three borders of the sort.  The proposal's parameters are
// wave-uniform (scalar loads).  Per (proposal, chunk): trials i0 + lane, + 64, ...; a batch of kLbaBatch trials per lane
// contributes one log of the product of its floored densities, as in lba_range_sum; the lanes' sums are added in a fixed order.
// Same arithmetic per (trial, proposal) as lba_trial; another order of the sum over trials (the LBA's log-likelihoods are
// compared with the oracle at 1e-5).
// ------------------------------------------------------------------------------------------------
#ifndef DEMC_DEVICE_HELPERS_ONLY
template <int NA>
__global__ __launch_bounds__(256, 2) void k_lba_wave(KParams p, int n_chunks, unsigned long long* __restrict__ clk) {
    __shared__ double s_tab[kPhiIntervals * kPhiRow];
    for (int i = threadIdx.x; i < kPhiIntervals * kPhiRow; i += 256) s_tab[i] = kPhiTable[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = blockIdx.x * 4 + wave;
    const int n_prop = p.n_groups * p.n_act;
    const int chunk = blockIdx.y;
    if (q >= n_prop) return;
    const size_t slot = (size_t)slot_of(p, q);
    const double* th = p.prop + slot * p.D;  // wave-uniform
    const int na = NA > 0 ? NA : p.n_acc;
    double nu[8], nuS[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        nu[a] = a < na ? th[a] : 0.0;
        nuS[a] = kPhiS * nu[a];
    }
    const double A = th[na], kk = th[na + 1], tau = th[na + 2], b = A + kk, inv_A = 1.0 / A;
    const double kS = kPhiS * kk, bS = kPhiS * b, inv_SA = inv_A * (1.0 / kPhiS);
    double pneg = 1.0;
#pragma unroll
    for (int a = 0; a < (NA > 0 ? NA : 8); ++a)
        if (a < na) {
            double qq, Ph;
            phiS_Phi_table(s_tab, -nuS[a], qq, Ph);
            pneg *= Ph;
        }
    const double inv_norm = 1.0 / (1.0 - pneg);
    // chunks of whole batches (64 lanes x kLbaBatch trials), the last one ragged
    constexpr long long kStep = 64LL * kLbaBatch;
    const long long n_steps = (p.N + kStep - 1) / kStep, per = (n_steps + n_chunks - 1) / n_chunks;
    const long long i0 = (long long)chunk * per * kStep, i1r = i0 + per * kStep, i1 = i1r < p.N ? i1r : p.N;
    double acc = 0.0;
    long long base = i0;
    for (; base + kStep <= i1; base += kStep) {  // whole batches: every lane has its kLbaBatch trials, no predicate in the body
        double cc[kLbaBatch], rr[kLbaBatch];
#pragma unroll
        for (int j = 0; j < kLbaBatch; ++j) {
            cc[j] = p.data[base + 64LL * j + lane];
            rr[j] = p.data2[base + 64LL * j + lane];
        }
        double prod = 1.0;
        // (both factors of lba_factor and a select per lane: a batch-level test for ONE winner in all 512 trials -- all but three
        // batches of a proposal -- with the scalar-branch form behind it measured 2 % SLOWER, profiles/r05/NOTES.md section 10)
#pragma unroll
        for (int j = 0; j < kLbaBatch; ++j)
            prod *= lba_trial<NA, kPhiRow, double>(s_tab, na, nu, nuS, kS, bS, tau, inv_A, inv_SA, inv_norm, cc[j], rr[j]);
        if (__builtin_expect(!(prod < 1e300), 0)) {  // (see lba_range_sum: the batch again, a log per trial)
            double s2 = 0.0;
#pragma unroll 1
            for (int j = 0; j < kLbaBatch; ++j)
                s2 += log(lba_trial<NA, kPhiRow, double>(s_tab, na, nu, nuS, kS, bS, tau, inv_A, inv_SA, inv_norm, cc[j], rr[j]));
            acc += s2;
        } else
            acc += log(prod);
    }
#pragma unroll 1
    for (long long i = base + lane; i < i1; i += 64)  // the ragged end of the data (fewer than 512 trials): a log per trial
        acc += log(lba_trial<NA, kPhiRow, double>(s_tab, na, nu, nuS, kS, bS, tau, inv_A, inv_SA, inv_norm, p.data[i], p.data2[i]));
    acc = subgroup_sum(acc, 64);
    if (lane == 0) p.partial[(size_t)chunk * p.P + slot] = acc;
    if (clk) {  // (demc_timing_clock: as k_direct_mvn -- end-of-life stamps only, behind every load of the kernel)
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u, hwid = __builtin_amdgcn_s_getreg(4 | (31 << 11));
        if (threadIdx.x == 0) {
            const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
            clk[3 * wg] = t1;
            clk[3 * wg + 1] = r1;
            clk[3 * wg + 2] = (xcc << 8) | ((hwid >> 8) & 0xffu);
        }
    }
}
#endif

#ifdef DEMC_EXPERIMENTS
// ------------------------------------------------------------------------------------------------
// K2 (LBA, Examples/Run_LBA.jl:33-37) with EIGHT shifted copies of the table -- an A/B EXPERIMENT (make EXPERIMENTS=1,
// DEMC_LBA_WIDE=1), not the product path.  k_obs_loglike keeps one packed copy of the Phi polynomials in LDS (80-byte rows);
// the lanes of a wave are proposals at the same trial, a population that has not converged reads a different row in almost
// every lane, and the LDS pipe -- the kernel's limiting pipe, busy 0.80 -- spent 56 % of its active cycles on bank conflicts
// (profiles/r03).  Here rows are padded to 128 bytes and copy j is shifted by 16 j bytes; lane l reads copy l & 7, the row
// address is a shift.  Measured (cfg5, 20 steps, twice each): 1.584 ms per launch against 1.522 for the packed copy -- 4 %
// SLOWER.  Why it cannot win: a ds_read_b128 is served in four groups of SIXTEEN lanes over 64 banks
// (MI355X_MICROARCH.md, LDS), so a conflict-free gather needs sixteen independently shifted slots per group -- sixteen copies
// of 256-byte-stride rows (560 KB) -- while eight copies of 128-byte rows leave two lanes per slot that collide whenever their
// rows have the same parity, and the 140 KB cost a third of the occupancy (one workgroup of 512 per CU instead of three of
// 256).  Same arithmetic per evaluation as k_obs_loglike (lba_range_sum), same partial sums per chunk: same bits.
// ------------------------------------------------------------------------------------------------
constexpr int kLbaCopyDoubles = kPhiIntervals * 16 + 2;  // 137 rows of 128 B + the 16-byte shift to the next copy
constexpr size_t kLbaTableBytes = (size_t)8 * kLbaCopyDoubles * sizeof(double);
template <int WG>
__global__ __launch_bounds__(WG) void k_lba_loglike(KParams p, int n_chunks) {
    extern __shared__ __attribute__((aligned(16))) double s_lba[];  // [8][137][16] (+ 2 doubles per copy); 16-byte reads
    for (int i = threadIdx.x; i < 8 * kPhiIntervals * kPhiRow; i += WG) {
        const int c = i / (kPhiIntervals * kPhiRow), e = i - c * (kPhiIntervals * kPhiRow);
        const int r = e / kPhiRow, k = e - r * kPhiRow;
        s_lba[c * kLbaCopyDoubles + r * 16 + k] = kPhiTable[e];
    }
    __syncthreads();
    const int q = blockIdx.x * WG + threadIdx.x;
    const int n_prop = p.n_groups * p.n_act;
    const int chunk = blockIdx.y;
    if (q >= n_prop) return;
    const size_t slot = (size_t)slot_of(p, q);
    const double* th = p.prop + slot * p.D;
    const long long per = (p.N + n_chunks - 1) / n_chunks;
    const long long i0 = chunk * per, i1 = (i0 + per < p.N) ? i0 + per : p.N;
    const double* tab = s_lba + (threadIdx.x & 7) * kLbaCopyDoubles;  // the lane's own copy
    const double acc = p.n_acc == 3 ? lba_range_sum<3, 16>(p, th, i0, i1, 1, tab)
                       : p.n_acc == 2 ? lba_range_sum<2, 16>(p, th, i0, i1, 1, tab) : lba_range_sum<0, 16>(p, th, i0, i1, 1, tab);
    p.partial[(size_t)chunk * p.P + slot] = acc;
}
#endif  // DEMC_EXPERIMENTS

// ------------------------------------------------------------------------------------------------
// K2 (hierarchical families): cost O(D) per proposal: one workgroup per proposal, lanes across subjects,
// coalesced reads of the proposal row, wave reduction (shuffles) then a 4-way LDS combine.
// ------------------------------------------------------------------------------------------------
#ifndef DEMC_DEVICE_HELPERS_ONLY  // (a translation unit that only wants the device helpers: demc_longrow.cpp)
__global__ __launch_bounds__(256) void k_hier_loglike(KParams p) {
    __shared__ double s_part[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = blockIdx.x;  // one workgroup per proposal: 256 lanes across the subjects
    const size_t slot = (size_t)slot_of(p, q);
    const double* th = p.prop + slot * p.D;
    const long long S = p.N;
    double acc = 0.0;
    if (p.family == FAM_HIER_BINOMIAL) {  // k_s ~ Binomial(n, logistic(mu_b0 + b0_s))  (BASELINE cfg4)
        const double mu0 = th[0], n = p.c0;
        const double* lgc = p.data + S;
        for (long long s = tid; s < S; s += 256) {
            const double eta = mu0 + th[2 + s], k = p.data[s];
            acc += lgc[s] - n * softplus_fast(-eta) - (n - k) * eta;  // softplus(eta) = eta + softplus(-eta)
        }
    } else {  // FAM_HIER_GAUSSIAN  Hierarchical_Example.jl:36-44
        const double mu0 = th[0], sg = th[2 + S];
        const int n = p.d;
        const double lsg = log(sg), isg = 1.0 / sg;
        for (long long s = tid; s < S; s += 256) {
            const double mu = mu0 + th[2 + s];
            double l = 0.0;
            for (int i = 0; i < n; ++i) {
                const double z = (p.data[s * n + i] - mu) * isg;
                l += -(z * z + kLog2Pi) / 2.0 - lsg;
            }
            acc += l;
        }
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) s_part[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) p.partial[slot] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}
#endif

// ------------------------------------------------------------------------------------------------
// K3: finalise the log-likelihood from the partial sums (fixed order -> deterministic), add the prior,
// Metropolis accept with one uniform per particle, then move the accepted row into theta and write
// the history row -- one pass over the particle's D scalars, LPP lanes per particle.
// ------------------------------------------------------------------------------------------------
// Sum of the per-chunk partial sums of one proposal, in a fixed (deterministic) order: every lane of the sub-group adds
// the chunks c = sl, sl+lpp, ... in increasing order, then the lanes are combined by the DPP tree -- the chunk loads of
// the sub-group go out together instead of one dependent load after another on a single lane.
__device__ inline double sum_partials(const KParams& p, size_t slot, int sl, int lpp) {
    double s = 0.0;
    if (lpp > 64) {  // one particle per workgroup (very large D): few chunks, lane 0 adds them
        for (int c = 0; c < p.n_partials; ++c) s += p.partial[(size_t)c * p.P + slot];
        return s;
    }
    for (int c = sl; c < p.n_partials; c += lpp) s += p.partial[(size_t)c * p.P + slot];
    return subgroup_sum(s, lpp);
}
__device__ inline double finalize_loglike(const KParams& p, size_t slot, double s) {
    double sg = 0.0;
    if (p.family == FAM_MVN_ISO) sg = p.prop[slot * p.D + p.d];
    if (p.family == FAM_GAUSSIAN) sg = p.prop[slot * p.D + 1];
    return loglike_from_stats(p, s, p.aux[slot], sg);
}

#ifndef DEMC_DEVICE_HELPERS_ONLY  // (a translation unit that only wants the device helpers: demc_longrow.cpp)
__global__ __launch_bounds__(256) void k_accept_store(KParams p) {
    const int tid = threadIdx.x;
    const int lpp = p.lpp3, ppp = 256 / lpp;
    const int sub = tid / lpp, sl = tid % lpp;
    const int n_prop = p.n_groups * p.n_act;
    const int q = blockIdx.x * ppp + sub;
    const bool valid = q < n_prop;
    int g = 0, pl = 0;
    const size_t slot = valid ? (size_t)slot_of(p, q, g, pl) : 0;
    const int D = p.D;
    int acc = 0;
    double w_new = 0.0;
    const double s_sum = sum_partials(p, slot, sl, lpp);
    if (sl == 0 && valid) {
        const double w = p.weight[slot];
        const bool oob = p.prop_oob[slot] != 0;
        double wp;
        if (p.fitness_kind == 1)  // evaluate_fun! utilities.jl:113-120
            wp = oob ? (p.update_kind == 1 ? -INFINITY : INFINITY) : finalize_loglike(p, slot, s_sum);
        else  // compute_posterior! utilities.jl:92-99
            wp = oob ? -INFINITY : p.prop_prior[slot] + finalize_loglike(p, slot, s_sum);
        const uint32_t eslot = (uint32_t)(p.group_offset + g) * (uint32_t)p.Np + (uint32_t)pl;
        const U4 ra = draw_block(p.seed, S_PART, p.sweep, (uint64_t)p.iter, eslot, 3);
        const double u_acc = replayed(p.rp_part, slot * 5 + 4, u53(ra.x, ra.y));
        acc = decide(p, u_acc, wp, w, p.prop_adj[slot]);  // mh_update! / maximize! / minimize!
        w_new = acc ? wp : w;
        if (acc) p.weight[slot] = wp;
        if (p.trace) {
            p.tr_w[slot] = wp;
            p.tr_acc[slot] = (unsigned char)acc;
        }
        if (p.store_row >= 0) {
            const size_t hrow = (size_t)p.store_row * p.P + slot;
            if (p.update_kind == 0 && p.mode == MODE_STEP) {  // utilities.jl:207-208
                p.acc_hist[hrow] = (unsigned char)acc;
                p.lp_hist[hrow] = w_new;
            }
            p.id_hist[hrow] = (int)p.id[slot];
        }
    }
    if (lpp > 64) {  // one particle per workgroup: hand the decision over through LDS
        __shared__ int s_acc;
        if (tid == 0) s_acc = acc;
        __syncthreads();
        acc = s_acc;
    } else
        acc = (int)subgroup_bcast<0>((uint32_t)acc, lpp, (tid & 63) & ~(lpp - 1));
    if (!valid) return;
    double* trow = p.theta + slot * D;
    const double* prow = p.prop + slot * D;
    double* hrow = (p.store_row >= 0) ? p.hist + ((size_t)p.store_row * p.P + slot) * p.hist_ld : nullptr;
    if (!acc && !hrow) return;
    for (int k = sl; 2 * k < D; k += lpp)
        for (int e = 0; e < 2; ++e) {
            const int j = 2 * k + e;
            if (j < D) {
                double v;
                if (acc) {
                    v = prow[j];
                    trow[j] = v;  // current.theta = proposal.theta utilities.jl:204
                } else
                    v = trow[j];
                if (hrow) hrow[j] = v;  // samples[iter, :, id] = theta utilities.jl:170-180
            }
        }
}
#endif

// history row for particles that were NOT active in the storing phase is never needed: every particle is
// active exactly once per sweep, and the store happens in the phase that updates it.

// ------------------------------------------------------------------------------------------------
// Migration (migration.jl:11-91).  pack: one workgroup per local group picks its candidate
// (select_particle: P(j) ~ exp(-(w_j - min w)); where the reference's exp.(-w)/sum would hold a NaN -- a weight of -Inf
// or NaN, or every weight +Inf -- argmin, as its findmin fallback; a single +Inf weight is just probability 0) and writes
// the row (slot, theta[D], weight, id).  apply: every workgroup recomputes the group subset from the
// shared Philox STEP stream (select_groups) and the i-th selected local group receives the candidate
// of the (i-1)-th (circshift(particles, 1)).
// ------------------------------------------------------------------------------------------------
#ifndef DEMC_DEVICE_HELPERS_ONLY  // (a translation unit that only wants the device helpers: demc_longrow.cpp)
__global__ __launch_bounds__(256) void k_mig_pack(KParams p, double* __restrict__ rows) {
    extern __shared__ double lds[];  // [Np] cumulative weights + [ceil(Np/16)] chunk totals
    __shared__ double s_red[4];
    __shared__ int s_redi[4];
    __shared__ int s_bad, s_pick;
    const int tid = threadIdx.x, g = blockIdx.x, Np = p.Np, D = p.D;
    const double* gw = p.weight + (size_t)g * Np;
    double* cdf = lds;
    double* ctot = lds + Np;
    // findmin(w) (first index of the minimum) and "any non-finite weight"
    double m = INFINITY;
    int am = 0x7fffffff, bad = 0;
    if (tid == 0) s_bad = 0;
    for (int i = tid; i < Np; i += 256) {
        const double w = gw[i];
        if (!(w > -INFINITY)) bad = 1;  // -Inf or NaN: exp(-w)/sum is NaN in the reference (migration.jl:66-69)
        if (w < m) { m = w; am = i; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double m2 = __shfl_xor(m, o);
        const int a2 = __shfl_xor(am, o);
        if (m2 < m || (m2 == m && a2 < am)) { m = m2; am = a2; }
    }
    __syncthreads();
    if (bad) s_bad = 1;
    if ((tid & 63) == 0) { s_red[tid >> 6] = m; s_redi[tid >> 6] = am; }
    __syncthreads();
    for (int k = 0; k < 4; ++k) {
        const double m2 = s_red[k];
        const int a2 = s_redi[k];
        if (m2 < m || (m2 == m && a2 < am)) { m = m2; am = a2; }
    }
    if (s_bad || !(m < INFINITY)) {  // (all weights +Inf: 0/0 there too)
        if (tid == 0) s_pick = am < Np ? am : 0;
    } else {
        // P(j) ~ exp(-(w_j - min w)); cumulative weights in the fixed three-level order shared with the oracle (chunk16_prefix)
        const int n_chunk = (Np + 15) >> 4;
        for (int i = tid; i < Np; i += 256) cdf[i] = exp(m - gw[i]);
        __syncthreads();
        for (int c = tid; c < n_chunk; c += 256) {
            double v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = (c * 16 + k < Np) ? cdf[c * 16 + k] : 0.0;
            const double pre = chunk16_prefix(v);  // the fixed order shared with the oracle (select_particle_stable)
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (c * 16 + k < Np) cdf[c * 16 + k] = v[k];
            ctot[c] = pre;
        }
        __syncthreads();
        if (tid == 0) {
            double off = 0.0;
            for (int c = 0; c < n_chunk; ++c) {
                const double t = ctot[c];
                ctot[c] = off;
                off = off + t;
            }
        }
        __syncthreads();
        for (int i = tid; i < Np; i += 256) cdf[i] = ctot[i >> 4] + cdf[i];
        __syncthreads();
        const U4 r = draw_block(p.seed, S_MIG, 0, (uint64_t)p.iter, (uint32_t)(p.group_offset + g), 0);
        const double t = u53(r.x, r.y) * cdf[Np - 1];
        int cnt = 0;  // first i with cdf[i] >= t == number of entries below t (monotone)
        for (int i = tid; i < Np; i += 256) cnt += (cdf[i] < t) ? 1 : 0;
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        __syncthreads();
        if ((tid & 63) == 0) s_redi[tid >> 6] = cnt;
        __syncthreads();
        if (tid == 0) {
            const int c = s_redi[0] + s_redi[1] + s_redi[2] + s_redi[3];
            s_pick = c < Np ? c : Np - 1;
        }
    }
    __syncthreads();
    int j = s_pick;
    if (p.rp_mig_particle && p.rp_mig_particle[g] >= 0) j = (int)p.rp_mig_particle[g];  // replayed select_particle
    const size_t slot = (size_t)g * Np + j;
    double* o = rows + (size_t)g * (D + 3);
    if (tid == 0) {
        o[0] = (double)j;
        o[D + 1] = p.weight[slot];
        o[D + 2] = (double)p.id[slot];
    }
    for (int k = tid; k < D; k += 256) o[1 + k] = p.theta[slot * D + k];
}
#endif

// demc_apply_migration: slot moves planned by the host.  Two launches: gather the source rows into a staging buffer
// ([n][D+2]: theta, weight, id), then scatter them to the destination slots -- reads complete before any write.
#ifndef DEMC_DEVICE_HELPERS_ONLY  // (a translation unit that only wants the device helpers: demc_longrow.cpp)
__global__ __launch_bounds__(256) void k_slot_moves(KParams p, const int* __restrict__ slots, double* __restrict__ stage, int n,
                                                    int scatter) {
    const int D = p.D, W = D + 2;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)n * W) return;
    const int k = (int)(e / W), j = (int)(e % W);
    const size_t slot = (size_t)slots[k];
    if (!scatter)
        stage[e] = j < D ? p.theta[slot * D + j] : j == D ? p.weight[slot] : __longlong_as_double(p.id[slot]);
    else if (j < D)
        p.theta[slot * D + j] = stage[e];
    else if (j == D)
        p.weight[slot] = stage[e];
    else
        p.id[slot] = __double_as_longlong(stage[e]);
}
#endif

#ifndef DEMC_DEVICE_HELPERS_ONLY  // (a translation unit that only wants the device helpers: demc_longrow.cpp)
__global__ __launch_bounds__(256) void k_mig_apply(KParams p, const double* __restrict__ all_rows, int n_groups_total) {
    extern __shared__ int perm[];  // [n_groups_total] the shuffle, then [n_groups_total] its random words
    __shared__ int s_ns;
    const int tid = threadIdx.x, D = p.D;
    const int ng = n_groups_total;
    uint32_t* words = reinterpret_cast<uint32_t*>(perm + ng);
    // select_groups migration.jl:31-35.  The swaps of the partial shuffle depend on each other, their random words do
    // not: all lanes draw the Philox blocks (word i = word i&3 of block 1 + i/4 of the STEP stream) and fill the identity
    // first, one lane then walks the swaps.
    for (int b = tid; 4 * b < ng; b += 256) {
        const U4 r = draw_block(p.seed, S_STEP, 0, (uint64_t)p.iter, 0, 1 + (uint32_t)b);
        words[4 * b] = r.x;
        if (4 * b + 1 < ng) words[4 * b + 1] = r.y;
        if (4 * b + 2 < ng) words[4 * b + 2] = r.z;
        if (4 * b + 3 < ng) words[4 * b + 3] = r.w;
    }
    for (int i = tid; i < ng; i += 256) perm[i] = i;
    __syncthreads();
    if (p.rp_n_mig > 0) {  // replayed select_groups: the caller's ordered sub-group
        for (int i = tid; i < p.rp_n_mig; i += 256) perm[i] = p.rp_mig_groups[i];
        if (tid == 0) s_ns = p.rp_n_mig;
    } else if (tid == 0) {
        const U4 r = draw_block(p.seed, S_STEP, 0, (uint64_t)p.iter, 0, 0);
        const int ns = 2 + (int)mulhi32(r.z, (uint32_t)(ng - 1));
        for (int i = 0; i < ns; ++i) {
            const int j = i + (int)mulhi32(words[i], (uint32_t)(ng - i));
            const int t = perm[i]; perm[i] = perm[j]; perm[j] = t;
        }
        s_ns = ns;
    }
    __syncthreads();
    const int ns = s_ns;
    for (int i = blockIdx.x; i < ns; i += gridDim.x) {
        const int gd = perm[i], gs = perm[(i + ns - 1) % ns];  // shift_particles! migration.jl:84-91
        const int gl = gd - p.group_offset;
        if (gl < 0 || gl >= p.n_groups) continue;
        const double* dst = all_rows + (size_t)gd * (D + 3);
        const double* src = all_rows + (size_t)gs * (D + 3);
        const size_t slot = (size_t)gl * p.Np + (size_t)dst[0];
        for (int k = tid; k < D; k += 256) p.theta[slot * D + k] = src[1 + k];
        if (tid == 0) {
            p.weight[slot] = src[D + 1];
            p.id[slot] = (long long)src[D + 2];
        }
    }
}
#endif

// History cells between their padded device layout (cells hist_ld doubles apart) and the dense [rows][P][D] layout of the C-ABI
// (demc_set_history_rows / demc_get_history): dst[c][j] = src[c][j] for the D scalars of every cell.
#ifndef DEMC_DEVICE_HELPERS_ONLY
__global__ __launch_bounds__(256) void k_hist_repack(double* __restrict__ dst, const double* __restrict__ src, long long cells, int D, int ld_dst,
                                                    int ld_src) {
    const long long total = cells * D;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const long long c = t / D;
        const int j = (int)(t - c * D);
        dst[c * ld_dst + j] = src[c * ld_src + j];
    }
}
#endif

// ------------------------------------------------------------------------------------------------
// Chains export (bundle_samples, main.jl:222-250): re-key the slot-keyed history by particle id and lay it out as the
// value array (parameters, acceptance, lp).  One thread per (row, slot, j); reads are contiguous in j.
// ------------------------------------------------------------------------------------------------
#ifndef DEMC_DEVICE_HELPERS_ONLY  // (a translation unit that only wants the device helpers: demc_longrow.cpp)
__global__ __launch_bounds__(256) void k_export_chains(KParams p, long long row0, long long n, int layout, long long id0,
                                                       double* __restrict__ out) {
    const long long D2 = p.D + 2;
    const long long total = n * p.P * D2;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const long long j = t % D2, rs = t / D2;
        const long long slot = rs % p.P, r = rs / p.P;
        const size_t hrow = (size_t)(row0 + r) * p.P + slot;
        const long long id = (long long)p.id_hist[hrow] - id0;
        const double v = j < p.D ? p.hist[hrow * p.hist_ld + j] : (j == p.D ? (double)p.acc_hist[hrow] : p.lp_hist[hrow]);
        if (layout == 0)
            out[(size_t)((id * D2 + j) * n + r)] = v;
        else
            out[(size_t)((r * D2 + j) * p.P + id)] = v;
    }
}
#endif

}  // namespace demc
