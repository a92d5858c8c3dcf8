// demc_resobs.hpp -- the resident kernel of the DEFAULT sampler on the per-observation families, written lean (round 6).
//
// The reference's own gates for this path run the default sampler -- DE(): random_gamma, no snooker, kappa = 1, no blocks, Metropolis
// (structs.jl:80-101) -- on a model with a HANDFUL of parameters and tens to hundreds of observations: test/gaussian_tests.jl:39-41
// (Normal(mu, sigma), Np = 6), test/binomial_tests.jl (one probability), test/lognormal_race_tests.jl:40-42 (LNR, four groups of 24),
// Examples/Gaussian_Example.jl (BASELINE cfg1).  The general kernel serves them with the lane geometry of its proposal stage -- at
// most one dim pair per lane, i.e. ONE lane per particle at D = 2: that lane draws the particle's six Philox blocks one after
// the other and then walks all observations itself (cfg1: 21 us per iteration, a chain of latencies on a nearly empty chip), and
// the race model (a table in LDS) does not fuse at all: K1 -> k_obs_loglike -> K3 per colour phase, six dependent launches an
// iteration.  This file is k_res_mvn's recipe (demc_resmvn.hpp) for that case and nothing else:
//   * one workgroup per group holds the group in LDS and runs both colour phases of every iteration up to the next migration;
//   * SIXTEEN lanes per particle -- one DPP row: lanes 0..3 draw the particle's four PART blocks and lanes 4..7 its NOISE blocks
//     in the same Philox pass (one block's latency instead of six), row_newbcast hands the words round; lane j proposes scalar j
//     (D <= 16), checks its bounds and adds its prior term; theta' goes through an LDS row once, and all sixteen lanes then stride
//     over the observations (cfg1: four terms a lane instead of fifty); sums over the row on the DPP network in a fixed tree;
//   * select_base's cumulative weights by wave 0 (wave_cdf, the oracle's fixed order), the base picked by a two-level count over
//     the particle's sixteen lanes (chunk ends, then inside the chunk);
//   * the LNR's log Phi(-z) table rides in LDS for the whole launch, so the race model is ONE kernel too;
//   * rare paths (Box-Muller of a mutation sweep, non-Normal priors) out of line.
// Same addressed draws and the same per-scalar arithmetic as k_propose: proposals and decisions are the general kernel's; the sums
// over observations and prior terms run in another lane order (log-densities equal to rounding).
#pragma once
#include "demc_kernels.hpp"

namespace demc {

template <int WG>
__global__ __launch_bounds__(WG, 2) void k_res_obs(KParams p) {
    constexpr int L = 16;        // lanes per particle
    constexpr int PPP = WG / L;  // particles per pass
    extern __shared__ double lds[];
    __shared__ unsigned char s_mut[1024];  // beta coin of every iteration of this launch (n_iters <= 1024)
    __shared__ DimSeg s_seg[kMaxDimSeg];   // bounds / prior table, run-length encoded
    for (int i = threadIdx.x; i < p.n_seg * (int)(sizeof(DimSeg) / sizeof(double)); i += WG)
        reinterpret_cast<double*>(s_seg)[i] = reinterpret_cast<const double*>(p.dimseg)[i];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = p.D, Np = p.Np;
    int g = blockIdx.x;
    if (p.glist) g = p.glist[g];
    const int g_glob = p.group_offset + g;
    double* grows = p.theta + (size_t)g * Np * D;
    double* gw = p.weight + (size_t)g * Np;
    const int half = Np / 2, nact_max = Np - half;
    // LDS: tile [Np][D] | weights [Np] | cdf [nact_max] | theta' rows [PPP][D] | LNR: the log Phi(-z) table
    double* tile = lds;
    double* w_s = tile + (size_t)Np * D;
    double* cdf = w_s + Np;
    double* scr = cdf + nact_max;
    double* tab = scr + (size_t)PPP * D;
    for (int i = tid; i < Np * D; i += WG) tile[i] = grows[i];
    for (int i = tid; i < Np; i += WG) w_s[i] = gw[i];
    for (int i = tid; i < p.n_iters; i += WG) {
        const U4 r = draw_block(p.seed, S_GROUP, 0, (uint64_t)(p.iter + i), (uint32_t)g_glob, 0);
        s_mut[i] = u53(r.x, r.y) <= p.beta ? 1 : 0;  // mutate_or_crossover! main.jl:199-207
    }
    if (p.family == FAM_LNR)
        for (int i = tid; i < kLogPhiRows * kLogPhiRow; i += WG) tab[i] = kLogPhiTable[i];
    // lane geometry: particle q of the pass = tid / 16, lane sl of the particle owns scalar sl
    const int qp = tid >> 4, sl = tid & 15;
    int sgj = 0;  // the table segment of the lane's scalar
    {
        const int jj = sl < D ? sl : 0;
        for (int i = 1; i < p.n_seg; ++i) sgj += (jj >= p.dimseg[i].start) ? 1 : 0;
    }
    const double eps = p.eps, eps2 = p.eps - (-p.eps);
    __syncthreads();

    const long long n_steps = (long long)p.n_iters * 2;
    for (long long step = 0; step < n_steps; ++step) {
        const int ph = (int)(step & 1);
        const int it_rel = (int)(step >> 1);
        const long long iter = p.iter + it_rel;
        const int a_lo = ph ? half : 0, n_act = ph ? Np - half : half;
        const int pool_lo = ph ? 0 : half, pool_n = ph ? half : Np - half;
        const long long store_row = (p.hist && iter - 1 < p.n_rows) ? iter - 1 : -1;
        const bool is_mut = s_mut[it_rel] != 0;
        const bool use_base = !is_mut && iter <= p.burnin;  // crossover.jl:164

        // ---- select_base's cumulative weights over the resting colour (wave 0; crossover.jl:282-289, stabilised) ----
        if (use_base) {
            if (wave == 0) {
                const double* pw = w_s + pool_lo;
                double e[4], m = -INFINITY;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    e[r] = (lane + 64 * r < pool_n) ? pw[lane + 64 * r] : -INFINITY;
                    m = fmax(m, e[r]);
                }
                m = wave_max(m);
#pragma unroll
                for (int r = 0; r < 4; ++r) e[r] = (lane + 64 * r < pool_n) ? exp(e[r] - m) : 0.0;
                if (pool_n <= 64) {
                    double e1[1] = {e[0]};
                    wave_cdf<1>(e1, pool_n);
                    e[0] = e1[0];
                } else if (pool_n <= 128) {
                    double e2[2] = {e[0], e[1]};
                    wave_cdf<2>(e2, pool_n);
                    e[0] = e2[0]; e[1] = e2[1];
                } else
                    wave_cdf<4>(e, pool_n);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (lane + 64 * r < pool_n) cdf[lane + 64 * r] = e[r];
            }
            lds_barrier();  // cdf visible
        }
        for (int q0 = 0; q0 < n_act; q0 += PPP) {  // passes of PPP particles (independent of each other: movers read the resting colour)
            const int q = q0 + qp;
            const bool valid = q < n_act;
            const int pl = a_lo + (valid ? q : 0);
            const size_t slot = (size_t)g * Np + pl;
            const uint32_t eslot = (uint32_t)g_glob * (uint32_t)Np + (uint32_t)pl;
            const int pid = (int)p.id[slot];  // (asked for here, needed at the store)
            // ---- the particle's addressed draws in ONE Philox pass: lanes 0..3 PART blocks 0..3, lanes 4..7 NOISE blocks 0..3 ----
            const U4 mine = draw_block(p.seed, sl < 4 ? S_PART : S_NOISE, 0, (uint64_t)iter, eslot, (uint32_t)(sl & 3));
            const U4 r0 = bcast_u4<0>(mine, L, 0), ri = bcast_u4<1>(mine, L, 0), rg = bcast_u4<2>(mine, L, 0), ra = bcast_u4<3>(mine, L, 0);
            const U4 n0 = bcast_u4<4>(mine, L, 0), n1 = bcast_u4<5>(mine, L, 0), n2 = bcast_u4<6>(mine, L, 0), n3 = bcast_u4<7>(mine, L, 0);
            const int nb_i = sl >> 2;
            const U4 nb = nb_i == 0 ? n0 : nb_i == 1 ? n1 : nb_i == 2 ? n2 : n3;  // the NOISE block behind the lane's scalar
            const uint32_t w_even = (sl & 2) ? nb.z : nb.x, w_odd = (sl & 2) ? nb.w : nb.y;
            const uint32_t nwj = (sl & 1) ? w_odd : w_even;
            const double u_base = u53(r0.z, r0.w), u_acc = u53(ra.x, ra.y);
            uint32_t ia = 0, ib = 0;
            pick_pair(ri.x, ri.y, (uint32_t)pool_n, ia, ib);  // two_colour: the pool is the resting half, self is not in it
            const double g1 = 0.5 + (1.0 - 0.5) * u53(rg.x, rg.y);                   // crossover.jl:162
            const double g2 = use_base ? 0.5 + (1.0 - 0.5) * u53(rg.z, rg.w) : 0.0;  // crossover.jl:164
            int ibase = 0;
            if (use_base) {
                const double total = cdf[pool_n - 1];
                if (!(total > 0.0) || !(total < INFINITY)) {
                    ibase = (int)(u_base * pool_n);
                    ibase = ibase < pool_n ? ibase : pool_n - 1;
                } else {
                    // first i with cdf[i] >= t, else last = the number of entries below t (cdf is monotone): lane sl looks at the END
                    // of chunk sl (pools of up to 256: sixteen chunks), then at entry sl of the chunk that holds t
                    const double t = u_base * total;
                    const int n_chunk = (pool_n + 15) >> 4;
                    const int last = 16 * sl + 15 < pool_n ? 16 * sl + 15 : pool_n - 1;
                    int below = (sl < n_chunk && cdf[sl < n_chunk ? last : 0] < t) ? 1 : 0;
                    below = subgroup_sum(below, L);
                    const int c0 = 16 * (below < n_chunk ? below : n_chunk - 1);
                    const int i = c0 + sl;
                    int cnt = (i < pool_n && cdf[i < pool_n ? i : pool_n - 1] < t) ? 1 : 0;
                    cnt = subgroup_sum(cnt, L);
                    ibase = c0 + cnt;
                    ibase = ibase < pool_n ? ibase : pool_n - 1;
                }
            }
            // ---- proposal of the lane's scalar, bounds, prior ----
            const int j = sl < D ? sl : 0;
            const double tj = tile[(size_t)pl * D + j];
            double v;
            if (is_mut) {  // pt + Normal(0, sigma)  mutation.jl:15-18 (Box-Muller on the scalar pair's two words)
                const double2 z = box_muller_outofline(w_even, w_odd);
                v = tj + p.sigma * ((sl & 1) ? z.y : z.x);
            } else {  // ((Pt + g1*(Pm-Pn)) + g2*(Pb-Pt)) + b  crossover.jl:168
                const double aj = tile[(size_t)(pool_lo + (int)ia) * D + j], bj = tile[(size_t)(pool_lo + (int)ib) * D + j];
                const double t1 = aj - bj;
                double t6 = tj + t1 * g1;
                if (use_base) {
                    const double t4 = tile[(size_t)(pool_lo + ibase) * D + j] - tj;
                    t6 = t6 + t4 * g2;
                }
                v = t6 + (-eps + eps2 * u32unit(nwj));
            }
            int oob = 0;
            double prior = 0.0;
            if (sl < D) {
                const DimTab* tb = &s_seg[sgj].t;
                oob = !(v >= tb->lo && v <= tb->hi);  // in_bounds utilities.jl:70-78
                if (tb->kind == PR_NORMAL) {
                    const double z = (v - tb->a) * tb->b;
                    prior = tb->c - 0.5 * (z * z);
                } else if (tb->kind != PR_FLAT)
                    prior = prior_term_outofline(tb, v);
                scr[(size_t)qp * D + sl] = v;
            }
            prior = subgroup_sum(prior, L);
            oob = subgroup_sum(oob, L);
            wave_lds_sync();  // theta' of the particle is in its LDS row (written and read inside one wave)
            // ---- model.loglike: the sixteen lanes stride over the observations (utilities.jl:92-99) ----
            const double* th = scr + (size_t)qp * D;
            double part;
            if (p.family == FAM_LNR)
                part = lnr_range_sum(p, th, sl, p.N, L, tab);
            else
                part = obs_range_sum(p, th, sl, p.N, L);
            const double S = subgroup_sum(part, L);
            const double sg = p.family == FAM_GAUSSIAN ? th[1] : 1.0;
            // ---- compute_posterior! + mh_update! + store_samples! (utilities.jl:92-99, 55-58, 201-210, 161-180) ----
            const double w = w_s[pl];
            const double wp = oob ? -INFINITY : prior + loglike_from_stats(p, S, 0.0, sg);
            const double ex = exp(wp - w);
            const int acc = (ex >= 1.0) || (u_acc <= ex);
            wave_lds_sync();  // (every lane has read theta' before the row is reused by the next pass)
            if (valid) {
                if (sl == 0) {
                    if (acc) {
                        p.weight[slot] = wp;
                        w_s[pl] = wp;
                    }
                    if (store_row >= 0) {
                        const size_t hrow = (size_t)store_row * p.P + slot;
                        p.acc_hist[hrow] = (unsigned char)acc;
                        p.lp_hist[hrow] = acc ? wp : w;
                        p.id_hist[hrow] = pid;
                    }
                }
                if (sl < D) {
                    const double x = acc ? v : tj;
                    if (acc) {
                        tile[(size_t)pl * D + sl] = x;
                        p.theta[slot * D + sl] = x;
                    }
                    if (store_row >= 0) p.hist[((size_t)store_row * p.P + slot) * p.hist_ld + sl] = x;
                }
            }
        }
        lds_barrier();  // the other colour reads what this phase wrote (rows, weights); LDS only (see k_res_mvn)
    }
}

// The instances the runtime launches (demc_resobs.cpp instantiates them, demc_hip.cpp declares them extern).
#define DEMC_RESOBS_INSTANCES(X) X(256)
#ifdef DEMC_RESOBS_EXTERN
#define DEMC_X_(...) extern template __global__ void k_res_obs<__VA_ARGS__>(KParams);
DEMC_RESOBS_INSTANCES(DEMC_X_)
#undef DEMC_X_
#endif

}  // namespace demc
