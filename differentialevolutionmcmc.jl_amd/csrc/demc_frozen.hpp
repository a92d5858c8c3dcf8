// demc_frozen.hpp -- K1 for the block sweeps of LONG rows over a whole population: the rows are streamed, nothing is parked (round 5).
//
// block_update! (main.jl:169-179) sweeps the group once per block; in a hierarchical model the blocks are the handful of
// hyper-parameters and the subjects (Examples/Hierarchical_Example.jl:88-92: [true, true, fill(false, n_subj), true] and its
// complement), and reset! (crossover.jl:336-352) puts every scalar outside the block back.  k_longrow (demc_longrow.hpp) parks
// theta' in an 80 KB LDS row per workgroup -- two workgroups, eight waves per CU -- and in-kernel stamps showed its pass, and the
// first form of this kernel's, running at the vector pipe's issue rate (profiles/r05/NOTES.md sections 8, 9).  Here ONE workgroup
// per particle streams the rows once, with no LDS row, so three or four 256-thread workgroups share a CU:
//   <256>      the block holds a few scalars (<= kFrozenMax) and the rest of a crossover proposal is frozen: one pass over the
//              particle's OWN row (one prior term and one softplus -- hier. Binomial -- or p.d residuals -- hier. Gaussian -- per
//              scalar), the block's scalars proposed by single lanes, partner rows read at those scalars only -- whole, for the
//              projections, by the particles whose snooker coin fired.  120 registers: four workgroups per CU.
//   <256,big>  the block is most of the row (hier. Binomial, even rows): the proposal of every scalar formed on the fly from own,
//              partner and base rows -- a Philox noise block of four scalars per thread and round, rows one round ahead -- and
//              formed AGAIN by an accepted particle for its stores.  155 registers: three workgroups per CU.
// Both: softplus from two small LDS tables (softplus_tab, demc_device.hpp); what is linear in a piece of the row summed raw and
// scaled once (subject() / flush()); past burn-in the history row of an iteration's last sweep written in the pass (the current
// values; an accepted proposal overwrites the block's); partners from the population or cells of the history (DE-MC_Z), the base row and
// select_base's weights from the sweep-start snapshot when the launch has one (KParams::base_theta); <256> can leave the rows it
// streamed behind as the NEXT sweep's snapshot (KParams::snap_theta).
//
// Same addressed draws and the same arithmetic per proposed scalar as k_longrow / k_propose (proposals bit for bit); the sums of
// the prior and likelihood terms run in another order and softplus_tab is not softplus_fast bit for bit (tests compare
// log-posteriors to 1e-9 / 1e-8 and every decision).  Mutation sweeps (mutate_or_crossover!, main.jl:199-207: the group's coin;
// mutation! ignores the block, main.jl:205) move the whole row: v = theta + sigma z per scalar in the same pass, and an ACCEPTED
// mutation forms the row again to store it.
//
// Taken by launch_phase (demc_hip.cpp) for: hierarchical families, rows long enough for a workgroup per particle, at least two
// workgroups' worth of moving particles per CU, a block mask, kappa = 1, pools of at most 256, no trace, no replay.  Everything
// else -- recombination, odd or unblocked rows, the hier. Gaussian subject block, one particle per CU -- stays with k_longrow.
#pragma once
#include "demc_kernels.hpp"

namespace demc {

constexpr int kFrozenMax = 8;  // scalars a block may move for this kernel

// softplus_fast (demc_device.hpp) with its 35 polynomial and range-reduction constants handed in as wave-uniform values: the
// same operations in the same order -- the same bits -- but the compiler keeps a uniform value in an SGPR pair and feeds it to
// v_fma_f64 as the addend, where a literal costs a v_mov_b64 into the accumulator in front of every v_fmac (31 of ~95 vector
// instructions per scalar in the first form of the frozen loop).
struct SoftplusC {
    double magic, log2e, ln2hi, ln2lo, e[12], a[16];
};
__device__ __forceinline__ SoftplusC softplus_consts() {
    auto u = [](double x) {
        return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
    };
    SoftplusC c;
    c.magic = u(6755399441055744.0); c.log2e = u(1.4426950408889634074);
    c.ln2hi = u(-6.93147180369123816490e-01); c.ln2lo = u(-1.90821492927058770002e-10);
    const double ef[12] = {1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0,
                           1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5};
    const double af[16] = {1.0 / 33.0, 1.0 / 31.0, 1.0 / 29.0, 1.0 / 27.0, 1.0 / 25.0, 1.0 / 23.0, 1.0 / 21.0, 1.0 / 19.0,
                           1.0 / 17.0, 1.0 / 15.0, 1.0 / 13.0, 1.0 / 11.0, 1.0 / 9.0, 1.0 / 7.0, 1.0 / 5.0, 1.0 / 3.0};
#pragma unroll
    for (int i = 0; i < 12; ++i) c.e[i] = u(ef[i]);
#pragma unroll
    for (int i = 0; i < 16; ++i) c.a[i] = u(af[i]);
    return c;
}
__device__ __forceinline__ double softplus_fast_u(double x, const SoftplusC& c) {
    // exp_nonpos(-|x|)
    const double a = fmax(-fabs(x), -700.0);
    const double tn = fma(a, c.log2e, c.magic);
    const double nf = tn - c.magic;
    const int n = __double2loint(tn);
    double r = fma(nf, c.ln2hi, a);
    r = fma(nf, c.ln2lo, r);
    double p = c.e[0];
#pragma unroll
    for (int i = 1; i < 12; ++i) p = fma(p, r, c.e[i]);
    p = fma(p, r, 1.0); p = fma(p, r, 1.0);
    const double t = p * __hiloint2double((n + 1023) << 20, 0);
    // log1p(t) = 2 atanh(t / (2 + t))
    const double den = 2.0 + t;
    double q = __builtin_amdgcn_rcp(den);
    q = fma(fma(-den, q, 1.0), q, q);
    q = fma(fma(-den, q, 1.0), q, q);
    double z = t * q;
    z = fma(fma(-den, z, t), q, z);
    const double w = z * z;
    double s = c.a[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) s = fma(s, w, c.a[i]);
    s = fma(s, w, 1.0);
    return fmax(x, 0.0) + 2.0 * z * s;
}

// MINW = waves per SIMD the register budget is cut for (3: 168 registers, no spill at two dim pairs per round; 4: 128).
// PAIRS = dim pairs of a thread per round of the frozen loop (2: four independent softplus chains; 1: two).
// BIG = the block may hold long runs (the subject block of Examples/Hierarchical_Example.jl:88-92: every scalar but two moves):
// their proposals are formed on the fly, a noise block per thread and round.  (An instance of its own: the rows in flight take
// registers the frozen loop does not need.)
#ifndef DEMC_SOFTPLUS_TAB
#define DEMC_SOFTPLUS_TAB 1  // (0: the table-free softplus_fast_u, for A/B builds -- make EXTRA=-DDEMC_SOFTPLUS_TAB=0)
#endif
template <int WG, int MINW = 3, int PAIRS = 2, bool BIG = false>
__global__ __launch_bounds__(WG, MINW) void k_frozen_sweep(KParams p) {
    __shared__ double s_red[3][WG / 64];
    __shared__ double s_snk[3];  // snooker: <Pm, Pd>, <Pn, Pd>, <Pd, Pd> over the whole row (utilities.jl:239-246)
    __shared__ int s_redi[WG / 64];
    __shared__ DimSeg s_seg[kMaxDimSeg];
    __shared__ int s_base;
    __shared__ double s_new[kFrozenMax + 2];   // theta' at the block's scalars | theta'[0] | the observation sd (hier. Gaussian)
    __shared__ double s_ref[2][kMaxDimSeg];    // 1 / theta'[ref], log theta'[ref] per table segment (Normal(a, theta[ref]) priors)
    __shared__ int s_acc;
    __shared__ __attribute__((aligned(16))) double s_sp[kSpDoubles];  // softplus_tab's tables
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = p.D, Np = p.Np;
#ifdef DEMC_STAMPS  // (diagnostic build, tools/frozen_stamps.py: shader cycles of wave 0 at the kernel's stages, 8 slots a workgroup in the
                    // trace's weight array; start and end of every workgroup on the 100 MHz clock in its adjustment array)
    const unsigned long long t0s = __builtin_amdgcn_s_memtime();
    auto stamp = [&](int i) {
        if (tid == 0 && ((long long)blockIdx.x + 1) * 8 <= p.P) p.tr_w[blockIdx.x * 8 + i] = (double)(__builtin_amdgcn_s_memtime() - t0s);
    };
    if (tid == 0 && 2 * ((long long)blockIdx.x + 1) <= p.P) p.prop_adj[2 * blockIdx.x] = (double)__builtin_amdgcn_s_memrealtime();
#else
    auto stamp = [](int) {};
#endif
    constexpr int kSegDoubles = (int)(sizeof(DimSeg) / sizeof(double));
    for (int i = tid; i < p.n_seg * kSegDoubles; i += WG) reinterpret_cast<double*>(s_seg)[i] = reinterpret_cast<const double*>(p.dimseg)[i];
#if DEMC_SOFTPLUS_TAB
    load_softplus_table(s_sp, tid, WG);  // (visible behind the barrier every workgroup passes before its pass)
#endif
    // particle of this workgroup; the particles of a group share an XCD
    const int vb = blockIdx.x;
    int g, qg;
    if ((p.n_groups & 7) == 0) {
        const int xcd = vb & 7, j = vb >> 3;
        qg = j % p.n_act;
        g = (j / p.n_act) * 8 + xcd;
    } else {
        g = vb / p.n_act;
        qg = vb % p.n_act;
    }
    if (p.glist) g = p.glist[g];
    const int pl = p.a_lo + qg;
    const size_t slot = (size_t)g * Np + pl;
    const int g_glob = p.group_offset + g;
    const uint32_t eslot = (uint32_t)g_glob * (uint32_t)Np + (uint32_t)pl;
    const double* grows = p.theta + (size_t)g * Np * D;
    const double* gw = p.weight + (size_t)g * Np;
    // (the weights select_base reads and the base row: the sweep-start snapshot when the launch has one -- KParams::base_theta:
    // DE-MC_Z inside burn-in on the synchronous schedule)
    const double* gwb = p.base_weight ? p.base_weight + (size_t)g * Np : gw;
    const double* grows_b = p.base_theta ? p.base_theta + (size_t)g * Np * D : grows;
    const double* pt = grows + (size_t)pl * D;
    const double w_cur = gw[pl];
    double* srow = (!BIG && p.snap_theta) ? p.snap_theta + slot * D : nullptr;  // the row as this sweep leaves it (by-product snapshot)
    // the history row of a sweep that is the iteration's last: the row as it stands after the decision (utilities.jl:161-180)
    double* hrow = (p.store_row >= 0) ? p.hist + ((size_t)p.store_row * p.P + slot) * p.hist_ld : nullptr;
    const bool maybe_base = p.proposal_kind == 0 && p.iter <= p.burnin && wave == 0;
    double pw_r[4] = {0.0, 0.0, 0.0, 0.0};
    if (maybe_base) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (lane + 64 * r < p.pool_n) pw_r[r] = gwb[p.pool_lo + lane + 64 * r];
    }
    // ---- per-particle scalars (k_longrow's addressing: PART blocks 0..3 by lanes 0..3, the group's block by lane 6) ----
    const bool glane = lane == 6;
    const U4 mine = draw_block(p.seed, glane ? S_GROUP : S_PART, p.sweep, (uint64_t)p.iter, glane ? (uint32_t)g_glob : eslot,
                               (uint32_t)(lane < 6 ? lane : 0));
    auto get = [&](uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); };
    const bool is_mut = u53(get(mine.x, 6), get(mine.y, 6)) <= p.beta;  // main.jl:199-207
    const double u_snk = u53(get(mine.x, 0), get(mine.y, 0)), u_base = u53(get(mine.z, 0), get(mine.w, 0));
    const uint32_t ri0 = get(mine.x, 1), ri1 = get(mine.y, 1), ri2 = get(mine.z, 1);
    const double u_g1 = u53(get(mine.x, 2), get(mine.y, 2)), u_g2 = u53(get(mine.z, 2), get(mine.w, 2));
    const double u_acc = u53(get(mine.x, 3), get(mine.y, 3));
    // 0 DE crossover, 1 snooker (theta_snooker > 0 only: with 0 the 2^-53 event of crossover.jl:31 is not taken, as in the PLAIN
    // instances), 2 mutation
    const bool snooker = !is_mut && p.theta_snooker > 0.0 && u_snk <= p.theta_snooker;
    const int kind = is_mut ? 2 : snooker ? 1 : 0;
    // Past burn-in a crossover / snooker sweep writes the CURRENT value of every scalar it reads into the history row on its way
    // (the stores ride under the pass); an accepted proposal then overwrites the block's scalars, a rejected one has nothing left
    // to do -- where the row is otherwise read a second time for the copy: 80 KB per particle at cfg4, and in 10^4 dimensions a
    // converged population rejects nearly everything (whole cfg4 from a converged start: 0.459 -> 0.421 ms per iteration).
    // Inside burn-in, where a prior-drawn population accepts every other proposal, the overwritten stores cost 2 %: the copy stays.
    const bool h_pass = hrow != nullptr && kind != 2 && p.iter > p.burnin;
    const double *Pa = pt, *Pb2 = pt, *Pc = pt, *Pbase = pt;
    double g1 = 0.0, g2 = 0.0;
    bool use_base = false;
    if (!is_mut) {
        if (p.partner_kind == 1) {  // resample (crossover.jl:113-124): distinct cells of rows 1:(iter-1) x local particles (k_longrow's draws)
            const uint64_t hd0 = ((uint64_t)get(mine.y, 4) << 32) | get(mine.x, 4), hd1 = ((uint64_t)get(mine.w, 4) << 32) | get(mine.z, 4),
                           hd2 = ((uint64_t)get(mine.y, 5) << 32) | get(mine.x, 5);
            const uint64_t ub = (uint64_t)(p.iter - 1), M = ub * (uint64_t)p.P;
            uint64_t a = mulhi64(hd0, M), b = mulhi64(hd1, M - 1), c = 0;
            if (b >= a) ++b;
            if (snooker) {
                c = mulhi64(hd2, M - 2);
                const uint64_t lo = a < b ? a : b, hi = a < b ? b : a;
                if (c >= lo) ++c;
                if (c >= hi) ++c;
            }
            Pa = p.hist + ((a % ub) * (uint64_t)p.P + a / ub) * (uint64_t)p.hist_ld;
            Pb2 = p.hist + ((b % ub) * (uint64_t)p.P + b / ub) * (uint64_t)p.hist_ld;
            Pc = p.hist + ((c % ub) * (uint64_t)p.P + c / ub) * (uint64_t)p.hist_ld;
        } else if (snooker) {
            uint32_t a, b, c;  // snooker_update! draws 3 from the whole pool (crossover.jl:241)
            pick_triple(ri0, ri1, ri2, (uint32_t)p.pool_n, a, b, c);
            Pa = grows + (size_t)((int)a + p.pool_lo) * D; Pb2 = grows + (size_t)((int)b + p.pool_lo) * D; Pc = grows + (size_t)((int)c + p.pool_lo) * D;
        } else {
            uint32_t a, b;
            if (p.exclude_self) {  // setdiff(group, [Pt]) crossover.jl:158
                pick_pair(ri0, ri1, (uint32_t)p.pool_n - 1, a, b);
                const uint32_t t = (uint32_t)(pl - p.pool_lo);
                a += (a >= t); b += (b >= t);
            } else
                pick_pair(ri0, ri1, (uint32_t)p.pool_n, a, b);
            Pa = grows + (size_t)((int)a + p.pool_lo) * D; Pb2 = grows + (size_t)((int)b + p.pool_lo) * D;
        }
        if (snooker)
            g1 = 1.2 + (2.2 - 1.2) * u_g1;  // crossover.jl:249
        else if (p.proposal_kind == 0) {
            g1 = 0.5 + (1.0 - 0.5) * u_g1;  // crossover.jl:162
            use_base = p.iter <= p.burnin;  // crossover.jl:164
            if (use_base) g2 = 0.5 + (1.0 - 0.5) * u_g2;
        } else if (p.proposal_kind == 1)
            g1 = 2.38;  // crossover.jl:191
        else
            g1 = 2.38 / sqrt(2.0 * (double)D);  // crossover.jl:218
    }
    stamp(0);  // per-particle scalars drawn, partner rows known
    // ---- the few scalars every term depends on (theta'[0]; the observation sd; the scale a Normal(a, theta[ref]) prior points at):
    // one lane each.  What their proposals read -- the lane's scalar of the own row and of the two partner rows, the noise block --
    // is asked for NOW, before the base pick and its barrier: only the base row's scalar has to wait for those ----
    const long long S = p.N;
    const bool hier_b = p.family == FAM_HIER_BINOMIAL, hier_g = p.family == FAM_HIER_GAUSSIAN;
    int need = -1;
    if (tid == 0) need = 0;                                  // theta'[0]: the population mean of the subject effects
    else if (tid == 1) need = hier_g ? 2 + (int)S : -1;      // the observation sd (hier. Gaussian)
    else if (tid < 2 + p.n_seg && p.dimseg[tid - 2].t.kind == PR_NORMAL_REF) need = p.dimseg[tid - 2].t.ref;
    double n_t = 0.0, n_a = 0.0, n_b = 0.0;
    bool n_moves = false;
    U4 n_nb = {0, 0, 0, 0};
    if (need >= 0) {
        n_t = pt[need];
        n_moves = kind == 2 || p.mask[need] != 0;  // reset! (crossover.jl:336-352); mutation ignores the block
        if (n_moves) {
            n_nb = draw_block(p.seed, S_NOISE, p.sweep, (uint64_t)p.iter, eslot, (uint32_t)(need >> 2));
            if (kind == 0) { n_a = Pa[need]; n_b = Pb2[need]; }
            if (kind == 1) n_a = Pa[need];  // Pz (snooker_update!, crossover.jl:241-253)
        }
    }
    // ---- select_base (crossover.jl:282-289): stabilised softmax, the fixed three-level order (wave_cdf); wave 0, in registers ----
    const int n_cdf = p.pool_n;
    if (use_base && wave == 0) {
        double m = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (lane + 64 * r < n_cdf) m = fmax(m, pw_r[r]);
        m = wave_max(m);
        double e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = (lane + 64 * r < n_cdf) ? exp(pw_r[r] - m) : 0.0;
        if (n_cdf <= 64) {
            double e1[1] = {e[0]};
            wave_cdf<1>(e1, n_cdf);
            e[0] = e1[0];
        } else
            wave_cdf<4>(e, n_cdf);
        const int last = n_cdf - 1;
        const double lastv = (last >> 6) == 0 ? e[0] : (last >> 6) == 1 ? e[1] : (last >> 6) == 2 ? e[2] : e[3];
        const double total = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(lastv), last & 63),
                                              __builtin_amdgcn_readlane(__double2loint(lastv), last & 63));
        int b;
        if (!(total > 0.0) || !(total < INFINITY)) {
            b = (int)(u_base * n_cdf);
            b = b < n_cdf ? b : n_cdf - 1;
        } else {
            const double t = u_base * total;
            int cnt = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) cnt += __popcll(__ballot(lane + 64 * r < n_cdf && e[r] < t));
            b = cnt < n_cdf ? cnt : n_cdf - 1;
        }
        if (lane == 0) s_base = b + p.pool_lo;
    }
    if (p.proposal_kind == 0 && p.iter <= p.burnin) {  // (wave-uniform: inside burn-in every workgroup passes here)
        __syncthreads();  // base pick
        if (use_base) Pbase = grows_b + (size_t)s_base * D;
    }
    stamp(1);  // base picked
    // ---- snooker: project(Pm, Pd), project(Pn, Pd) with Pd = Pt - Pz need whole-row dot products first (utilities.jl:239-246): the
    // one pass over the three partner rows a frozen sweep makes, by the particles whose snooker coin fired (one in ten) ----
    double cm = 0.0, cn = 0.0, s2_snk = 0.0;
    if (p.theta_snooker > 0.0) {  // (wave-uniform: the barriers below are passed by every workgroup of such a run)
        double vm = 0.0, vn = 0.0, vd = 0.0;
        if (kind == 1)
            for (int j = tid; j < D; j += WG) {
                const double dj = pt[j] - Pa[j];
                vm += Pb2[j] * dj; vn += Pc[j] * dj; vd += dj * dj;
            }
        vm = subgroup_sum(vm, 64); vn = subgroup_sum(vn, 64); vd = subgroup_sum(vd, 64);
        if (lane == 0) { s_red[0][wave] = vm; s_red[1][wave] = vn; s_red[2][wave] = vd; }
        __syncthreads();
        if (tid == 0) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0;
            for (int i = 0; i < WG / 64; ++i) { a0 += s_red[0][i]; a1 += s_red[1][i]; a2 += s_red[2][i]; }
            s_snk[0] = a0; s_snk[1] = a1; s_snk[2] = a2;
        }
        __syncthreads();
        if (kind == 1) { cm = s_snk[0] / s_snk[2]; cn = s_snk[1] / s_snk[2]; s2_snk = s_snk[2]; }
    }

    stamp(2);  // snooker projections
    // ---- theta' of one scalar, by whoever asks (the block's scalars, and the few every term depends on) ----
    const double eps = p.eps, eps2 = p.eps - (-p.eps);
    auto in_block = [&](int j) -> bool { return p.mask[j] != 0; };
    auto theta_new = [&](int j) -> double {
        const double tj = pt[j];
        if (kind != 2 && !in_block(j)) return tj;  // reset! (crossover.jl:336-352)
        const U4 nb = draw_block(p.seed, S_NOISE, p.sweep, (uint64_t)p.iter, eslot, (uint32_t)(j >> 2));
        if (kind == 2) {  // pt + Normal(0, sigma): mutation.jl:15-18 (pair j >> 1 of the block: words x, y or z, w)
            const double2 z = box_muller_outofline((j & 2) ? nb.z : nb.x, (j & 2) ? nb.w : nb.y);
            return tj + p.sigma * ((j & 1) ? z.y : z.x);
        }
        const uint32_t wj = (j & 2) ? ((j & 1) ? nb.w : nb.z) : ((j & 1) ? nb.y : nb.x);
        const double bj = -eps + eps2 * u32unit(wj);  // b = Uniform(-eps, eps) crossover.jl:166
        if (kind == 1) {  // (Pt + gamma*(Pr1 - Pr2)) + b  crossover.jl:253
            const double dj = tj - Pa[j];
            const double t1 = dj * cm - dj * cn;
            return (tj + t1 * g1) + bj;
        }
        const double t1 = Pa[j] - Pb2[j];              // ((Pt + g1*(Pm-Pn)) + g2*(Pb-Pt)) + b  crossover.jl:168
        double t6 = tj + t1 * g1;
        if (use_base) {
            const double t4 = Pbase[j] - tj;
            t6 = t6 + t4 * g2;
        }
        return t6 + bj;
    };
    if (need >= 0) {
        double val = n_t;
        if (n_moves && kind == 2) {
            const double2 z = box_muller_outofline((need & 2) ? n_nb.z : n_nb.x, (need & 2) ? n_nb.w : n_nb.y);
            val = n_t + p.sigma * ((need & 1) ? z.y : z.x);
        } else if (n_moves && kind == 1) {
            const uint32_t wj = (need & 2) ? ((need & 1) ? n_nb.w : n_nb.z) : ((need & 1) ? n_nb.y : n_nb.x);
            const double dj = n_t - n_a;
            const double t1 = dj * cm - dj * cn;
            val = (n_t + t1 * g1) + (-eps + eps2 * u32unit(wj));
        } else if (n_moves) {
            const uint32_t wj = (need & 2) ? ((need & 1) ? n_nb.w : n_nb.z) : ((need & 1) ? n_nb.y : n_nb.x);
            const double bj = -eps + eps2 * u32unit(wj);
            const double t1 = n_a - n_b;
            double t6 = n_t + t1 * g1;
            if (use_base) {
                const double t4 = Pbase[need] - n_t;
                t6 = t6 + t4 * g2;
            }
            val = t6 + bj;
        }
        if (tid < 2)
            s_new[kFrozenMax + tid] = val;
        else {
            s_ref[0][tid - 2] = 1.0 / val;
            s_ref[1][tid - 2] = log(val);
        }
    }
    __syncthreads();
    stamp(3);  // theta'[0] and the reference scalars known to every wave
    const double mu0 = s_new[kFrozenMax];
    double sg_obs = 1.0, lsg_obs = 0.0, isg_obs = 1.0;
    if (hier_g) {
        sg_obs = s_new[kFrozenMax + 1];
        lsg_obs = log(sg_obs);
        isg_obs = 1.0 / sg_obs;
    }
    const double n_bin = p.c0;
    const bool prior_on = p.fitness_kind == 0;
#if DEMC_SOFTPLUS_TAB
    const SoftplusTab spt = softplus_tab_consts(s_sp);
    auto sp_of = [&](double x) { return softplus_tab(x, spt); };
#else
    const SoftplusC spc = softplus_consts();
    auto sp_of = [&](double x) { return softplus_fast_u(x, spc); };
#endif
    // ---- the pass: scalar j by thread j mod WG (consecutive lanes, consecutive scalars: 8-byte accesses, 512 B per wave) ----
    int oob = 0;
    double prior = 0.0, like = 0.0;
    auto term = [&](int j, double v) {
        int q = 0;
        for (int i = 1; i < p.n_seg; ++i) q += (j >= s_seg[i].start) ? 1 : 0;
        const DimTab* tb = &s_seg[q].t;
        oob |= !(v >= tb->lo && v <= tb->hi);  // in_bounds utilities.jl:70-78 (NaN fails)
        if (prior_on && tb->kind != PR_FLAT) {
            if (tb->kind == PR_NORMAL_REF) {
                const double z = (v - tb->a) * s_ref[0][q];
                prior += -(z * z + kLog2Pi) / 2.0 - s_ref[1][q];
            } else if (tb->kind == PR_NORMAL) {
                const double z = (v - tb->a) * tb->b;
                prior += tb->c - 0.5 * (z * z);
            } else
                prior += prior_term_ref_outofline(tb, v, s_ref[0][q], s_ref[1][q]);
        }
        const long long s = (long long)j - 2;  // the subject behind the scalar
        if (s >= 0 && s < S) {
            if (hier_b) {  // k log p + (n-k) log(1-p), p = logistic(eta): one softplus per subject
                const double eta = mu0 + v;
                like += -n_bin * sp_of(-eta) - (n_bin - p.data[s]) * eta;
            } else {  // Hierarchical_Example.jl:36-44: p.d observations per subject
                const double mu = mu0 + v;
                const int n = p.d;
                double l = 0.0;
                for (int o = 0; o < n; ++o) {
                    const double z = (p.data[s * n + o] - mu) * isg_obs;
                    l += -(z * z + kLog2Pi) / 2.0 - lsg_obs;
                }
                like += l;
            }
        }
    };
    auto uni = [](double x) {  // a wave-uniform value, held in SGPRs
        return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
    };
    // a piece of the row inside ONE table segment with a plain prior and a subject behind every scalar (all of a hierarchical
    // row but its two ends): bounds and prior entry wave-uniform (SGPRs), the body is the subject's term and nothing else
    struct SegC { double lo, hi, a, b, c, r_inv, r_log; int kd; };
    auto seg_consts = [&](int q) -> SegC {
        const DimTab tb = s_seg[q].t;
        SegC sc;
        sc.lo = uni(tb.lo); sc.hi = uni(tb.hi); sc.a = uni(tb.a); sc.b = uni(tb.b); sc.c = uni(tb.c);
        sc.r_inv = s_ref[0][q]; sc.r_log = s_ref[1][q];
        sc.kd = __builtin_amdgcn_readfirstlane(tb.kind);
        return sc;
    };
    auto seg_fast = [&](const SegC& sc, int a_, int b_) -> bool {
        return (sc.kd == PR_FLAT || sc.kd == PR_NORMAL || sc.kd == PR_NORMAL_REF) && a_ >= 2 && (long long)b_ <= S + 2 && hier_b && (D & 1) == 0;
    };
    // one scalar's terms (off: a lane past the end adds nothing).  What is linear in the piece is summed raw -- z^2, softplus,
    // (n - k) eta: an FMA or an addition a scalar -- and scaled once per piece by flush(): the constants of the Normal densities
    // times the piece's length by thread 0, -1/2 and -n on every thread's sums
    double az2 = 0.0, asp = 0.0, ace = 0.0;
    auto subject = [&](const SegC& sc, double v, double kk, bool on) {
        const int ob = !(v >= sc.lo && v <= sc.hi);
        if (prior_on && sc.kd != PR_FLAT) {
            const double z = (v - sc.a) * (sc.kd == PR_NORMAL_REF ? sc.r_inv : sc.b);
            const double zm = on ? z : 0.0;
            az2 = fma(zm, zm, az2);
        }
        const double eta = mu0 + v;
        const double sp = sp_of(-eta);
        const double cm = on ? n_bin - kk : 0.0;
        asp += on ? sp : 0.0;
        ace = fma(cm, eta, ace);
        oob |= on ? ob : 0;
    };
    auto flush = [&](const SegC& sc, int n) {  // n = the scalars subject() saw in this piece, over all threads
        if (prior_on && sc.kd != PR_FLAT) {
            prior += -0.5 * az2;
            if (tid == 0) prior += (double)n * (sc.kd == PR_NORMAL_REF ? -0.5 * kLog2Pi - sc.r_log : sc.c);
        }
        like += -n_bin * asp - ace;
        az2 = 0.0; asp = 0.0; ace = 0.0;
    };
    // the runs of the block mask (run-length table in the kernarg: scalar loads, wave-uniform): scalars of runs OUTSIDE the block
    // are taken as they are, the block's few scalars are proposed by the first threads
    auto run_lo = [&](int r) { return p.mrun_start[r]; };
    auto run_hi = [&](int r) { return r + 1 < p.n_mrun ? p.mrun_start[r + 1] : D; };
    double ds1 = 0.0;  // snooker: |Pt' - Pz|^2 - |Pt - Pz|^2, which only the block's scalars contribute to
    if (kind != 2) {
        for (int r = 0; r < p.n_mrun; ++r) {
            const int lo = run_lo(r), hi = run_hi(r);
            if ((p.mrun_in >> r) & 1u) {  // inside the block
                auto moved = [&](int j) {
                    const double v = theta_new(j);
                    term(j, v);
                    if (h_pass) hrow[j] = pt[j];
                    if (kind == 1) {  // adjust_loglike norms (crossover.jl:268-273)
                        const double z = Pa[j], a1 = v - z, a0 = pt[j] - z;
                        ds1 += a1 * a1 - a0 * a0;
                    }
                };
                if (!BIG || hi - lo <= WG) {  // a few hyper-parameters: one scalar per thread (the host sends longer runs to BIG)
                    if (tid < hi - lo) moved(lo + tid);
                    continue;
                }
                if constexpr (BIG)
                // a long run (the subject block): the proposal of every scalar formed on the fly from the own row and the partner
                // rows -- no LDS row; an accepted particle forms it again for its stores -- a noise block (four scalars) per thread
                // and round, the next round's rows asked for before this round's arithmetic
                for (int q = 0; q < p.n_seg; ++q) {
                    const int s_lo = p.seg_start[q], s_hi = q + 1 < p.n_seg ? p.seg_start[q + 1] : D;
                    const int a_ = lo > s_lo ? lo : s_lo, b_ = hi < s_hi ? hi : s_hi;
                    if (a_ >= b_) continue;
                    const SegC sc = seg_consts(q);
                    const int m_lo = (a_ + 3) >> 2, m_hi = b_ >> 2;  // noise blocks wholly inside the piece
                    if (!seg_fast(sc, a_, b_) || m_lo >= m_hi) {
                        for (int j = a_ + tid; j < b_; j += WG) moved(j);
                        continue;
                    }
                    const int n_edge = (4 * m_lo - a_) + (b_ - 4 * m_hi);
                    if (tid < n_edge) moved(tid < 4 * m_lo - a_ ? a_ + tid : 4 * m_hi + (tid - (4 * m_lo - a_)));
                    const int m_last = m_hi - 1;
                    auto ld = [&](const double* base, int m, int off) {
                        return *reinterpret_cast<const double2*>(base + (4 * (long long)(m < m_last ? m : m_last) + off));
                    };
                    const bool two = kind == 0;  // (wave-uniform: the second partner row; the base row inside burn-in)
                    const double2 zero2 = make_double2(0.0, 0.0);
                    int m = m_lo + tid;
                    double2 t0 = ld(pt, m, 0), t1 = ld(pt, m, 2), a0 = ld(Pa, m, 0), a1 = ld(Pa, m, 2);
                    double2 b0 = two ? ld(Pb2, m, 0) : zero2, b1 = two ? ld(Pb2, m, 2) : zero2;
                    double2 e0 = use_base ? ld(Pbase, m, 0) : zero2, e1 = use_base ? ld(Pbase, m, 2) : zero2;
                    double2 c0 = ld(p.data, m, -2), c1 = ld(p.data, m, 0);
                    for (; m < m_hi; m += WG) {
                        const int mn = m + WG;
                        const double2 nt0 = ld(pt, mn, 0), nt1 = ld(pt, mn, 2), na0 = ld(Pa, mn, 0), na1 = ld(Pa, mn, 2);
                        const double2 nb0 = two ? ld(Pb2, mn, 0) : zero2, nb1 = two ? ld(Pb2, mn, 2) : zero2;
                        const double2 ne0 = use_base ? ld(Pbase, mn, 0) : zero2, ne1 = use_base ? ld(Pbase, mn, 2) : zero2;
                        const double2 nc0 = ld(p.data, mn, -2), nc1 = ld(p.data, mn, 0);
                        if (h_pass) {
                            *reinterpret_cast<double2*>(hrow + 4 * (size_t)m) = t0;
                            *reinterpret_cast<double2*>(hrow + 4 * (size_t)m + 2) = t1;
                        }
                        const U4 nb = draw_block(p.seed, S_NOISE, p.sweep, (uint64_t)p.iter, eslot, (uint32_t)m);
                        auto one = [&](double tj, double aj, double bj2, double ej, uint32_t wj, double kk) {
                            const double bj = -eps + eps2 * u32unit(wj);
                            double v;
                            if (kind == 1) {
                                const double dj = tj - aj;
                                const double t1_ = dj * cm - dj * cn;
                                v = (tj + t1_ * g1) + bj;
                                const double a1_ = v - aj;
                                ds1 += a1_ * a1_ - dj * dj;
                            } else {
                                const double t1_ = aj - bj2;
                                double t6 = tj + t1_ * g1;
                                if (use_base) {
                                    const double t4 = ej - tj;
                                    t6 = t6 + t4 * g2;
                                }
                                v = t6 + bj;
                            }
                            subject(sc, v, kk, true);
                        };
                        one(t0.x, a0.x, b0.x, e0.x, nb.x, c0.x); one(t0.y, a0.y, b0.y, e0.y, nb.y, c0.y);
                        one(t1.x, a1.x, b1.x, e1.x, nb.z, c1.x); one(t1.y, a1.y, b1.y, e1.y, nb.w, c1.y);
                        t0 = nt0; t1 = nt1; a0 = na0; a1 = na1; b0 = nb0; b1 = nb1; e0 = ne0; e1 = ne1; c0 = nc0; c1 = nc1;
                    }
                    flush(sc, 4 * (m_hi - m_lo));
                }
                continue;
            }
            // ... cut at the table's segment borders
            for (int q = 0; q < p.n_seg; ++q) {
                const int s_lo = p.seg_start[q], s_hi = q + 1 < p.n_seg ? p.seg_start[q + 1] : D;
                const int a_ = lo > s_lo ? lo : s_lo, b_ = hi < s_hi ? hi : s_hi;
                if (a_ >= b_) continue;
                const SegC sc = seg_consts(q);
                if (!seg_fast(sc, a_, b_)) {
                    for (int j = a_ + tid; j < b_; j += WG) {
                        const double v = pt[j];
                        term(j, v);
                        if (srow) srow[j] = v;
                        if (h_pass) hrow[j] = v;
                    }
                    continue;
                }
                if (((a_ | b_) & 1) == 0) {
                    // whole dim pairs: 16-byte loads, PAIRS pairs of a thread per round (independent softplus chains), the NEXT
                    // round's rows and counts asked for before this round's arithmetic -- without that every round waits out a
                    // full HBM latency (the first form of this loop: 168 us per launch where k_longrow takes 137)
                    const int P_lo = a_ >> 1, P_hi = b_ >> 1, last = P_hi - 1;
                    // (pair k = scalars 2k, 2k + 1; their subjects' counts sit two doubles earlier in p.data: off = -2)
                    auto ldp = [&](const double* base, int k, int off = 0) {
                        return *reinterpret_cast<const double2*>(base + (2 * (long long)(k < last ? k : last) + off));
                    };
                    int k = P_lo + tid;
                    if constexpr (PAIRS == 2) {
                        double2 v0 = ldp(pt, k), v1 = ldp(pt, k + WG), c0 = ldp(p.data, k, -2), c1 = ldp(p.data, k + WG, -2);
                        for (; k < P_hi; k += 2 * WG) {
                            const double2 nv0 = ldp(pt, k + 2 * WG), nv1 = ldp(pt, k + 3 * WG);
                            const double2 nc0 = ldp(p.data, k + 2 * WG, -2), nc1 = ldp(p.data, k + 3 * WG, -2);
                            const bool on1 = k + WG < P_hi;
                            if (srow) {
                                *reinterpret_cast<double2*>(srow + 2 * (long long)k) = v0;
                                if (on1) *reinterpret_cast<double2*>(srow + 2 * (long long)(k + WG)) = v1;
                            }
                            if (h_pass) {
                                *reinterpret_cast<double2*>(hrow + 2 * (long long)k) = v0;
                                if (on1) *reinterpret_cast<double2*>(hrow + 2 * (long long)(k + WG)) = v1;
                            }
                            subject(sc, v0.x, c0.x, true); subject(sc, v0.y, c0.y, true);
                            subject(sc, v1.x, c1.x, on1); subject(sc, v1.y, c1.y, on1);
                            v0 = nv0; v1 = nv1; c0 = nc0; c1 = nc1;
                        }
                    } else {
                        double2 v0 = ldp(pt, k), c0 = ldp(p.data, k, -2);
                        for (; k < P_hi; k += WG) {
                            const double2 nv0 = ldp(pt, k + WG), nc0 = ldp(p.data, k + WG, -2);
                            if (srow) *reinterpret_cast<double2*>(srow + 2 * (long long)k) = v0;
                            if (h_pass) *reinterpret_cast<double2*>(hrow + 2 * (long long)k) = v0;
                            subject(sc, v0.x, c0.x, true); subject(sc, v0.y, c0.y, true);
                            v0 = nv0; c0 = nc0;
                        }
                    }
                } else {
                    for (int j = a_ + tid; j < b_; j += WG) {
                        const double v = pt[j];
                        subject(sc, v, p.data[j - 2], true);
                        if (srow) srow[j] = v;
                        if (h_pass) hrow[j] = v;
                    }
                }
                flush(sc, b_ - a_);
            }
        }
    } else {
        // mutation: every scalar moves (mutation! ignores the block).  Inside a fast segment a thread takes whole noise blocks --
        // four scalars: one Philox block, two Box-Muller pairs, 16-byte loads -- and the segment's ragged ends (and every other
        // segment) go scalar by scalar
        for (int q = 0; q < p.n_seg; ++q) {
            const int s_lo = p.seg_start[q], s_hi = q + 1 < p.n_seg ? p.seg_start[q + 1] : D;
            const SegC sc = seg_consts(q);
            const int m_lo = (s_lo + 3) >> 2, m_hi = s_hi >> 2;  // noise blocks wholly inside the segment
            if (!seg_fast(sc, s_lo, s_hi) || m_lo >= m_hi) {
                for (int j = s_lo + tid; j < s_hi; j += WG) term(j, theta_new(j));
                continue;
            }
            const int n_edge = (4 * m_lo - s_lo) + (s_hi - 4 * m_hi);
            if (tid < n_edge) {
                const int j = tid < 4 * m_lo - s_lo ? s_lo + tid : 4 * m_hi + (tid - (4 * m_lo - s_lo));
                term(j, theta_new(j));
            }
            for (int m = m_lo + tid; m < m_hi; m += WG) {
                const double2 ta = *reinterpret_cast<const double2*>(pt + 4 * (size_t)m), tb2 = *reinterpret_cast<const double2*>(pt + 4 * (size_t)m + 2);
                const double2 ca = *reinterpret_cast<const double2*>(p.data + 4 * (size_t)m - 2), cb = *reinterpret_cast<const double2*>(p.data + 4 * (size_t)m);
                const U4 nb = draw_block(p.seed, S_NOISE, p.sweep, (uint64_t)p.iter, eslot, (uint32_t)m);
                const double2 za = box_muller_outofline(nb.x, nb.y), zb = box_muller_outofline(nb.z, nb.w);
                subject(sc, ta.x + p.sigma * za.x, ca.x, true); subject(sc, ta.y + p.sigma * za.y, ca.y, true);
                subject(sc, tb2.x + p.sigma * zb.x, cb.x, true); subject(sc, tb2.y + p.sigma * zb.y, cb.y, true);
            }
            flush(sc, 4 * (m_hi - m_lo));
        }
    }
    stamp(4);  // the pass (wave 0's share)
    // ---- one reduction: waves on the DPP network, then a fixed tree over the waves ----
    prior = subgroup_sum(prior, 64); like = subgroup_sum(like, 64); oob = subgroup_sum(oob, 64);
    if (p.theta_snooker > 0.0) ds1 = subgroup_sum(ds1, 64);
    if (lane == 0) {
        s_red[0][wave] = prior; s_red[1][wave] = like; s_red[2][wave] = ds1;
        s_redi[wave] = oob;
    }
    __syncthreads();
    if (tid == 0) {
        auto tree = [&](const double* s) {
            double r = 0.0;
            for (int i = 0; i < WG / 64; i += 2) r += s[i] + (i + 1 < WG / 64 ? s[i + 1] : 0.0);
            return r;
        };
        double pr = tree(s_red[0]), lk = tree(s_red[1]);
        if (hier_b) lk = p.c2 + lk;  // + sum_s log C(n, k_s): data-only, summed once at demc_set_model
        int ob = 0;
        for (int i = 0; i < WG / 64; ++i) ob |= s_redi[i];
        // ---- compute_posterior! + mh_update! (utilities.jl:92-99, 55-58, 201-210) ----
        double wp;
        if (p.fitness_kind == 1)
            wp = ob ? (p.update_kind == 1 ? -INFINITY : INFINITY) : lk;
        else
            wp = ob ? -INFINITY : pr + lk;
        double adj = 0.0;  // adjust_loglike (crossover.jl:268-273): (d - 1) (log |Pt' - Pz| - log |Pt - Pz|)
        if (kind == 1) adj = (double)(D - 1) * (0.5 * log(s2_snk + tree(s_red[2])) - 0.5 * log(s2_snk));
        const int acc = decide_mh(p.mode, p.update_kind, u_acc, wp, w_cur, adj);
        if (acc) p.weight[slot] = wp;
        if (!BIG && p.snap_weight) p.snap_weight[slot] = acc ? wp : w_cur;
        if (p.store_row >= 0) {
            const size_t hrow = (size_t)p.store_row * p.P + slot;
            if (p.update_kind == 0 && p.mode == MODE_STEP) {  // utilities.jl:207-208
                p.acc_hist[hrow] = (unsigned char)acc;
                p.lp_hist[hrow] = acc ? wp : w_cur;
            }
            p.id_hist[hrow] = (int)p.id[slot];
        }
        s_acc = acc;
    }
    __syncthreads();
    const int acc = s_acc;
    stamp(5);  // reduced, decided (the slowest wave in)
    // ---- the row moves: an accepted crossover writes the block's scalars, an accepted mutation the row (formed again); the
    // history row (a sweep that is the iteration's last) is the row as it stands after the decision ----
    double* trow = p.theta + slot * D;
    if (acc && kind != 2) {
        for (int r = 0; r < p.n_mrun; ++r) {
            const int lo = run_lo(r), hi = run_hi(r);
            if ((p.mrun_in >> r) & 1u) {
                auto put = [&](int j) {
                    const double v = theta_new(j);
                    trow[j] = v;  // utilities.jl:204
                    if (hrow) hrow[j] = v;
                    if (srow) srow[j] = v;
                };
                if (!BIG || hi - lo <= WG) {
                    if (tid < hi - lo) put(lo + tid);
                    continue;
                }
                if constexpr (BIG) {
                const int m_lo = (lo + 3) >> 2, m_hi = hi >> 2;
                if ((D & 1) != 0 || m_lo >= m_hi) {
                    for (int j = lo + tid; j < hi; j += WG) put(j);
                    continue;
                }
                const int n_edge = (4 * m_lo - lo) + (hi - 4 * m_hi);
                if (tid < n_edge) put(tid < 4 * m_lo - lo ? lo + tid : 4 * m_hi + (tid - (4 * m_lo - lo)));
                auto ld = [&](const double* base, int m, int off) { return *reinterpret_cast<const double2*>(base + (4 * (long long)m + off)); };
                const bool two = kind == 0;
                const double2 zero2 = make_double2(0.0, 0.0);
                for (int m = m_lo + tid; m < m_hi; m += WG) {
                    const double2 t0 = ld(pt, m, 0), t1 = ld(pt, m, 2), a0 = ld(Pa, m, 0), a1 = ld(Pa, m, 2);
                    const double2 b0 = two ? ld(Pb2, m, 0) : zero2, b1 = two ? ld(Pb2, m, 2) : zero2;
                    const double2 e0 = use_base ? ld(Pbase, m, 0) : zero2, e1 = use_base ? ld(Pbase, m, 2) : zero2;
                    const U4 nb = draw_block(p.seed, S_NOISE, p.sweep, (uint64_t)p.iter, eslot, (uint32_t)m);
                    auto one = [&](double tj, double aj, double bj2, double ej, uint32_t wj) -> double {
                        const double bj = -eps + eps2 * u32unit(wj);
                        if (kind == 1) {
                            const double dj = tj - aj;
                            const double t1_ = dj * cm - dj * cn;
                            return (tj + t1_ * g1) + bj;
                        }
                        const double t1_ = aj - bj2;
                        double t6 = tj + t1_ * g1;
                        if (use_base) {
                            const double t4 = ej - tj;
                            t6 = t6 + t4 * g2;
                        }
                        return t6 + bj;
                    };
                    const double2 v0 = make_double2(one(t0.x, a0.x, b0.x, e0.x, nb.x), one(t0.y, a0.y, b0.y, e0.y, nb.y));
                    const double2 v1 = make_double2(one(t1.x, a1.x, b1.x, e1.x, nb.z), one(t1.y, a1.y, b1.y, e1.y, nb.w));
                    *reinterpret_cast<double2*>(trow + 4 * (size_t)m) = v0; *reinterpret_cast<double2*>(trow + 4 * (size_t)m + 2) = v1;
                    if (hrow) { *reinterpret_cast<double2*>(hrow + 4 * (size_t)m) = v0; *reinterpret_cast<double2*>(hrow + 4 * (size_t)m + 2) = v1; }
                }
                }
            } else if (hrow && !h_pass)  // (past burn-in the history scalars of the runs outside the block went out in the pass)
                for (int j = lo + tid; j < hi; j += WG) hrow[j] = pt[j];
        }
    } else if (acc) {
        const int n_blocks = (D + 3) >> 2;
        for (int m = tid; m < n_blocks; m += WG) {
            const U4 nb = draw_block(p.seed, S_NOISE, p.sweep, (uint64_t)p.iter, eslot, (uint32_t)m);
            const double2 za = box_muller_outofline(nb.x, nb.y), zb = box_muller_outofline(nb.z, nb.w);
            const double zz[4] = {za.x, za.y, zb.x, zb.y};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * m + e < D) {
                    const double v = pt[4 * m + e] + p.sigma * zz[e];
                    trow[4 * m + e] = v;
                    if (hrow) hrow[4 * m + e] = v;
                    if (srow) srow[4 * m + e] = v;
                }
        }
    } else if (hrow && !h_pass) {
        for (int j = tid; j < D; j += WG) hrow[j] = pt[j];  // utilities.jl:170-180
    }
    // (by-product snapshot: the frozen scalars of a crossover / snooker sweep went out in the pass; what is left is the block's
    // scalars of a rejected proposal and the whole row of a rejected mutation)
    if (srow && !acc) {
        if (kind == 2) {
            for (int j = tid; j < D; j += WG) srow[j] = pt[j];
        } else {
            for (int r = 0; r < p.n_mrun; ++r)
                if ((p.mrun_in >> r) & 1u) {
                    const int lo = run_lo(r), hi = run_hi(r);
                    if (tid < hi - lo) srow[lo + tid] = pt[lo + tid];
                }
        }
    }
    stamp(6);
#ifdef DEMC_STAMPS
    if (tid == 0 && 2 * ((long long)blockIdx.x + 1) <= p.P)
        p.prop_adj[2 * blockIdx.x + 1] = (double)__builtin_amdgcn_s_memrealtime() + 0.25 * kind + 0.125 * acc;
#endif
}

// Every instance the runtime launches (demc_frozen.cpp instantiates them, demc_hip.cpp declares them extern).  <WG, MINW, PAIRS, BIG>
#define DEMC_FROZEN_INSTANCES(X) X(256, 3, 2, false) X(256, 3, 2, true)
#ifdef DEMC_EXPERIMENTS  // A/B builds: other workgroup sizes, register budgets and pairs per round (profiles/r05/NOTES.md section 8)
#define DEMC_FROZEN_INSTANCES_EXP(X) \
    X(64, 3, 2, false) X(128, 3, 2, false) X(512, 3, 2, false) X(768, 3, 2, false) X(1024, 4, 2, false) X(256, 4, 2, false) \
    X(256, 4, 1, false) X(256, 3, 1, false) X(256, 5, 1, false) X(512, 2, 2, true) X(1024, 4, 2, true) X(256, 2, 2, true)
#else
#define DEMC_FROZEN_INSTANCES_EXP(X)
#endif
#ifdef DEMC_FROZEN_EXTERN
#define DEMC_X_(...) extern template __global__ void k_frozen_sweep<__VA_ARGS__>(KParams);
DEMC_FROZEN_INSTANCES(DEMC_X_)
DEMC_FROZEN_INSTANCES_EXP(DEMC_X_)
#undef DEMC_X_
#endif

}  // namespace demc
