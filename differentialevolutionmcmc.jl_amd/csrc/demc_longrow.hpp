// demc_longrow.hpp -- K1 for LONG rows (D in the thousands: hierarchical models with one scalar per subject, BASELINE cfg4).
//
// One workgroup of WG threads per moving particle, the whole update in one launch (proposal, bounds, prior, the
// hierarchical likelihood, Metropolis accept, state and history write-back), like k_propose's fused tail -- but built for
// rows that stream from HBM/L2 instead of an LDS tile:
//   * ONE pass over the row.  The proposal of scalar j, its prior term AND the likelihood term of the subject behind it are
//     formed together, theta' parked in LDS for the row moves; the few scalars every term needs (the hyper-parameters:
//     theta'[0], theta'[ref], the observation sd) are proposed first, one lane each, and handed round through LDS.
//     k_propose makes two passes (proposal, then the subjects from an LDS copy).
//   * The pass runs wave by wave in SPANS.  A lane's unit of work is one noise block (four consecutive scalars), a wave's
//     round 64 blocks.  Rounds that lie inside one uniform region -- one table segment with a plain prior, one side of the
//     sweep's block, a subject behind every scalar: all of a hierarchical row but its two ends -- run in a loop compiled
//     for what the region needs (run_span: frozen row, crossover with / without the base row, snooker moving / frozen,
//     mutation; with or without Binomial counts): the same loads in the same order every round, the next block's rows
//     requested before the current block's arithmetic, the table entry in SGPRs, four independent softplus chains.  The
//     static shape is what lets the compiler wait for the current block only (s_waitcnt vmcnt(n)); with the loads under
//     run-time conditions it waits for all of them and the prefetch overlaps nothing.
//   * Regions are found from run-length tables in the kernarg (segment starts, runs of the block mask): scalar loads and
//     SALU.  The rounds at the ends of regions are done one scalar per lane by all waves together; a sweep the span loops
//     do not cover (recombination, an odd D, more mask runs than the table holds) takes the general per-pair body, which
//     skips the noise draws and partner loads of scalars outside the block (reset!, crossover.jl:336-352).
//   * The per-particle scalars (coins, partner indices, gammas, accept uniform, the group's mutation coin) are drawn by
//     seven lanes of every wave and shared through SGPRs -- no LDS, no barrier; the base pick (burn-in only) is done by
//     wave 0 in registers.
//   * 512 threads per workgroup, or 256 with two workgroups per CU when the launch has that many (launch_phase).
// Same addressed draws and the same arithmetic per scalar as k_propose with a workgroup per particle: proposals and decisions
// are the ones that kernel produces; the sums of the prior and likelihood terms run in another order (tests/
// test_gpu_parity.py::test_longrow_*).
//
// Reference: crossover!/snooker_update!/mutation!/recombination!/reset!/in_bounds/compute_posterior!/mh_update!/
// store_samples! (crossover.jl:30-99,154-257,301-352; mutation.jl:13-25; utilities.jl:70-99,161-180,201-226).
#pragma once
#include "demc_kernels.hpp"

namespace demc {

// value held by lane `src` of the wave, as a wave-uniform scalar
__device__ inline uint32_t wave_get(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }

// (A/B build only, make EXPERIMENTS=1: leave the kernel at a given point, to time what lies before it --
// tools/lr_exit_experiment.py; the product build compiles these away)
#ifdef DEMC_EXPERIMENTS
#define DEMC_LR_EXIT(n) \
    if (p.n_split == -(n)) return
#define DEMC_LR_NEXT(n) \
    if (p.n_split == -(n)) continue  /* inside the persistent loop: on to the workgroup's next particle */
#else
#define DEMC_LR_EXIT(n)
#define DEMC_LR_NEXT(n)
#endif

#ifndef DEMC_LR_PREFETCH
#define DEMC_LR_PREFETCH 1  // blocks requested ahead of the one being worked on in the span loops
#endif
// (Tried and dropped, round 3: theta' parked in the handle's proposal buffer instead of LDS, so that the register file and
// not two 80 KB rows of LDS sets the occupancy -- 'k_longrow<256, GS>'.  Whole cfg4, per launch: LDS form 0.174 ms; global
// scratch at the same two workgroups per CU 0.191; at three (168 VGPRs, 44 spilled) 0.202; at four (128 VGPRs, 88 spilled)
// 0.219.  More workgroups in flight make the launch SLOWER: the phase is not waiting for latency that more waves could hide.)
// PER_CU workgroups of WG threads share a CU (each parks an 80 KB row in LDS: at most two): WG / 64 * PER_CU / 4 waves per SIMD,
// which is what the register budget follows from -- (256, 2) and (512, 1), the forms that ship: two waves per SIMD, 256 registers.
// Round 5 measured the forms with MORE waves per CU (A/B builds only; profiles/r05/NOTES.md): (384, 2) -- three waves per SIMD,
// 168 registers, no spill, seven rounds per particle instead of ten, two workgroups per CU confirmed by the occupancy query --
// 0.216 ms per launch of the whole cfg4 against 0.167; (512, 2) does not fit twice (LDS) and spills at 128 registers: 0.239.
// With round 3's global-scratch forms and round 4's LITE instance that makes four measurements that say the same: this kernel
// gets slower with more waves in flight.
template <int WG, int PER_CU = (WG <= 256 ? 2 : 1)>
__global__ __launch_bounds__(WG, (WG / 64 * PER_CU + 3) / 4) void k_longrow(KParams p_arg) {
    // The parameters are read through the kernarg segment pointer, made opaque once per particle of the persistent loop below:
    // otherwise every field the body reads is loaded once, ahead of the loop, and stays live across it -- hundreds of SGPRs,
    // spilled into VGPR lanes, which then spill themselves (256 VGPRs + scratch against 220 for the one-particle kernel).
    typedef const KParams __attribute__((address_space(4))) * KArg;
    KArg kp = (KArg)__builtin_amdgcn_kernarg_segment_ptr();  // (KParams is the kernel's only explicit argument: offset 0)
    (void)p_arg;
    {
    const auto& p = *kp;
    extern __shared__ double lds[];  // theta' of the particle [D (+1 if odd)] | cumulative pool weights [pool_n + chunks], the
                                     // latter only for pools of more than 256 (smaller ones: in wave 0's registers)
    __shared__ double s_red_[2][5][WG / 64];  // (by particle parity: a fast wave's partial sums of the next particle must not
    __shared__ int s_redi_[2][WG / 64];       //  meet a slow wave still reading this one's)
    __shared__ DimSeg s_seg[kMaxDimSeg];
    __shared__ int s_base;
    __shared__ double s_ref[2][kMaxDimSeg];
    __shared__ double s_hyp[2];
    // softplus_tab's tables (2 576 bytes): where the CU's LDS has room for them -- one workgroup per CU; two rows of cfg4's length
    // leave 496 bytes, and those instances keep the table-free softplus_fast
    constexpr bool kSpTab = PER_CU == 1;
    __shared__ __attribute__((aligned(16))) double s_sp[kSpTab ? kSpDoubles : 2];
    DEMC_STAMP_INIT();
#ifdef DEMC_STAMPS
    const unsigned long long t_real0__ = __builtin_amdgcn_s_memrealtime();  // (100 MHz: the shader clock of the run = stamp 10 / stamp 18 x 100 MHz)
#endif
    DEMC_LR_EXIT(1);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave: an SGPR)
    const int D = p.D, Np = p.Np;
    // the prior-table segments go to LDS: loaded now, stored just before the first barrier (no wait on a cold line here)
    constexpr int kSegDoubles = (int)(sizeof(DimSeg) / sizeof(double));
    static_assert(kMaxDimSeg * kSegDoubles <= WG, "one segment word per lane");
    const bool seg_lane = tid < p.n_seg * kSegDoubles;
    const double seg_word = seg_lane ? reinterpret_cast<const double*>(p.dimseg)[tid] : 0.0;

    SoftplusTab spt{};
    if constexpr (kSpTab) {
        load_softplus_table(s_sp, tid, WG);  // (visible behind the first particle's first barrier)
        spt = softplus_tab_consts(s_sp);
    }
    auto sp_of = [&](double x) -> double {
        if constexpr (kSpTab) return softplus_tab(x, spt);
        else return softplus_fast(x);
    };
    double* scr = lds;
    double* cdf = lds + ((D + 1) & ~1);
    const bool even = (D & 1) == 0;
    // ---- PERSISTENT: the workgroup takes particles vb = blockIdx.x, blockIdx.x + gridDim.x, ... (launch_phase sizes the grid
    // to the workgroups that are resident at once).  What that buys is the ROW MOVES: a workgroup that ends with its stores
    // (80 KB to the history row, 80 KB more when accepted) keeps its CU slot until the last of them is acknowledged -- ~29 k of
    // a ~100 k-cycle workgroup at D = 10 002 -- and the next one then waits ~3.6 k cycles for its kernarg.  Here the accepted /
    // current row of particle n is written from inside the span loops of particle n + 1, block by block just before the
    // block's LDS slot takes the new proposal (`pend_*`; run_span's DEFER form); only the last particle stores at its end.
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    bool pend = false, pend_acc = false, pend_masked = false;        // a row move left for the next particle's spans
    double *pend_trow = nullptr, *pend_hrow = nullptr;
    unsigned long long pend_wbits = 0;
    const int n_prop = p.n_groups * p.n_act;
    int par = 0;
    for (int vb = blockIdx.x; vb < n_prop; vb += gridDim.x, par ^= 1) {
    asm volatile("" : "+s"(kp));
    const auto& p = *kp;
#ifdef DEMC_STAMPS_TIMELINE
    const unsigned long long t_part__ = __builtin_amdgcn_s_memrealtime();
#endif
    // (the same for the thread index: what the body derives from it -- lane masks, row addresses -- is recomputed per particle
    // instead of being carried, in registers, across the whole loop)
    int tid_o = threadIdx.x;
    asm volatile("" : "+v"(tid_o));
    const int tid = tid_o, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = p.D, Np = p.Np;
    double* scr = lds;
    double* cdf = lds + ((D + 1) & ~1);
    const bool even = (D & 1) == 0;
    double (&s_red)[5][WG / 64] = s_red_[par];
    int (&s_redi)[WG / 64] = s_redi_[par];
    const bool prev = pend, prev_acc = pend_acc, prev_masked = pend_masked;
    double* const prev_trow = pend_trow;
    double* const prev_hrow = pend_hrow;
    const unsigned long long prev_wbits = pend_wbits;
    pend = false;
    // particle of this pass; the particles of a group share an XCD (their partner rows then share its L2)
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
    // (the weights select_base reads and the base row: the sweep-start snapshot when the launch has one -- KParams::base_theta)
    const double* gwb = p.base_weight ? p.base_weight + (size_t)g * Np : gw;
    const double* pt = grows + (size_t)pl * D;
    const double w_cur = gw[pl];
    DEMC_STAMP_AT(16, 64, DEMC_STAMP_NOW() + (unsigned long long)(D & 1));  // kernarg in, addresses formed
    // the partner pool's weights for the base pick, asked for before anything else waits (wave 0, burn-in only)
    const bool maybe_base = p.mode == MODE_STEP && p.proposal_kind == 0 && p.iter <= p.burnin && wave == 0 && p.pool_n <= 256;
    double pw_r[4] = {0.0, 0.0, 0.0, 0.0};
    if (maybe_base) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (lane + 64 * r < p.pool_n) pw_r[r] = gwb[p.pool_lo + lane + 64 * r];
    }

    // ---- per-particle scalars: lanes 0..5 of every wave evaluate one Philox block each, the words travel through SGPRs ----
    // (lane 6 evaluates the group's block -- the coin of mutate_or_crossover! -- in the same pass)
    const bool hist_partners = p.partner_kind == 1;
    const bool glane = lane == 6;
    const U4 mine = draw_block(p.seed, glane ? S_GROUP : S_PART, p.sweep, (uint64_t)p.iter, glane ? (uint32_t)g_glob : eslot,
                               (uint32_t)(lane < 6 ? lane : 0));
    const bool is_mut = p.mode == MODE_STEP && u53(wave_get(mine.x, 6), wave_get(mine.y, 6)) <= p.beta;  // main.jl:199-207
    DEMC_STAMP_AT(17, 64, DEMC_STAMP_NOW() + (unsigned long long)(is_mut ? 1 : 0));  // Philox block of the per-particle scalars done
    const double u_snk = u53(wave_get(mine.x, 0), wave_get(mine.y, 0)), u_base = u53(wave_get(mine.z, 0), wave_get(mine.w, 0));
    const uint32_t ri0 = wave_get(mine.x, 1), ri1 = wave_get(mine.y, 1), ri2 = wave_get(mine.z, 1);
    const double u_g1 = u53(wave_get(mine.x, 2), wave_get(mine.y, 2)), u_g2 = u53(wave_get(mine.z, 2), wave_get(mine.w, 2));
    const double u_acc = u53(wave_get(mine.x, 3), wave_get(mine.y, 3));
    int kind = 3;  // 0 DE, 1 snooker, 2 mutation, 3 identity
    int i0 = -1, i1 = -1, i2 = -1;
    const double *Pa = pt, *Pb2 = pt, *Pc = pt, *Pbase = pt;
    double g1 = 0.0, g2 = 0.0;
    bool use_base = false;
    if (p.mode == MODE_STEP) {
        if (is_mut)
            kind = 2;
        else {
            const bool snooker = u_snk <= p.theta_snooker;  // crossover.jl:31
            kind = snooker ? 1 : 0;
            if (hist_partners) {  // resample (crossover.jl:113-124): distinct cells of rows 1:(iter-1) x local particles
                const uint64_t hd0 = ((uint64_t)wave_get(mine.y, 4) << 32) | wave_get(mine.x, 4),
                               hd1 = ((uint64_t)wave_get(mine.w, 4) << 32) | wave_get(mine.z, 4),
                               hd2 = ((uint64_t)wave_get(mine.y, 5) << 32) | wave_get(mine.x, 5);
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
                i0 = (int)(a & 0x7fffffff); i1 = (int)(b & 0x7fffffff); i2 = snooker ? (int)(c & 0x7fffffff) : -1;
            } else if (snooker) {
                uint32_t a, b, c;  // snooker_update! draws 3 from the whole pool (crossover.jl:241)
                pick_triple(ri0, ri1, ri2, (uint32_t)p.pool_n, a, b, c);
                i0 = (int)a + p.pool_lo; i1 = (int)b + p.pool_lo; i2 = (int)c + p.pool_lo;
                Pa = grows + (size_t)i0 * D; Pb2 = grows + (size_t)i1 * D; Pc = grows + (size_t)i2 * D;
            } else {
                uint32_t a, b;
                if (p.exclude_self) {  // setdiff(group, [Pt]) crossover.jl:158
                    pick_pair(ri0, ri1, (uint32_t)p.pool_n - 1, a, b);
                    const uint32_t t = (uint32_t)(pl - p.pool_lo);
                    a += (a >= t); b += (b >= t);
                } else
                    pick_pair(ri0, ri1, (uint32_t)p.pool_n, a, b);
                i0 = (int)a + p.pool_lo; i1 = (int)b + p.pool_lo;
                Pa = grows + (size_t)i0 * D; Pb2 = grows + (size_t)i1 * D;
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
    }

    DEMC_STAMP_AT(12, 64, DEMC_STAMP_NOW());  // per-particle scalars drawn (stamps: wave 1, a typical wave; wave 0 picks the base)
    // ---- select_base (crossover.jl:282-289) over the partner pool: stabilised softmax, cumulative weights in the fixed
    // three-level order shared with the oracle and k_propose (quads, chunks of 16, chunk totals: wave_cdf).  Wave 0
    // alone; the other waves go straight to their first row loads and noise draws and meet it at the barrier below.
    const int n_cdf = p.pool_n;
    if (use_base && wave == 0) {
        const double* pw = gwb + p.pool_lo;
        double m = -INFINITY;
        if (n_cdf <= 256) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (lane + 64 * r < n_cdf) m = fmax(m, pw_r[r]);
        } else
            for (int i = lane; i < n_cdf; i += 64) m = fmax(m, pw[i]);
        m = wave_max(m);
        int b;
        if (n_cdf <= 256) {  // the usual case: the pool sits in this wave's registers, no LDS round trips (wave_cdf)
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
            if (!(total > 0.0) || !(total < INFINITY)) {
                b = (int)(u_base * n_cdf);
                b = b < n_cdf ? b : n_cdf - 1;
            } else {  // first i with cdf[i] >= t, else last = number of entries below t (cdf is monotone)
                const double t = u_base * total;
                int cnt = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) cnt += __popcll(__ballot(lane + 64 * r < n_cdf && e[r] < t));
                b = cnt < n_cdf ? cnt : n_cdf - 1;
            }
        } else {
        for (int i = lane; i < n_cdf; i += 64) cdf[i] = exp(pw[i] - m);
        wave_lds_sync();
        const int n_chunk = (n_cdf + 15) >> 4;
        double* ctot = cdf + n_cdf;
        for (int c = lane; c < n_chunk; c += 64) {
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
        if (lane == 0) {
            double off = 0.0;
            for (int c = 0; c < n_chunk; ++c) {
                const double t = ctot[c];
                ctot[c] = off;
                off = off + t;
            }
        }
        wave_lds_sync();
        for (int i = lane; i < n_cdf; i += 64) cdf[i] = ctot[i >> 4] + cdf[i];
        wave_lds_sync();
        const double total = cdf[n_cdf - 1];
        if (!(total > 0.0) || !(total < INFINITY)) {
            b = (int)(u_base * n_cdf);
            b = b < n_cdf ? b : n_cdf - 1;
        } else {  // first i with cdf[i] >= t, else last (cdf is monotone): binary search
            const double t = u_base * total;
            int lo = 0, hi = n_cdf - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (cdf[mid] >= t) hi = mid; else lo = mid + 1;
            }
            b = lo;
        }
        }
        if (lane == 0) s_base = b + p.pool_lo;
    }

    // ---- the scalars every term depends on, proposed by every lane for itself: proposed value of dim pair k ----
    const double eps = p.eps, eps2 = p.eps - (-p.eps);
    double cm = 0.0, cn = 0.0;  // snooker projection coefficients (set below)
    const long long S = p.N;
    const bool hier_b = p.family == FAM_HIER_BINOMIAL, hier_g = p.family == FAM_HIER_GAUSSIAN;
    struct PairIn {
        double2 t, a, b, c;  // current row, first / second partner, base row (or snooker's third row)
        double2 kk;          // hierarchical Binomial: counts of the pair's two subjects
        bool keep0, keep1;   // reset!: scalar outside the block of this sweep
    };
    auto load_pair = [&](int k, bool with_base, bool all_rows = false) -> PairIn {
        PairIn r;
        const int j0 = 2 * k;
        const bool has1 = j0 + 1 < D;
        auto ld2 = [&](const double* row) -> double2 {
            if (even) return *reinterpret_cast<const double2*>(row + j0);
            return make_double2(row[j0], has1 ? row[j0 + 1] : 0.0);
        };
        r.t = ld2(pt);
        r.a = r.t; r.b = r.t; r.c = r.t;
        r.keep0 = r.keep1 = false;
        if (p.mask) {  // reset! (crossover.jl:336-352)
            r.keep0 = !p.mask[j0];
            r.keep1 = has1 && !p.mask[j0 + 1];
        }
        r.kk = make_double2(0.0, 0.0);
        if (hier_b) {  // scalar 2 + s belongs to subject s (the log binomial coefficients are summed once, on the host: p.c2)
            const long long sA = (long long)j0 - 2, sB = sA + 1;
            if (sA >= 0 && sB < S && even)
                r.kk = *reinterpret_cast<const double2*>(p.data + sA);
            else {
                if (sA >= 0 && sA < S) r.kk.x = p.data[sA];
                if (has1 && sB >= 0 && sB < S) r.kk.y = p.data[sB];
            }
        }
        // a scalar pair wholly outside the block keeps its values: no partner rows needed (mutation ignores the mask)
        // (a snooker proposal still needs its rows: adjust_loglike's norms run over every scalar, crossover.jl:268-273)
        // (all_rows: the partner loads go out without waiting for the mask bytes -- the few general rounds of a fast sweep)
        const bool frozen = !all_rows && kind == 0 && r.keep0 && (r.keep1 || !has1);
        if ((kind == 0 || kind == 1) && !frozen) {
            r.a = ld2(Pa);
            r.b = ld2(Pb2);
            if (kind == 1) r.c = ld2(Pc);
            else if (with_base) r.c = ld2(Pbase);
        }
        return r;
    };
    U4 nz = {0, 0, 0, 0}, rc = nz;  // the noise / recombination block last drawn by this lane, and which one it is
    int nz_block = -1, rc_block = -1;
    bool base_on = false;
    auto cross = [&](double tj, double aj, double bj2, double cj, double uu) -> double {
        const double bj = -eps + eps2 * uu;  // b = Uniform(-eps, eps) crossover.jl:166
        if (kind == 1) {  // (Pt + gamma*(Pr1 - Pr2)) + b  crossover.jl:253
            const double dj = tj - aj;
            const double t1 = dj * cm - dj * cn;
            return (tj + t1 * g1) + bj;
        }
        const double t1 = aj - bj2;  // ((Pt + g1*(Pm-Pn)) + g2*(Pb-Pt)) + b  crossover.jl:168
        double t6 = tj + t1 * g1;
        if (base_on) {
            const double t4 = cj - tj;
            t6 = t6 + t4 * g2;
        }
        return t6 + bj;
    };
    // proposal of both scalars of pair k from loaded rows (recombination! and reset! applied)
    auto propose_pair = [&](int k, const PairIn& in, double& v0, double& v1) {
        const int j0 = 2 * k, j1 = j0 + 1;
        const bool has1 = j1 < D;
        v0 = in.t.x; v1 = in.t.y;
        if (kind == 3) return;
        const bool keep0 = in.keep0, keep1 = in.keep1;
        if (kind != 2 && keep0 && (keep1 || !has1)) return;  // outside the block: nothing to draw (mutation ignores the mask)
        // noise block k >> 1 covers the dim pairs 2(k >> 1), 2(k >> 1) + 1 (four scalars per block); a lane owns both pairs
        // of a block and visits them back to back, so the block is drawn once and kept
        if (nz_block != (k >> 1)) {
            nz = draw_block(p.seed, S_NOISE, p.sweep, (uint64_t)p.iter, eslot, (uint32_t)(k >> 1));
            nz_block = k >> 1;
        }
        const double u0 = u32unit((k & 1) ? nz.z : nz.x), u1 = u32unit((k & 1) ? nz.w : nz.y);
        if (kind == 2) {  // pt + Normal(0, sigma): mutation.jl:15-18 (rare sweep: out of line, see box_muller_outofline)
            const double2 z = box_muller_outofline((k & 1) ? nz.z : nz.x, (k & 1) ? nz.w : nz.y);
            v0 = in.t.x + p.sigma * z.x;
            v1 = in.t.y + p.sigma * z.y;
            return;
        }
        v0 = cross(in.t.x, in.a.x, in.b.x, in.c.x, u0);
        if (has1) v1 = cross(in.t.y, in.a.y, in.b.y, in.c.y, u1);
        if (p.kappa != 1.0) {  // recombination! crossover.jl:301-312
            if (rc_block != (k >> 1)) {
                rc = draw_block_outofline(p.seed, S_RECOMB, p.sweep, (uint64_t)p.iter, eslot, (uint32_t)(k >> 1));
                rc_block = k >> 1;
            }
            if (u32unit((k & 1) ? rc.z : rc.x) <= 1.0 - p.kappa) v0 = in.t.x;
            if (u32unit((k & 1) ? rc.w : rc.y) <= 1.0 - p.kappa) v1 = in.t.y;
        }
        if (keep0) v0 = in.t.x;
        if (keep1) v1 = in.t.y;
    };

    // This lane's noise blocks, in the order it visits them: tid, tid + WG, ...; block m = dim pairs 2m and 2m + 1
    const int n_blocks = (D + 3) >> 2;
    auto pair_at = [&](int i) -> int { return 2 * (tid + (i >> 1) * WG) + (i & 1); };
    auto ld2_at = [&](const double* row, int k) -> double2 {
        const int j0 = 2 * k;
        return even ? *reinterpret_cast<const double2*>(row + j0) : make_double2(row[j0], j0 + 1 < D ? row[j0 + 1] : 0.0);
    };
    // ---- which body serves a wave's 64 blocks (scalars [j_lo, j_lo + 256)): the branch-free one when they are all there,
    // share a table segment with a plain prior, have a subject each, and lie on one side of the sweep's block.  Decided
    // from the run-length tables in the kernarg -- scalar loads and SALU only, nothing to wait for -- and remembered for
    // the region [reg_lo, reg_hi) around the wave's position.  (No calls inside the branch-free body -- values live across
    // a call are spilled around it: recombination, mutation and the rarer prior families take the general body.) ----
    const bool fast_ok = even && (kind == 0 || kind == 1 || kind == 2) && (hier_b || hier_g) && p.kappa == 1.0 && p.n_mrun > 0;
    int reg_lo = 0, reg_hi = 0, reg_q = 0;
    bool reg_inb = false;
    // the uniform region around scalar j: [reg_lo, reg_hi) (empty when the segment's prior is not a plain one)
    auto find_region = [&](int j) {
        if (j >= reg_lo && j < reg_hi) return;
        int qq = 0, r = 0;
        for (int i = 1; i < p.n_seg; ++i) qq += (j >= p.seg_start[i]) ? 1 : 0;
        for (int i = 1; i < p.n_mrun; ++i) r += (j >= p.mrun_start[i]) ? 1 : 0;
        const int lo_s = p.seg_start[qq], hi_s = qq + 1 < p.n_seg ? p.seg_start[qq + 1] : D;
        const int lo_m = p.mrun_start[r], hi_m = r + 1 < p.n_mrun ? p.mrun_start[r + 1] : D;
        reg_q = qq;
        reg_inb = ((p.mrun_in >> r) & 1u) != 0;
        reg_lo = ((p.seg_plain >> qq) & 1u) ? (lo_s > lo_m ? lo_s : lo_m) : D;  // (other priors: an empty region)
        reg_hi = hi_s < hi_m ? hi_s : hi_m;
        reg_lo = reg_lo > 2 ? reg_lo : 2;  // scalar 2 + s belongs to subject s: a region holds scalars with subjects only
        reg_hi = (long long)reg_hi < S + 2 ? reg_hi : (int)(S + 2);
    };
    auto classify = [&](int j_lo, bool& inb, int& q) -> bool {
        const int j_hi = j_lo + 255;
        if (!(fast_ok && j_hi < D)) return false;
        find_region(j_lo);
        inb = reg_inb;
        q = reg_q;
        return j_lo >= reg_lo && j_hi < reg_hi;
    };
    // A round that is NOT wholly inside a region but overlaps one: the first round of a hierarchical row (its first two
    // scalars are the hyper-parameters) and the ragged last one.  The overlap [c_lo, c_hi) -- whole dim pairs -- runs in the
    // span loop's MASKED form (one round; lanes outside the overlap contribute nothing and store nothing), and only the
    // scalars outside it are left to the scalar-per-lane step.  The region is looked up at the round's last scalar
    // (a region that ends inside the round: at its first).
    auto classify_partial = [&](int j_lo, bool& inb, int& q, int& c_lo, int& c_hi) -> bool {
        if (!fast_ok || j_lo >= D) return false;
        const int j_end = j_lo + 256 < D ? j_lo + 256 : D;
        for (int probe = 0; probe < 2; ++probe) {
            find_region(probe ? j_lo : j_end - 1);
            c_lo = reg_lo > j_lo ? reg_lo : j_lo;
            c_hi = reg_hi < j_end ? reg_hi : j_end;
            if (c_lo < c_hi && ((c_lo | c_hi) & 1) == 0) {
                inb = reg_inb;
                q = reg_q;
                return true;
            }
        }
        return false;
    };
    // ---- the row move the previous particle of this workgroup left behind (persistent form) ----
    // The row sits in the LDS copy (`scr`): the accepted proposal, or -- refilled at the decision -- the current row of a
    // rejected particle.  Block m of it is written just before round m's proposal takes the LDS slot.  In the span loops (DEFER) through BUFFER instructions
    // whose descriptors carry the on / off state -- zero records = the hardware's range check drops the access -- so that
    // every round issues the same memory instructions whatever there is to write: the loop's waits stay `vmcnt(n)` with a
    // fixed n (a store under a branch would make the compiler wait for everything in flight, the stores included).
    const bool defer_on = prev && n_blocks <= 32 * WG;  // (prev_wbits notes 64 pairs = 32 rounds per lane)
    auto rsrc = [&](const double* base, bool on) {
        const uint64_t a = (uint64_t)(size_t)base;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        // (every input through readfirstlane: a descriptor the compiler cannot PROVE wave-uniform gets a waterfall loop -- a
        // dozen instructions and a branch -- around each access, cdna_hip_programming.md T20)
        const int nrec = __builtin_amdgcn_readfirstlane(on ? D * 8 : 0);
        return __builtin_amdgcn_make_buffer_rsrc((void*)(size_t)(((uint64_t)hi << 32) | lo), 0, nrec, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t R_hist = rsrc(prev_hrow, defer_on && prev_hrow != nullptr);
    const __amdgpu_buffer_rsrc_t R_theta = rsrc(prev_trow, defer_on && prev_acc);
    constexpr unsigned kOff = 0x7ffffff0u;  // a byte offset past every row: this lane's store is dropped
    // the same for one block outside the span loops (a MASKED round, a round left to the scalar-per-lane step): plain code
    auto flush_block = [&](int i) {
        const int m = tid + i * WG;
        if (!defer_on || m >= n_blocks) return;
        for (int h = 0; h < 2; ++h) {
            const int j0 = 4 * m + 2 * h;
            if (j0 >= D) break;
            const double2 v = *reinterpret_cast<const double2*>(scr + j0);
            const bool in_block = !prev_masked || (p.mask[j0] | p.mask[j0 + 1]) != 0;
            if (prev_acc && in_block) *reinterpret_cast<double2*>(prev_trow + j0) = v;  // utilities.jl:204
            if (prev_hrow) *reinterpret_cast<double2*>(prev_hrow + j0) = v;             // utilities.jl:170-180
        }
    };
    // ... and the blocks of it that lie in rounds the span loops do not take in their full form (MASKED rounds, rounds left to
    // the scalar-per-lane step) go out NOW, in plain code, before the first barrier: another lane writes the orphan scalars of
    // such a round into the LDS copy later on, and here the stores are far from the span loops' waits.
    if (defer_on) {
        const int w0f = wave * 64;
        const int n_itf = n_blocks > w0f ? (n_blocks - w0f + WG - 1) / WG : 0;
        for (int i = 0; i < n_itf; ++i) {
            bool inb_;
            int q_;
            if (classify(4 * (w0f + i * WG), inb_, q_)) {
                const int skip = (reg_hi - 256 - 4 * (w0f + i * WG)) / (4 * WG);  // the rounds that stay inside the region
                i += skip > 0 ? skip : 0;
            } else
                flush_block(i);
        }
    }
    // snooker: project(Pm,Pd), project(Pn,Pd) need whole-row dot products first (utilities.jl:239-246)
    if (kind == 1) {
        double vm = 0.0, vn = 0.0, vd = 0.0;
        for (int k = tid; 2 * k < D; k += WG)
            for (int e = 0; e < 2; ++e) {
                const int j = 2 * k + e;
                if (j < D) {
                    const double dj = pt[j] - Pa[j];
                    vm += Pb2[j] * dj; vn += Pc[j] * dj; vd += dj * dj;
                }
            }
        vm = subgroup_sum(vm, 64); vn = subgroup_sum(vn, 64); vd = subgroup_sum(vd, 64);
        if (lane == 0) { s_red[0][wave] = vm; s_red[1][wave] = vn; s_red[2][wave] = vd; }
    }
    DEMC_STAMP(0);  // wave 0: base pick done
    if (seg_lane) reinterpret_cast<double*>(s_seg)[tid] = seg_word;
    lds_barrier();  // base pick (wave 0), snooker partial sums, segment table
    DEMC_STAMP_AT(1, 64, DEMC_STAMP_NOW());
    if (kind == 1) {
        double vm = 0.0, vn = 0.0, vd = 0.0;
        // the tree k_propose's group_sum uses: pairs of waves, then pairs of pairs
        auto tree = [&](const double* s) {
            const double lo = (s[0] + s[1]) + (s[2] + s[3]);
            if constexpr (WG == 384) return lo + (s[4] + s[5]);
            else return WG > 256 ? lo + ((s[4] + s[5]) + (s[6] + s[7])) : lo;
        };
        vm = tree(s_red[0]); vn = tree(s_red[1]); vd = tree(s_red[2]);
        cm = vm / vd; cn = vn / vd;
        lds_barrier();  // s_red is reused by the final reduction
    }
    if (use_base) {
        i2 = s_base;
        Pbase = (p.base_theta ? p.base_theta + (size_t)g * Np * D : grows) + (size_t)i2 * D;
        base_on = true;
    }

    // ---- the scalars every term depends on (the hyper-parameters theta'[0], the observation sd, the prior scales
    // theta'[ref]): one lane proposes each (a proposal depends on the scalar's index only), LDS hands them round ----
    {
        int need = -1;
        if (tid == 0) need = (hier_b || hier_g) ? 0 : -1;
        else if (tid == 1) need = hier_g ? 2 + (int)S : -1;
        else if (tid < 2 + p.n_seg && s_seg[tid - 2].t.kind == PR_NORMAL_REF) need = s_seg[tid - 2].t.ref;
        if (need >= 0) {
            const PairIn in = load_pair(need >> 1, base_on);
            double a0, a1;
            propose_pair(need >> 1, in, a0, a1);
            const double val = (need & 1) ? a1 : a0;
            if (tid < 2)
                s_hyp[tid] = val;
            else {  // Normal(a, theta'[ref]) priors (hierarchical scale): 1/scale and log scale per table segment, once
                s_ref[0][tid - 2] = 1.0 / val;
                s_ref[1][tid - 2] = log(val);
            }
        }
    }
    DEMC_STAMP(4);  // hyper-parameter scalars proposed (wave 0)
    lds_barrier();
    // family constants of the fused likelihood term
    double mu0 = 0.0, sg_obs = 1.0, lsg_obs = 0.0, isg_obs = 1.0;
    if (hier_b || hier_g) mu0 = s_hyp[0];
    if (hier_g) {
        sg_obs = s_hyp[1];
        lsg_obs = log(sg_obs);
        isg_obs = 1.0 / sg_obs;
    }
    const double n_bin = p.c0;
    DEMC_LR_NEXT(2);
    int oob = 0;
    double prior = 0.0, like = 0.0, s1 = 0.0, s2 = 0.0;
    unsigned long long wbits = 0;  // which of this lane's pairs lie (partly) inside the block of the sweep
    unsigned lbits = 0;            // which of its rounds were left to the scalar-per-lane step (their mask is read again)
    int n_done = 0;
    auto process = [&](int k, const PairIn& cur) {
        const int j0 = 2 * k;
        const bool has1 = j0 + 1 < D;
        const double k0 = cur.kk.x, k1 = cur.kk.y;
        if (!(cur.keep0 && (cur.keep1 || !has1)) && n_done < 64) wbits |= 1ull << n_done;
        ++n_done;
        int q0 = 0, q1 = 0;  // table segment of each scalar = number of segment starts at or below it
        {
            const int j1 = has1 ? j0 + 1 : j0;
            for (int i = 1; i < p.n_seg; ++i) {
                const int st = s_seg[i].start;
                q0 += (j0 >= st) ? 1 : 0;
                q1 += (j1 >= st) ? 1 : 0;
            }
        }
        const DimTab tab0 = s_seg[q0].t, tab1 = s_seg[q1].t;
        double v0, v1;
        propose_pair(k, cur, v0, v1);
        if (kind == 1) {  // adjust_loglike norms (crossover.jl:268-273)
            const double a0 = v0 - cur.a.x, b0 = cur.t.x - cur.a.x;
            s1 += a0 * a0; s2 += b0 * b0;
            if (has1) {
                const double a1 = v1 - cur.a.y, b1 = cur.t.y - cur.a.y;
                s1 += a1 * a1; s2 += b1 * b1;
            }
        }
        for (int e = 0; e < 2; ++e) {
            if (e == 1 && !has1) break;
            const double v = e ? v1 : v0;
            const DimTab t = e ? tab1 : tab0;
            oob |= !(v >= t.lo && v <= t.hi);  // in_bounds utilities.jl:70-78 (NaN fails)
            if (p.fitness_kind == 0 && t.kind != PR_FLAT) {
                const int q = e ? q1 : q0;
                if (t.kind == PR_NORMAL_REF) {  // Normal(a, theta'[ref]): the bulk of a hierarchical row
                    const double z = (v - t.a) * s_ref[0][q];
                    prior += -(z * z + kLog2Pi) / 2.0 - s_ref[1][q];
                } else if (t.kind == PR_NORMAL) {
                    const double z = (v - t.a) * t.b;
                    prior += t.c - 0.5 * (z * z);
                } else
                    prior += prior_term_ref_outofline(&s_seg[q].t, v, s_ref[0][q], s_ref[1][q]);  // (the entry in LDS: no local copy)
            }
            // likelihood term of the subject behind this scalar
            const long long s = (long long)j0 + e - 2;
            if (s >= 0 && s < S) {
                if (hier_b) {  // k log p + (n-k) log(1-p), p = logistic(eta): one softplus per subject
                    const double eta = mu0 + v, kk = e ? k1 : k0;
                    like += -n_bin * sp_of(-eta) - (n_bin - kk) * eta;
                } else if (hier_g) {  // Hierarchical_Example.jl:36-44: p.d observations per subject
                    const double mu = mu0 + v;
                    const int n = p.d;
                    double l = 0.0;
                    for (int i = 0; i < n; ++i) {
                        const double z = (p.data[s * n + i] - mu) * isg_obs;
                        l += -(z * z + kLog2Pi) / 2.0 - lsg_obs;
                    }
                    like += l;
                }
            }
        }
        if (even)
            *reinterpret_cast<double2*>(scr + j0) = make_double2(v0, v1);
        else {
            scr[j0] = v0;
            if (has1) scr[j0 + 1] = v1;
        }
        if (p.write_prop) {
            p.prop[slot * D + j0] = v0;
            if (has1) p.prop[slot * D + j0 + 1] = v1;
        }
    };
    // ---- the pass over the row, wave by wave in SPANS: a wave's rounds i (blocks wave*64 + i*WG + lane) that stay inside one
    // uniform region run in a tight loop compiled for what the region needs -- which rows, which proposal form, whether
    // there are counts to load -- so that every round issues the same loads in the same order: the rows of the NEXT block go
    // out first, and the compiler can wait for exactly the current block's (s_waitcnt vmcnt(n) with n = loads of a round)
    // while they fly.  (With loads under run-time conditions it must wait for all of them, and did: the prefetch of round 1
    // overlapped nothing.)  The last round of a span re-reads its own block in place of a next one.  Rounds outside a
    // region (the first and the last blocks of a hierarchical row) take the general per-pair body, one block at a time. ----
    auto uni = [](double x) {  // a wave-uniform value, held in SGPRs
        return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
    };
    const bool prior_on = p.fitness_kind == 0;
    double T_lo = 0.0, T_hi = 0.0, T_a = 0.0, T_b = 0.0, T_c = 0.0, R_inv = 0.0, R_log = 0.0;  // the region's table entry
    int T_kind = PR_FLAT;
    int n_fast_blocks__ = 0;  // (diagnostic build)
    enum { M_FROZEN = 0, M_SNK_FROZEN = 1, M_DE = 2, M_DE_BASE = 3, M_SNK = 4, M_MUT = 5 };
    struct Blk {
        double2 t0, t1, k0, k1, a0, a1, b0, b1, c0, c1;
    };
    // MASKED: ONE round whose lanes are only partly inside the region (classify_partial): dim pair e of block m counts iff
    // [4m + 2e, 4m + 2e + 2) lies in [c_lo, c_hi); a lane past the end of the row works on the last block, every address
    // stays inside the row, what does not count is selected away (no branches: the loads keep their fixed order).
    auto run_span = [&](auto mode_c, auto hb_c, auto masked_c, auto defer_c, int i0, int i1, int c_lo, int c_hi) {
        constexpr int MODE = decltype(mode_c)::value;
        constexpr bool HB = decltype(hb_c)::value;
        constexpr bool MASKED = decltype(masked_c)::value;
        constexpr bool DEFER = decltype(defer_c)::value;
        constexpr bool MOVES = MODE == M_DE || MODE == M_DE_BASE || MODE == M_SNK;
        constexpr bool ROW_A = MODE != M_FROZEN && MODE != M_MUT, ROW_B = MODE == M_DE || MODE == M_DE_BASE, ROW_C = MODE == M_DE_BASE;
        auto load = [&](int m) -> Blk {
            Blk r;
            if constexpr (MASKED) m = m < n_blocks ? m : n_blocks - 1;
            const size_t o = 4 * (size_t)m;
            // second dim pair of the block / the counts of the first: addresses that fall off the row (the last block of a row
            // with D = 2 mod 4; block 0, whose first pair has no subjects) are replaced by the neighbouring pair's -- masked away
            const size_t o2 = (!MASKED || o + 2 < (size_t)D) ? o + 2 : o;
            const size_t ok = (!MASKED || o >= 2) ? o - 2 : 0;
            // (the own row is read by this workgroup alone, once -- but loading it non-temporally, so that it would not push the
            // partner rows out of L2, made the whole cfg4 9 % SLOWER; non-temporal history stores changed nothing)
            r.t0 = *reinterpret_cast<const double2*>(pt + o);
            r.t1 = *reinterpret_cast<const double2*>(pt + o2);
            if constexpr (HB) {  // scalar 2 + s belongs to subject s
                r.k0 = *reinterpret_cast<const double2*>(p.data + ok);
                r.k1 = *reinterpret_cast<const double2*>(p.data + o);
            }
            if constexpr (ROW_A) {
                r.a0 = *reinterpret_cast<const double2*>(Pa + o);
                r.a1 = *reinterpret_cast<const double2*>(Pa + o2);
            }
            if constexpr (ROW_B) {
                r.b0 = *reinterpret_cast<const double2*>(Pb2 + o);
                r.b1 = *reinterpret_cast<const double2*>(Pb2 + o2);
            }
            if constexpr (ROW_C) {
                r.c0 = *reinterpret_cast<const double2*>(Pbase + o);
                r.c1 = *reinterpret_cast<const double2*>(Pbase + o2);
            }
            return r;
        };
        // (DEMC_LR_PREFETCH = 2 keeps two blocks in flight beside the one being worked on.  Measured on the whole cfg4 and on
        // its share: no difference to one block ahead -- 168 us per launch either way -- for 22 more VGPRs: the span loops
        // wait for arithmetic, not for rows.  A round is 351 VALU instructions of which 38 run at a quarter of the FP64
        // rate (20 v_mad_u64_u32 of the Philox block; rndne, cvt, ldexp and rcp of four softplus), ~1860 cycles per wave.)
        Blk cur = load(tid + i0 * WG);
#if DEMC_LR_PREFETCH >= 2
        Blk nx1 = load(tid + (i0 + 1 < i1 ? i0 + 1 : i0) * WG);
#endif
        for (int i = i0; i < i1; ++i) {
            int m = tid + i * WG;
            bool vp0 = true, vp1 = true;  // which of the block's two dim pairs count (MASKED)
            if constexpr (MASKED) {
                vp0 = 4 * m >= c_lo && 4 * m + 2 <= c_hi;
                vp1 = 4 * m + 2 >= c_lo && 4 * m + 4 <= c_hi;
                m = m < n_blocks ? m : n_blocks - 1;
            }
#if DEMC_LR_PREFETCH >= 2
            const Blk nxt = nx1;
            nx1 = load(tid + (i + 2 < i1 ? i + 2 : i1 - 1) * WG);
#else
            const Blk nxt = load(tid + (i + 1 < i1 ? i + 1 : i) * WG);
#endif
            ++n_fast_blocks__;
            double v0 = cur.t0.x, v1 = cur.t0.y, v2 = cur.t1.x, v3 = cur.t1.y;
            if constexpr (MODE == M_MUT) {  // pt + Normal(0, sigma) on every scalar: mutation.jl:15-18 (a group in ten takes this
                // sweep -- and its workgroups set the length of the launch, so it runs in the span loops like the others)
                const U4 nb = draw_block(p.seed, S_NOISE, p.sweep, (uint64_t)p.iter, eslot, (uint32_t)m);
                const double2 za = box_muller(nb.x, nb.y), zb = box_muller(nb.z, nb.w);
                v0 = v0 + p.sigma * za.x;
                v1 = v1 + p.sigma * za.y;
                v2 = v2 + p.sigma * zb.x;
                v3 = v3 + p.sigma * zb.y;
            }
            if constexpr (MOVES) {
                const U4 nb = draw_block(p.seed, S_NOISE, p.sweep, (uint64_t)p.iter, eslot, (uint32_t)m);
                // b = Uniform(-eps, eps) crossover.jl:166
                const double b0 = -eps + eps2 * u32unit(nb.x), b1 = -eps + eps2 * u32unit(nb.y);
                const double b2 = -eps + eps2 * u32unit(nb.z), b3 = -eps + eps2 * u32unit(nb.w);
                if constexpr (MODE == M_SNK) {  // (Pt + gamma*(Pr1 - Pr2)) + b  crossover.jl:253
                    const double d0 = v0 - cur.a0.x, d1 = v1 - cur.a0.y, d2 = v2 - cur.a1.x, d3 = v3 - cur.a1.y;
                    v0 = (v0 + (d0 * cm - d0 * cn) * g1) + b0;
                    v1 = (v1 + (d1 * cm - d1 * cn) * g1) + b1;
                    v2 = (v2 + (d2 * cm - d2 * cn) * g1) + b2;
                    v3 = (v3 + (d3 * cm - d3 * cn) * g1) + b3;
                } else if constexpr (MODE == M_DE_BASE) {  // ((Pt + g1*(Pm-Pn)) + g2*(Pb-Pt)) + b  crossover.jl:168
                    v0 = ((v0 + (cur.a0.x - cur.b0.x) * g1) + (cur.c0.x - v0) * g2) + b0;
                    v1 = ((v1 + (cur.a0.y - cur.b0.y) * g1) + (cur.c0.y - v1) * g2) + b1;
                    v2 = ((v2 + (cur.a1.x - cur.b1.x) * g1) + (cur.c1.x - v2) * g2) + b2;
                    v3 = ((v3 + (cur.a1.y - cur.b1.y) * g1) + (cur.c1.y - v3) * g2) + b3;
                } else {
                    v0 = (v0 + (cur.a0.x - cur.b0.x) * g1) + b0;
                    v1 = (v1 + (cur.a0.y - cur.b0.y) * g1) + b1;
                    v2 = (v2 + (cur.a1.x - cur.b1.x) * g1) + b2;
                    v3 = (v3 + (cur.a1.y - cur.b1.y) * g1) + b3;
                }
            }
            // (a MASKED round leaves the in-block test of its pairs to the row moves, which read the mask again: lbits)
            if (!MASKED && MOVES && i < 32) wbits |= 3ull << (2 * i);  // (pairs 2i, 2i + 1 of this lane's sequence)
            if constexpr (MODE == M_SNK || MODE == M_SNK_FROZEN) {  // adjust_loglike norms (crossover.jl:268-273), scalar by scalar
                const double a0 = v0 - cur.a0.x, b0 = cur.t0.x - cur.a0.x, a1 = v1 - cur.a0.y, b1 = cur.t0.y - cur.a0.y;
                const double a2 = v2 - cur.a1.x, b2 = cur.t1.x - cur.a1.x, a3 = v3 - cur.a1.y, b3 = cur.t1.y - cur.a1.y;
                if constexpr (MASKED) {
                    s1 += vp0 ? a0 * a0 : 0.0; s2 += vp0 ? b0 * b0 : 0.0;
                    s1 += vp0 ? a1 * a1 : 0.0; s2 += vp0 ? b1 * b1 : 0.0;
                    s1 += vp1 ? a2 * a2 : 0.0; s2 += vp1 ? b2 * b2 : 0.0;
                    s1 += vp1 ? a3 * a3 : 0.0; s2 += vp1 ? b3 * b3 : 0.0;
                } else {
                    s1 += a0 * a0; s2 += b0 * b0;
                    s1 += a1 * a1; s2 += b1 * b1;
                    s1 += a2 * a2; s2 += b2 * b2;
                    s1 += a3 * a3; s2 += b3 * b3;
                }
            }
            {  // in_bounds utilities.jl:70-78 (NaN fails)
                const int o01 = (int)!(v0 >= T_lo && v0 <= T_hi) | (int)!(v1 >= T_lo && v1 <= T_hi);
                const int o23 = (int)!(v2 >= T_lo && v2 <= T_hi) | (int)!(v3 >= T_lo && v3 <= T_hi);
                oob |= (vp0 ? o01 : 0) | (vp1 ? o23 : 0);
            }
            if (prior_on && T_kind != PR_FLAT) {
                double q0, q1, q2, q3;
                if (T_kind == PR_NORMAL_REF) {  // Normal(a, theta'[ref]): the bulk of a hierarchical row
                    const double z0 = (v0 - T_a) * R_inv, z1 = (v1 - T_a) * R_inv, z2 = (v2 - T_a) * R_inv, z3 = (v3 - T_a) * R_inv;
                    q0 = -(z0 * z0 + kLog2Pi) / 2.0 - R_log;
                    q1 = -(z1 * z1 + kLog2Pi) / 2.0 - R_log;
                    q2 = -(z2 * z2 + kLog2Pi) / 2.0 - R_log;
                    q3 = -(z3 * z3 + kLog2Pi) / 2.0 - R_log;
                } else {  // PR_NORMAL
                    const double z0 = (v0 - T_a) * T_b, z1 = (v1 - T_a) * T_b, z2 = (v2 - T_a) * T_b, z3 = (v3 - T_a) * T_b;
                    q0 = T_c - 0.5 * (z0 * z0);
                    q1 = T_c - 0.5 * (z1 * z1);
                    q2 = T_c - 0.5 * (z2 * z2);
                    q3 = T_c - 0.5 * (z3 * z3);
                }
                if constexpr (MASKED) {
                    prior += vp0 ? q0 : 0.0; prior += vp0 ? q1 : 0.0; prior += vp1 ? q2 : 0.0; prior += vp1 ? q3 : 0.0;
                } else {
                    prior += q0; prior += q1; prior += q2; prior += q3;
                }
            }
            if constexpr (HB) {  // k log p + (n-k) log(1-p), p = logistic(eta): four independent softplus chains
                const double e0 = mu0 + v0, e1 = mu0 + v1, e2 = mu0 + v2, e3 = mu0 + v3;
                const double l0 = -n_bin * sp_of(-e0) - (n_bin - cur.k0.x) * e0;
                const double l1 = -n_bin * sp_of(-e1) - (n_bin - cur.k0.y) * e1;
                const double l2 = -n_bin * sp_of(-e2) - (n_bin - cur.k1.x) * e2;
                const double l3 = -n_bin * sp_of(-e3) - (n_bin - cur.k1.y) * e3;
                if constexpr (MASKED) {
                    like += vp0 ? l0 : 0.0; like += vp0 ? l1 : 0.0; like += vp1 ? l2 : 0.0; like += vp1 ? l3 : 0.0;
                } else {
                    like += l0; like += l1; like += l2; like += l3;
                }
            } else {  // Hierarchical_Example.jl:36-44: p.d observations per subject
                const int n = p.d;
                const double vv[4] = {v0, v1, v2, v3};
                for (int e = 0; e < 4; ++e) {
                    if (MASKED && !(e < 2 ? vp0 : vp1)) continue;
                    const double mu = mu0 + vv[e];
                    const long long sb = (long long)4 * m + e - 2;
                    double l = 0.0;
                    for (int o = 0; o < n; ++o) {
                        const double z = (p.data[sb * n + o] - mu) * isg_obs;
                        l += -(z * z + kLog2Pi) / 2.0 - lsg_obs;
                    }
                    like += l;
                }
            }
            if constexpr (DEFER) {  // (full rounds only: both dim pairs of block m exist)
                const v4u w0 = *reinterpret_cast<const v4u*>(scr + 4 * m), w1 = *reinterpret_cast<const v4u*>(scr + 4 * m + 2);
                // theta row: the pairs inside the block of the sweep (reset!: the others did not change); history row: all
                const bool in0 = !prev_masked || ((prev_wbits >> (2 * i)) & 1ull), in1 = !prev_masked || ((prev_wbits >> (2 * i + 1)) & 1ull);
                __builtin_amdgcn_raw_buffer_store_b128(w0, R_theta, in0 ? (unsigned)m * 32u : kOff, 0, 0);        // utilities.jl:204
                __builtin_amdgcn_raw_buffer_store_b128(w1, R_theta, in1 ? (unsigned)m * 32u + 16u : kOff, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(w0, R_hist, (unsigned)m * 32u, 0, 0);                       // utilities.jl:170-180
                __builtin_amdgcn_raw_buffer_store_b128(w1, R_hist, (unsigned)m * 32u + 16u, 0, 0);
            }
            if (vp0) *reinterpret_cast<double2*>(scr + 4 * m) = make_double2(v0, v1);
            if (vp1) *reinterpret_cast<double2*>(scr + 4 * m + 2) = make_double2(v2, v3);
            cur = nxt;
        }
    };
    {
        const int w0 = wave * 64;
        const int n_it = n_blocks > w0 ? (n_blocks - w0 + WG - 1) / WG : 0;  // rounds of this wave
        int i = 0;
        while (i < n_it) {
            if (i == 1) DEMC_STAMP_AT(7, 0, DEMC_STAMP_NOW());  // wave 0: first round done
            const int j_lo = 4 * (w0 + i * WG);
            bool inb = false;
            int q = 0;
            int c_lo = 0, c_hi = 0;
            const bool full = classify(j_lo, inb, q);
            const bool part = !full && classify_partial(j_lo, inb, q, c_lo, c_hi);
            if (full || part) {
                int i1 = i + 1;
                if (full) {
                    i1 = i + (reg_hi - 256 - j_lo) / (4 * WG) + 1;  // rounds that stay inside the region
                    i1 = i1 < n_it ? i1 : n_it;
                } else if (i < 32)
                    lbits |= 1u << i;  // its pairs' in-block test is made again by the row moves (mask bytes)
                const DimTab t = s_seg[q].t;
                T_lo = uni(t.lo); T_hi = uni(t.hi); T_a = uni(t.a); T_b = uni(t.b); T_c = uni(t.c);
                T_kind = __builtin_amdgcn_readfirstlane(t.kind);
                R_inv = uni(s_ref[0][q]); R_log = uni(s_ref[1][q]);
                if (i + 1 < n_it) DEMC_STAMP_AT(8, WG - 64, DEMC_STAMP_NOW());  // last wave: entering its span
                const int mode = kind == 2 ? M_MUT : kind == 1 ? (inb ? M_SNK : M_SNK_FROZEN) : (inb ? (base_on ? M_DE_BASE : M_DE) : M_FROZEN);
                using std::integral_constant;
                auto go = [&](auto hb_c, auto masked_c, auto defer_c) {
                    if (mode == M_FROZEN) run_span(integral_constant<int, M_FROZEN>(), hb_c, masked_c, defer_c, i, i1, c_lo, c_hi);
                    else if (mode == M_DE_BASE) run_span(integral_constant<int, M_DE_BASE>(), hb_c, masked_c, defer_c, i, i1, c_lo, c_hi);
                    else if (mode == M_DE) run_span(integral_constant<int, M_DE>(), hb_c, masked_c, defer_c, i, i1, c_lo, c_hi);
                    else if (mode == M_SNK) run_span(integral_constant<int, M_SNK>(), hb_c, masked_c, defer_c, i, i1, c_lo, c_hi);
                    else if (mode == M_MUT) run_span(integral_constant<int, M_MUT>(), hb_c, masked_c, defer_c, i, i1, c_lo, c_hi);
                    else run_span(integral_constant<int, M_SNK_FROZEN>(), hb_c, masked_c, defer_c, i, i1, c_lo, c_hi);
                };
                using TT = integral_constant<bool, true>;
                using FF = integral_constant<bool, false>;
                if (!full) {  // a MASKED round (its block of a pending row went out in the prologue)
                    if (hier_b) go(TT(), TT(), FF()); else go(FF(), TT(), FF());
                } else if (defer_on && i1 <= 32) {
                    if (hier_b) go(TT(), FF(), TT()); else go(FF(), FF(), TT());
                } else {
                    if (hier_b) go(TT(), FF(), FF()); else go(FF(), FF(), FF());
                }
                i = i1;
            } else if (fast_ok) {  // a round at the edge of a region: left to the scalar-per-lane step below
                if (i < 32) lbits |= 1u << i;
                ++i;
            } else {
                const int m = tid + i * WG;
                n_done = 2 * i;
#pragma nounroll
                for (int h = 0; h < 2; ++h) {
                    const int k = 2 * m + h;
                    if (2 * k < D) {
                        const PairIn in = load_pair(k, base_on);
                        process(k, in);
                    }
                }
                ++i;
            }
        }
    }
    DEMC_LR_NEXT(3);
    DEMC_STAMP_AT(15, 64, DEMC_STAMP_NOW());      // spans done
    DEMC_STAMP_AT(13, 0, DEMC_STAMP_NOW());       // ... by wave 0
    DEMC_STAMP_AT(14, WG - 64, DEMC_STAMP_NOW());  // ... by the last wave
    // ---- the rounds at the edges of the regions (in a hierarchical row: the 256 scalars around the hyper-parameters and
    // the ragged end), ONE SCALAR PER LANE: every term of a scalar by the lane that holds it, whatever its table entry, its
    // side of the block and its subject.  Rounds are numbered along the row (round R = scalars [256 R, 256 R + 256), the
    // R-th of them in wave R % (WG/64)); the left-over ones alternate between the two halves of the workgroup, so a row
    // with two of them -- head and tail -- costs every wave one step. ----
    if (fast_ok) {
        const int n_rounds = (n_blocks + 63) >> 6;
        int R = 0, r = 0;
        while (R < n_rounds) {
            bool inb;
            int q;
            if (classify(256 * R, inb, q)) {
                R = reg_hi >> 8;  // first round not wholly below the region's end
                continue;
            }
            // a round the span loops took in their MASKED form: only what lies outside the overlap is left (in a hierarchical
            // row: the two hyper-parameters of the first round, nothing of the last)
            int c_lo = 0, c_hi = 0;
            const bool part = classify_partial(256 * R, inb, q, c_lo, c_hi);
            const int n_left = (256 * R + 256 < D ? 256 : D - 256 * R) - (part ? c_hi - c_lo : 0);
            if (n_left <= 0) {
                ++R;
                continue;
            }
            if ((r & 1) == (tid >= WG / 2 ? 1 : 0)) {
                for (int jj = tid % (WG / 2); jj < 256 && 256 * R + jj < D; jj += WG / 2) {
                    const int j = 256 * R + jj;
                    if (part && j >= c_lo && j < c_hi) continue;
                    int qj = 0;
                    for (int i = 1; i < p.n_seg; ++i) qj += (j >= s_seg[i].start) ? 1 : 0;
                    const bool keep = p.mask ? !p.mask[j] : false;  // reset! (crossover.jl:336-352)
                    const double tj = pt[j], aj = Pa[j], bj2 = Pb2[j], cj = kind == 0 && base_on ? Pbase[j] : tj;  // (mutation: all Pt)
                    const long long sj = (long long)j - 2;  // the subject behind the scalar
                    const bool subj = sj >= 0 && sj < S;
                    const double kj = hier_b && subj ? p.data[sj] : 0.0;
                    const U4 nb = draw_block(p.seed, S_NOISE, p.sweep, (uint64_t)p.iter, eslot, (uint32_t)(j >> 2));
                    const uint32_t wj = (j & 2) ? ((j & 1) ? nb.w : nb.z) : ((j & 1) ? nb.y : nb.x);
                    double v;
                    if (kind == 2) {  // mutation ignores the block (mutation.jl:15-18)
                        const double2 z = box_muller_outofline((j & 2) ? nb.z : nb.x, (j & 2) ? nb.w : nb.y);
                        v = tj + p.sigma * ((j & 1) ? z.y : z.x);
                    } else
                        v = keep ? tj : cross(tj, aj, bj2, cj, u32unit(wj));
                    if (kind == 1) {  // adjust_loglike norms (crossover.jl:268-273)
                        const double a0 = v - aj, b0 = tj - aj;
                        s1 += a0 * a0; s2 += b0 * b0;
                    }
                    const DimTab* tb = &s_seg[qj].t;
                    oob |= !(v >= tb->lo && v <= tb->hi);  // in_bounds utilities.jl:70-78 (NaN fails)
                    if (prior_on && tb->kind != PR_FLAT) prior += prior_term_ref_outofline(tb, v, s_ref[0][qj], s_ref[1][qj]);
                    if (subj) {
                        if (hier_b) {
                            const double eta = mu0 + v;
                            like += -n_bin * sp_of(-eta) - (n_bin - kj) * eta;
                        } else {
                            const double mu = mu0 + v;
                            const int n = p.d;
                            double l = 0.0;
                            for (int o = 0; o < n; ++o) {
                                const double z = (p.data[sj * n + o] - mu) * isg_obs;
                                l += -(z * z + kLog2Pi) / 2.0 - lsg_obs;
                            }
                            like += l;
                        }
                    }
                    scr[j] = v;
                }
            }
            ++r;
            ++R;
        }
    }
    DEMC_STAMP_AT(5, 64, DEMC_STAMP_NOW());  // the pass over the row done
    DEMC_STAMP_AT(2, 64, n_fast_blocks__);
    DEMC_STAMP_AT(3, 64, n_done);
    DEMC_LR_NEXT(4);
    // ---- one reduction for everything: waves on the DPP network, then the fixed tree over the waves through LDS ----
    prior = subgroup_sum(prior, 64); like = subgroup_sum(like, 64); oob = subgroup_sum(oob, 64);
    if (kind == 1) { s1 = subgroup_sum(s1, 64); s2 = subgroup_sum(s2, 64); }
    if (lane == 0) {
        s_red[0][wave] = prior; s_red[1][wave] = like; s_red[2][wave] = s1; s_red[3][wave] = s2;
        s_redi[wave] = oob;
    }
    lds_barrier();
    // (trace / per-phase forms: the proposals go to HBM; the span loops leave that to this copy of the row held in LDS)
    if (fast_ok && p.write_prop)
        for (int j = tid; j < D; j += WG) p.prop[slot * D + j] = scr[j];
    auto tree = [&](const double* s) {
        const double lo = (s[0] + s[1]) + (s[2] + s[3]);
        if constexpr (WG == 384) return lo + (s[4] + s[5]);
        else return WG > 256 ? lo + ((s[4] + s[5]) + (s[6] + s[7])) : lo;
    };
    prior = tree(s_red[0]); like = tree(s_red[1]);
    if (hier_b) like = p.c2 + like;  // + sum_s log C(n, k_s): data-only, summed once at demc_set_model
    oob = 0;
    for (int i = 0; i < WG / 64; ++i) oob |= s_redi[i];
    double adj = 0.0;
    if (kind == 1) adj = (double)(D - 1) * (0.5 * log(tree(s_red[2])) - 0.5 * log(tree(s_red[3])));

    DEMC_STAMP_AT(6, 64, DEMC_STAMP_NOW());  // reductions done
    // ---- compute_posterior! + mh_update! + store_samples! ----
    double wp;
    if (p.fitness_kind == 1)
        wp = oob ? (p.update_kind == 1 ? -INFINITY : INFINITY) : like;
    else
        wp = oob ? -INFINITY : prior + like;
    const int acc = decide_mh(p.mode, p.update_kind, u_acc, wp, w_cur, adj);
    if (tid == 0) {
        if (acc) p.weight[slot] = wp;
        if (p.trace) {
            p.tr_idx[slot * 4 + 0] = kind; p.tr_idx[slot * 4 + 1] = i0;
            p.tr_idx[slot * 4 + 2] = i1; p.tr_idx[slot * 4 + 3] = i2;
            p.tr_w[slot] = wp; p.tr_acc[slot] = (unsigned char)acc; p.prop_adj[slot] = adj;
        }
        if (p.store_row >= 0) {
            const size_t hrow = (size_t)p.store_row * p.P + slot;
            if (p.update_kind == 0 && p.mode == MODE_STEP) {  // utilities.jl:207-208
                p.acc_hist[hrow] = (unsigned char)acc;
                p.lp_hist[hrow] = acc ? wp : w_cur;
            }
            p.id_hist[hrow] = (int)p.id[slot];
        }
    }
    DEMC_LR_NEXT(5);
    double* trow = p.theta + slot * D;
    double* hrow = (p.store_row >= 0) ? p.hist + ((size_t)p.store_row * p.P + slot) * p.hist_ld : nullptr;
    // in a block sweep an accepted crossover proposal differs from the row only inside the block: write only those
    const bool masked = acc && p.mask && kind != 2 && kind != 3;
    // A whole row to move and another particle to follow in this workgroup: its span loops do it (see `pend` at the top).
    // (The LDS copy is complete for an accepted particle -- frozen scalars are parked as they are; a rejected one's row is
    // its row in HBM, which nothing writes in this launch.)
#ifdef DEMC_EXPERIMENTS
    const bool defer_allowed = p.n_split != -100;  // (A/B build: DEMC_LR_DEFER=0 keeps every row move at its particle's end)
#else
    constexpr bool defer_allowed = true;
#endif
    if (defer_allowed && fast_ok && (hrow || (acc && !masked)) && vb + (int)gridDim.x < n_prop && n_blocks <= 32 * WG && !p.write_prop) {
        pend = true; pend_acc = acc != 0; pend_masked = masked; pend_trow = trow; pend_hrow = hrow;
        pend_wbits = wbits;
        if (!acc) {  // rejected: the history row is the current row -- back into the LDS copy, four blocks' loads in flight
            for (int i0 = 0; i0 * WG + tid < n_blocks; i0 += 4) {
                const int last = n_blocks - 1;
                const int m0 = tid + i0 * WG, m1 = m0 + WG, m2 = m1 + WG, m3 = m2 + WG;
                const int c0 = m0 < last ? m0 : last, c1 = m1 < last ? m1 : last, c2 = m2 < last ? m2 : last, c3 = m3 < last ? m3 : last;
                auto second = [&](int c) { return 4 * (size_t)c + ((4 * c + 2 < D) ? 2 : 0); };  // (a row with D = 2 mod 4 ends on half a block)
                const double2 a0 = *reinterpret_cast<const double2*>(pt + 4 * (size_t)c0), b0 = *reinterpret_cast<const double2*>(pt + second(c0));
                const double2 a1 = *reinterpret_cast<const double2*>(pt + 4 * (size_t)c1), b1 = *reinterpret_cast<const double2*>(pt + second(c1));
                const double2 a2 = *reinterpret_cast<const double2*>(pt + 4 * (size_t)c2), b2 = *reinterpret_cast<const double2*>(pt + second(c2));
                const double2 a3 = *reinterpret_cast<const double2*>(pt + 4 * (size_t)c3), b3 = *reinterpret_cast<const double2*>(pt + second(c3));
                auto put = [&](int m, double2 a, double2 b) {
                    if (m < n_blocks) {
                        *reinterpret_cast<double2*>(scr + 4 * (size_t)m) = a;
                        if (4 * m + 2 < D) *reinterpret_cast<double2*>(scr + 4 * (size_t)m + 2) = b;
                    }
                };
                put(m0, a0, b0); put(m1, a1, b1); put(m2, a2, b2); put(m3, a3, b3);
            }
        }
    } else if (!acc && hrow && even) {
        // Rejected, and the history wants the row: the current row goes from HBM to HBM, FOUR blocks' loads in flight before the
        // first store.  (The general loop below takes a pair at a time -- load, wait, store: twenty dependent round trips per
        // lane at D = 10 002, ~29 k cycles of a ~70 k-cycle workgroup whenever the particle was rejected: profiles/r03 stamps.)
        for (int i0 = 0; i0 * WG + tid < n_blocks; i0 += 4) {
            const int last = n_blocks - 1;
            const int m0 = tid + i0 * WG, m1 = m0 + WG, m2 = m1 + WG, m3 = m2 + WG;
            const int c0 = m0 < last ? m0 : last, c1 = m1 < last ? m1 : last, c2 = m2 < last ? m2 : last, c3 = m3 < last ? m3 : last;
            auto second = [&](int c) { return 4 * (size_t)c + ((4 * c + 2 < D) ? 2 : 0); };
            const double2 a0 = *reinterpret_cast<const double2*>(pt + 4 * (size_t)c0), b0 = *reinterpret_cast<const double2*>(pt + second(c0));
            const double2 a1 = *reinterpret_cast<const double2*>(pt + 4 * (size_t)c1), b1 = *reinterpret_cast<const double2*>(pt + second(c1));
            const double2 a2 = *reinterpret_cast<const double2*>(pt + 4 * (size_t)c2), b2 = *reinterpret_cast<const double2*>(pt + second(c2));
            const double2 a3 = *reinterpret_cast<const double2*>(pt + 4 * (size_t)c3), b3 = *reinterpret_cast<const double2*>(pt + second(c3));
            auto put = [&](int m, double2 a, double2 b) {
                if (m < n_blocks) {
                    *reinterpret_cast<double2*>(hrow + 4 * (size_t)m) = a;  // utilities.jl:170-180
                    if (4 * m + 2 < D) *reinterpret_cast<double2*>(hrow + 4 * (size_t)m + 2) = b;
                }
            };
            put(m0, a0, b0); put(m1, a1, b1); put(m2, a2, b2); put(m3, a3, b3);
        }
    } else if (acc || hrow) {
        for (int it = 0; 2 * pair_at(it) < D; ++it) {  // the lane's pairs in the order the pass visited them (wbits)
            const int k = pair_at(it);
            const int j0 = 2 * k;
            const bool has1 = j0 + 1 < D;
            const bool noted = it < 64 && !((lbits >> (it >> 1)) & 1u);  // (wbits covers the rounds this lane went through itself)
            const bool in_block = noted ? ((wbits >> it) & 1ull) != 0 : (!p.mask || p.mask[j0] || (has1 && p.mask[j0 + 1]));
            // the pair's accepted value comes from the LDS copy; outside the block of a crossover sweep it is the current row's,
            // and where neither the state nor the history row wants the pair it is not touched at all
            const bool parked = acc && (!masked || in_block);
            if (!parked && !hrow) continue;  // nothing to write for this pair
            double v0, v1;
            if (parked) {
                if (even) {
                    const double2 v = *reinterpret_cast<const double2*>(scr + j0);
                    v0 = v.x; v1 = v.y;
                } else {
                    v0 = scr[j0];
                    v1 = has1 ? scr[j0 + 1] : 0.0;
                }
            } else {
                v0 = pt[j0];
                v1 = has1 ? pt[j0 + 1] : 0.0;
            }
            const bool wr = acc && (!masked || in_block);
            if (even) {
                if (wr) *reinterpret_cast<double2*>(trow + j0) = make_double2(v0, v1);  // utilities.jl:204
                if (hrow) *reinterpret_cast<double2*>(hrow + j0) = make_double2(v0, v1);  // utilities.jl:170-180
            } else {
                if (wr) { trow[j0] = v0; if (has1) trow[j0 + 1] = v1; }
                if (hrow) { hrow[j0] = v0; if (has1) hrow[j0 + 1] = v1; }
            }
        }
    }
    DEMC_STAMP_AT(9, 64, DEMC_STAMP_NOW());   // accept + row moves done
    DEMC_STAMP(10);
#ifdef DEMC_STAMPS
    DEMC_STAMP_AT(18, 0, __builtin_amdgcn_s_memrealtime() - t_real0__);
#endif
#ifdef DEMC_STAMPS_TIMELINE  // (instead of the stamps: start and end of EVERY particle on the 100 MHz clock all CUs share)
    if (threadIdx.x == 0 && 2 * ((long long)vb + 1) <= p.P) {
        p.tr_w[2 * vb] = (double)(t_part__ & 0xffffffffull);
        p.tr_w[2 * vb + 1] = (double)(__builtin_amdgcn_s_memrealtime() & 0xffffffffull) + (kind == 2 ? 0.5 : 0.0);  // (.5: a mutation)
    }
#endif
    }  // the workgroup's next particle
    }
}

#ifdef DEMC_LONGROW_EXTERN  // the instances live in demc_longrow.cpp (its own translation unit, its own compiler flags)
extern template __global__ void k_longrow<256>(KParams);
extern template __global__ void k_longrow<512>(KParams);
#ifdef DEMC_EXPERIMENTS
extern template __global__ void k_longrow<384, 2>(KParams);
extern template __global__ void k_longrow<512, 2>(KParams);
#endif
#endif

}  // namespace demc
