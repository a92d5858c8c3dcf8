// demc_resmvn.hpp -- the resident kernel of the DEFAULT sampler on the full-Sigma MvNormal family, written lean.
//
// k_propose's resident form (demc_kernels.hpp) carries every option of the sampler (snooker, recombination, block masks,
// history partners, optimiser updates, traces, every model family, several passes per phase, any lane geometry); its colour
// phase is a chain of ~12 dependent LDS / barrier steps with ~300 spilled scalars.  The BASELINE workloads cfg2 and cfg3 run
// the reference's DEFAULTS (random_gamma, no snooker, kappa = 1, no blocks, Metropolis: DE(), structs.jl:80-101) on
// MvNormal(mu, Sigma) with D = d <= 32 -- this file is that one case and nothing else:
//   * four lanes per particle, ONE pass per colour phase (moving half <= WG/4 particles), a lane owns whole noise blocks
//     (scalars 4m..4m+3, m = sl, sl+4, ...), so a phase draws one PART block and at most two NOISE blocks per lane;
//   * theta' stays in REGISTERS from proposal to write-back; LDS sees it once, centred, for the transposition into MFMA
//     operand order (y = Sigma^-1 (theta' - xbar) on v_mfma_f64_16x16x4_f64, one 16-particle tile per wave);
//   * the group coin of every iteration of the launch is drawn once, at kernel start; bounds / prior table and A^-1
//     fragments live in registers for the whole launch;
//   * select_base: cumulative weights on the DPP network (wave_cdf) by wave 0 while the other waves draw, then a two-level
//     count by the particle's four lanes (chunk ends, then inside the chunk: two LDS round trips);
//   * SUFFSTAT: the whole update here.  STREAMING (STREAM = true): additionally the observation stream of the
//     streaming-resident form (C workgroups per group, granule hand-over) -- same protocol as k_propose<...,STREAM>, but
//     every lane polls the granules of its own particle (no staging, no barrier inside the wait);
//   * DT = 8 (cfg2): two scalars per lane (all four lanes of the quad work on the row), Sigma^-1 (theta' - xbar) on the vector
//     pipe inside the quad, and the tile loop of the observation stage as one asm statement (cross_loop_lds_2x2:
//     accumulators pinned to AGPRs).
// Same addressed draws and the same per-scalar arithmetic as k_propose: proposals and decisions are the ones the general
// kernel produces (tests/test_gpu_parity.py::test_lean_resident_kernel_*); prior sums run in a different lane order
// (log-densities equal to rounding).
#pragma once
#include "demc_kernels.hpp"

namespace demc {

// DT > 0: an instance for D = d = DT (a multiple of 4) with one table segment (32: BASELINE cfg3, 8: cfg2): every "is this
// scalar inside the row" test, the ragged-block paths and the segment lookup fold away at compile time (the general instance,
// DT = 0, spends a third of its proposal stage on that control flow).
// HIST: DE-MC_Z (`sample = resample`, crossover.jl:113-124, synchronous schedule): the two partner rows are cells of the HISTORY
// (rows 1:iter-1 x all particles of the handle, which other groups wrote in earlier launches: ONE iteration per launch); during
// burn-in random_gamma also reads a base particle of the group's CURRENT population (crossover.jl:156-164) -- this workgroup's own
// group, all of whose writes it holds back until the second half has read what it needs.  So there is no tile, no weight copy
// and no barrier between the two halves of the group, which run as the "phases" of the one iteration; own rows, partner rows,
// base rows and weights come straight from HBM, every load of a lane issued before the first is needed.
// (The general kernel's lean instance served this before: 2.2x the cycles per particle, profiles/r04/stamps_demcz.txt.)
// HIST_ = 1: the iterations past burn-in (no base particle); 2: inside burn-in.  Two instances because the base row's loads,
// compiled in, cost the iterations that never use them 3.7 of 20.6 us per launch (registers at the cap, one more stream of loads).
// HIST_ = 3: DE-MC_Z as the reference's own runs configure it -- theta_snooker > 0 (test/multivariate_normal_tests.jl:50-59,
// Examples/Hierarchical_Example.jl:103-114): a particle whose snooker coin fires (crossover.jl:31) draws THREE history cells
// (crossover.jl:241), the third one in the slot the base row has in instance 2 (a snooker update reads no base particle), projects
// two of them onto Pt - Pz (utilities.jl:239-246: three dot products, summed over the quad) and carries adjust_loglike's norms
// (crossover.jl:268-273) into the decision; the other particles of the wave take the crossover as in instance 2, in or past burn-in.
// OCC = 2 (STREAM, 256 threads; A/B builds only): the instance compiled for TWO workgroups per CU -- 256 registers per lane, the
// observation stage's accumulators in AGPRs inside that budget -- so that a small population (BASELINE cfg2: 32 groups on 256 CUs)
// is cut into twice as many observation chunks and a CU holds workgroups of two DIFFERENT groups, one group's dependent chain
// (draws, proposal, hand-over round trip, decision) under the other's matrix stage.  Measured slower (plan_lean).
// ISO = MvNormal(mu, sigma^2 I) with sigma a PARAMETER (theta = (mu[d], sigma), D = d + 1 <= 32; test/multivariate_normal_tests.jl:
// 16-33, which the reference runs as DE-MC_Z with snooker, :50-59): the same body with the quadratic form on the vector pipe inside
// the quad -- aux = |mu - xbar|^2 and S = (mu - xbar) . sum_i x~_i are eight products per lane and one quad sum each, sigma comes
// from the lane that holds scalar d -- no A^-1, no LDS transposition, no matrix stage; the prior table has two segments (Normal on
// mu, Cauchy+ on sigma).  HIST instances only: with current-population partners the general kernel's lean instances serve it.
// DIR (round 6; STREAM instances with the row length compiled in): the streaming-resident form in DIRECT mode -- the likelihood in
// the residual form the reference writes (test/multivariate_normal_tests.jl:31-33), sum_i |z_i - m|^2 term by term with z_i = L^-1 x~_i
// whitened once and m = L^-1 (theta' - xbar) per proposal (SURVEY 8d's 3 N D per update: the work that does not collapse), on the
// FP64 vector pipe: the workgroup's chunk of whitened rows rides in LDS where the MFMA form keeps its fragment-ordered tiles, the
// proposal stage leaves m where it left y (the host hands (L^-1)' over in place of Sigma^-1), and a thread walks every n-th row of
// the chunk for ONE proposal with m in registers (rows are LDS broadcasts: the lanes of a slice read the same address).  Hand-over,
// decisions and stores are the streaming form's.  For populations too small to fill the chip with the K1 -> k_direct_mvn -> K3
// chain (BASELINE cfg2: six dependent launches per iteration for 4.9e8 flop).
template <int WG, bool STREAM, int DT = 0, int HIST_ = 0, int OCC = 1, bool ISO = false, bool DIR = false>
__global__ __launch_bounds__(WG, (WG == 256 && (!STREAM || OCC == 2)) ? 2 : 1) void k_res_mvn(KParams p) {
    constexpr bool HIST = HIST_ != 0;
    static_assert(!DIR || (STREAM && DT > 0 && (DT & 1) == 0), "the DIRECT form: streaming-resident instances with an even row length compiled in");
    static_assert(!(HIST && STREAM), "history partners: the SUFFSTAT form only");
    static_assert(!ISO || (HIST && (DT == 0 || DT == 31)), "the isotropic form: DE-MC_Z instances, general row length or D = 31 (the reference's test)");
    // (one iteration per launch: launch_lean_hist never asks for more, and checks it; the kernel's own guard is the trip count of
    // the phase loop below.  An early `return` on p.n_iters != 1 HERE costs ~24 VGPRs -- round 4: 240 -> 254 and 64 B of scratch
    // in the hot instance, +25 % per launch.)
    extern __shared__ double lds[];
    __shared__ unsigned char s_mut[1024];  // beta coin of every iteration of this launch (n_iters <= 1024)
    __shared__ DimSeg s_seg[kMaxDimSeg];   // bounds / prior table, run-length encoded (usually ONE segment for this family)
    for (int i = threadIdx.x; i < p.n_seg * (int)(sizeof(DimSeg) / sizeof(double)); i += WG)
        reinterpret_cast<double*>(s_seg)[i] = reinterpret_cast<const double*>(p.dimseg)[i];
    DEMC_STAMP_INIT();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = DT ? DT : p.D, Np = p.Np, d = DT ? (ISO ? DT - 1 : DT) : p.d;  // D == d (ISO: D == d + 1)
    int g, c_idx = 0;
    if (STREAM) {
        if ((p.n_groups & 7) == 0) {
            const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
            c_idx = j % p.st_C;
            g = (j / p.st_C) * 8 + xcd;
        } else {
            g = blockIdx.x / p.st_C;
            c_idx = blockIdx.x % p.st_C;
        }
    } else
        g = blockIdx.x;
    const int gi = g;
    if (p.glist) g = p.glist[g];
    const int g_glob = p.group_offset + g;
    double* grows = p.theta + (size_t)g * Np * D;
    double* gw = p.weight + (size_t)g * Np;
    // STREAM: ONE workgroup of the group writes HBM (state, weights, history) -- the one with the LAST chunk of observation
    // tiles, which is the short one (n_tiles rarely divides by C): its stores cost ~0.25 us per phase, and whoever stores its
    // hand-over granules last is whom the other seven wait for (tools/k1_stamps.py, slots 22 / 23)
    const bool wr_hbm = !STREAM || c_idx == p.st_C - 1;
    const bool even = DT > 0 ? (DT & 1) == 0 : (D & 1) == 0;
    const int half = Np / 2, nact_max = Np - half;
    // LDS: tile [Np][D] | weights [Np] | cdf [nact_max] + chunk offsets [16] | centred theta' rows [WG/4][D+2] |
    //      STREAM: y rows [WG/4][dpad] | per-wave partials [WG/64][nact_max] | X chunk
    double* tile = lds;
    double* w_s = tile + (HIST ? 0 : (size_t)Np * D);
    double* cdf = w_s + (HIST ? 0 : Np);
    double* coff = cdf + (HIST ? Np : nact_max);  // (HIST: select_base runs over the whole group, crossover.jl:282-289)
    const int scr_stride = D + 2;
    double* scr = coff + 16;
    double* ybuf = scr + (size_t)(WG / 4) * scr_stride;
    double* part_l = ybuf + (STREAM ? (size_t)(WG / 4) * p.dpad : 0);
    double* xs = reinterpret_cast<double*>((reinterpret_cast<size_t>(part_l + (STREAM ? (size_t)(DIR ? WG / 16 : WG / 64) * nact_max : 0)) + 15) & ~(size_t)15);
    const int xt_lo = STREAM ? c_idx * p.st_chunk_tiles : 0;
    const int xt_hi = STREAM ? (xt_lo + p.st_chunk_tiles < p.n_tiles ? xt_lo + p.st_chunk_tiles : p.n_tiles) : 0;

    // ---- once per launch: the group into LDS, the coins of every iteration, the loop-invariant tables into registers ----
    if constexpr (!HIST) {
    if (even) {
        const int n16 = (Np * D) >> 1;
        for (int c0 = wave * 64; c0 < n16; c0 += WG)
            if (c0 + lane < n16) lds_dma16(grows + 2 * (size_t)(c0 + lane), tile + 2 * (size_t)c0);
    } else
        for (int i = tid; i < Np * D; i += WG) tile[i] = grows[i];
    for (int i = tid; i < Np; i += WG) w_s[i] = gw[i];
    }
    for (int i = tid; i < p.n_iters; i += WG) {
        const U4 r = draw_block(p.seed, S_GROUP, 0, (uint64_t)(p.iter + i), (uint32_t)g_glob, 0);
        s_mut[i] = u53(r.x, r.y) <= p.beta ? 1 : 0;  // mutate_or_crossover! main.jl:199-207
    }
    if (STREAM) {
        for (int i = tid; i < (WG / 4) * p.dpad; i += WG) ybuf[i] = 0.0;
        if (DIR && xt_lo < xt_hi) {  // the chunk's whitened rows z_i [obs][DT], row-major (demc_set_model, DIRECT)
            const long long o_hi = (long long)xt_hi * 16 < p.N ? (long long)xt_hi * 16 : p.N;
            const double* src = p.data + (size_t)xt_lo * 16 * DT;
            const int n16 = (int)(((o_hi - (long long)xt_lo * 16) * DT) >> 1);
            for (int c0 = wave * 64; c0 < n16; c0 += WG)
                if (c0 + lane < n16) lds_dma16(src + 2 * (size_t)(c0 + lane), xs + 2 * (size_t)c0);
        } else if (p.st_x_lds && xt_lo < xt_hi) {
            const double* src = p.Xf + (size_t)xt_lo * (p.dpad >> 2) * 64;
            const int n16 = ((xt_hi - xt_lo) * (p.dpad >> 2) * 64) >> 1;
            for (int c0 = wave * 64; c0 < n16; c0 += WG)
                if (c0 + lane < n16) lds_dma16(src + 2 * (size_t)(c0 + lane), xs + 2 * (size_t)c0);
        }
        if (!DIR && p.st_x_lds)
            for (int i = tid; i < (p.dpad >> 2) * 64; i += WG) xs[(size_t)p.st_chunk_tiles * (p.dpad >> 2) * 64 + i] = 0.0;
    }
    // lane geometry: particle slot q = tid / 4 of the pass, lane sl of the particle owns noise blocks m = sl and sl + 4,
    // i.e. scalars 4 sl .. 4 sl + 3 and 16 + 4 sl .. 16 + 4 sl + 3 (D <= 32).  DT == 8: two scalars per lane instead (2 sl,
    // 2 sl + 1: half of noise block sl / 2) -- all four lanes of the quad work on the row, not two of them on four scalars each
    const int q = tid >> 2, sl = tid & 3;
    constexpr int SPL = DT == 8 ? 2 : 4;                         // scalars per lane and block
    const int jA = SPL * sl, jB = DT == 8 ? 64 : 16 + 4 * sl;    // first scalar of the lane's two blocks (DT == 8: one)
    const uint32_t nbA = DT == 8 ? (uint32_t)(sl >> 1) : (uint32_t)sl;  // the NOISE block behind the lane's first scalars
    // xbar of the lane's 8 scalars, and the table segment of each (4 bits apiece; the entries themselves stay in LDS)
    double xb[8];
    unsigned segs = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int j = (e < 4 ? jA : jB - 4) + e;
        const int jj = j < D ? j : 0;
        xb[e] = ISO ? 0.0 : p.xbar[jj];  // (ISO: from the LDS copy, iso_l)
        unsigned sg = 0;
        for (int i = 1; i < p.n_seg; ++i) sg += (jj >= p.dimseg[i].start) ? 1u : 0u;
        segs |= sg << (4 * e);
    }
    // A^-1 fragments for the MFMA preparation (B operand: k = 4 ks + (lane >> 4), column 16 nt + (lane & 15)).
    // HIST: parked in LDS ([2][8][64 lanes], the same for every wave) and read back in the matrix stage of each half -- that
    // kernel sits at the register cap (its loads come from HBM and stay in flight longer), and 32 registers held across the whole
    // launch for a stage of 2 k cycles are what tipped its allocation into scratch whenever anything else changed.
    constexpr bool BF_LDS = HIST;
    double* const bf_l = scr + (size_t)(WG / 4) * scr_stride;  // (HIST only: behind the centred rows; never with STREAM)
    // HIST_ == 3 (snooker: one more row in flight): the first half's rows, held back until the second half's loads are in, wait in
    // LDS, not in 16 registers per lane held across the second half's whole proposal stage (written and read back by the same
    // lane: no barrier; [pair][thread] double2 = conflict-free 16-byte accesses) -- that instance sat AT 256 registers with a spill;
    // 245 and none this way.  Measured on all three instances in round 5 (profiles/r05/NOTES.md): parking costs 3-4 % per launch
    // (and parking the current row's scalars and xbar as well, which are read on the phase's dependent chain, 12 %), so instances
    // 1 and 2 (221 / 241 registers, no scratch) keep theirs in registers.
    // Round 6: the odd-row ISO instances (D = 31: test/multivariate_normal_tests.jl) park in instance 2 as well -- their 16-byte loads at
    // 8-byte alignment and the sigma scalar's own table entry cost the registers the D = 32 instances have to spare (instance 2 spilled
    // 16 registers into 80 B of scratch, instance 3 26 into 112).
    constexpr bool PARK = HIST_ == 3 || (ISO && HIST_ == 2);
    double* const pend_l = reinterpret_cast<double*>((reinterpret_cast<size_t>(bf_l + 16 * 64) + 15) & ~(size_t)15);  // (Np may be odd)
    auto park8 = [&](double* base, const double (&x)[8]) {
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2)
            *reinterpret_cast<double2*>(base + ((size_t)e2 * WG + threadIdx.x) * 2) = make_double2(x[2 * e2], x[2 * e2 + 1]);
    };
    auto unpark8 = [&](const double* base, double (&x)[8]) {
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
            const double2 v = *reinterpret_cast<const double2*>(base + ((size_t)e2 * WG + threadIdx.x) * 2);
            x[2 * e2] = v.x; x[2 * e2 + 1] = v.y;
        }
    };
    double bfrag[2][8];
    if constexpr (!ISO) {
        const int kq = lane >> 4, col = lane & 15;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int k = 4 * ks + kq, c = 16 * nt + col;
                bfrag[nt][ks] = (k < d && c < d) ? p.Ainv[k * d + c] : 0.0;
                if (BF_LDS && wave == 0) bf_l[(nt * 8 + ks) * 64 + lane] = bfrag[nt][ks];
            }
    }
    // DT == 8: the product stays on the vector pipe (see the phase loop); lane sl of a quad owns columns 2 sl, 2 sl + 1 of A^-1
    constexpr bool DIRECT8 = DT == 8;
    double ai8[2][8], sx8[2];
    if constexpr (DIRECT8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            ai8[0][k] = p.Ainv[k * 8 + 2 * sl];
            ai8[1][k] = p.Ainv[k * 8 + 2 * sl + 1];
        }
        sx8[0] = p.sx ? p.sx[2 * sl] : 0.0;
        sx8[1] = p.sx ? p.sx[2 * sl + 1] : 0.0;
    }
    // ISO: xbar and sum_i x~_i are read from an LDS copy [2][32] where the quadratic form is formed (behind the held-back rows'
    // slot): 32 registers per lane held across the proposal stage were what made these instances spill 44 - 60 registers
    double* const iso_l = pend_l + (size_t)WG * 8;
    if constexpr (ISO) {
        if (tid < 64) {
            const int j = tid & 31;
            iso_l[tid] = j < d ? (tid < 32 ? p.xbar[j] : (p.sx ? p.sx[j] : 0.0)) : 0.0;  // (visible after the barrier below)
        }
    }
    const int mc0 = lane & 15, mc1 = mc0 + 16;  // the two columns this lane sees of every MFMA result
    const double sx0 = (p.sx && mc0 < d) ? p.sx[mc0] : 0.0, sx1 = (p.sx && mc1 < d) ? p.sx[mc1] : 0.0;
    if (even || (STREAM && p.st_x_lds)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // the ids of the two particles this quad moves (first half, second half): they only change between launches (migration),
    // and a load from HBM in the accept stage of every phase is a latency the whole group ends up waiting for
    const int id_lo = (int)p.id[(size_t)g * Np + (q < half ? q : 0)], id_hi = (int)p.id[(size_t)g * Np + half + (q < Np - half ? q : 0)];
    const double eps = p.eps, eps2 = p.eps - (-p.eps);
    // HIST: ONE iteration per launch (the held-back first-half stores use that iteration's store_row; the cdf is formed once).
    // The guard is the trip count: a launch that asks for anything else runs no phase at all.
    const long long n_steps = HIST ? (p.n_iters == 1 ? 2 : 0) : (long long)p.n_iters * 2;
    // HIST -- resample (crossover.jl:113-124) exactly as k_propose draws it: PART block 4 of the particle holds the two cell draws;
    // distinct cells of rows 1:(iter-1) x the handle's particles; cell x = (row x mod (iter-1), slot x div (iter-1))
    auto hist_rows = [&](long long iter_, uint32_t es, bool snk_, const double*& a_o, const double*& b_o, const double*& c_o) {
        const U4 cells = draw_block(p.seed, S_PART, 0, (uint64_t)iter_, es, (uint32_t)(4 + (threadIdx.x & 1)));
        const U4 h4 = bcast_u4<0>(cells, 4, 0);
        const uint64_t hd0 = ((uint64_t)h4.y << 32) | h4.x, hd1 = ((uint64_t)h4.w << 32) | h4.z;
        const uint64_t ub = (uint64_t)(iter_ - 1), M = ub * (uint64_t)p.P;
        const uint64_t a = mulhi64(hd0, M);
        uint64_t b = mulhi64(hd1, M - 1);
        if (b >= a) ++b;
        uint64_t c = 0;
        if (HIST_ == 3 && snk_) {  // the third cell, distinct from the two (PART block 5)
            const U4 h5 = bcast_u4<1>(cells, 4, 0);
            c = mulhi64(((uint64_t)h5.y << 32) | h5.x, M - 2);
            const uint64_t lo = a < b ? a : b, hi = a < b ? b : a;
            if (c >= lo) ++c;
            if (c >= hi) ++c;
        }
        auto cell = [&](uint64_t x) -> const double* {
            uint64_t row, sl_;
            if ((M >> 32) == 0) {  // (the 64-bit division is a ~100-instruction routine; wave-uniform branch)
                const uint32_t x32 = (uint32_t)x, u32 = (uint32_t)ub, qd = x32 / u32;
                sl_ = qd; row = x32 - qd * u32;
            } else {
                sl_ = x / ub; row = x - sl_ * ub;
            }
            return p.hist + (row * (uint64_t)p.P + sl_) * (uint64_t)p.hist_ld;
        };
        a_o = cell(a);
        b_o = cell(b);
        if (HIST_ == 3 && snk_) c_o = cell(c);
    };
    // HIST: what a particle writes -- state and weight when accepted, the history row (the accepted proposal or the current row:
    // x8, the lane's eight scalars of it) and its bookkeeping (utilities.jl:161-180, 201-210).  The FIRST half's writes are held back
    // until the second half's rows have arrived: a wave's loads queue behind its own earlier stores, and with the stores in front
    // the second half waited 12 k cycles for rows that take 1.4 k (profiles/r04/stamps_demcz.txt).
    auto hist_store = [&](int ph_, int acc_, double wp_, double w_, long long store_row_, const double (&x8)[8]) {
        if (q >= (ph_ ? Np - half : half)) return;
        const int pl_ = (ph_ ? half : 0) + q;
        const size_t slot_ = (size_t)g * Np + pl_;
        if (sl == 0) {
            if (acc_) p.weight[slot_] = wp_;
            if (store_row_ >= 0) {
                const size_t hrow_ = (size_t)store_row_ * p.P + slot_;
                p.acc_hist[hrow_] = (unsigned char)acc_;
                p.lp_hist[hrow_] = acc_ ? wp_ : w_;
                p.id_hist[hrow_] = ph_ ? id_hi : id_lo;
            }
        }
        double* trow = p.theta + slot_ * D;
        double* hrow = store_row_ >= 0 ? p.hist + ((size_t)store_row_ * p.P + slot_) * p.hist_ld : nullptr;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int j = blk ? jB : jA;
            if (j >= D || (!acc_ && !hrow)) continue;
#pragma unroll
            for (int e = 0; e < SPL; e += 2) {
                if (j + e >= D) continue;
                const double2 x = make_double2(x8[4 * blk + e], x8[4 * blk + e + 1]);
                if (even) {
                    if (acc_) *reinterpret_cast<double2*>(trow + j + e) = x;
                    if (hrow) *reinterpret_cast<double2*>(hrow + j + e) = x;
                } else if (j + e + 1 < D) {  // (odd row length: 16-byte stores at 8-byte alignment, as the loads)
                    typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
                    d2u xu;
                    xu.x = x.x; xu.y = x.y;
                    if (acc_) *reinterpret_cast<d2u*>(trow + j + e) = xu;
                    if (hrow) *reinterpret_cast<d2u*>(hrow + j + e) = xu;
                } else {
                    if (acc_) trow[j + e] = x.x;
                    if (hrow) hrow[j + e] = x.x;
                }
            }
        }
    };
    double pend_x[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, pend_wp = 0.0, pend_w = 0.0;  // (PARK: the rows wait in pend_l)
    int pend_acc = 0;
    bool pend = false;
    // The addressed draws of a colour phase (the particle's PART block, its NOISE blocks) depend on (seed, iteration, slot)
    // only, not on the state: STREAM draws those of the NEXT phase between storing its hand-over granules and polling for
    // the others' -- in the shadow of the L2 round trip, off the next phase's chain (and the first poll comes later, when
    // the peers' granules have had time to arrive).
    U4 pre_mine = {0, 0, 0, 0}, pre_nzA = pre_mine, pre_nzB = pre_mine;
    auto draw_phase = [&](long long st, U4& mine_o, U4& nzA_o, U4& nzB_o) {
        const int ph_ = (int)(st & 1);
        const long long iter_ = p.iter + (st >> 1);
        const int a_lo_ = ph_ ? half : 0, n_act_ = ph_ ? Np - half : half;
        const int q_ = threadIdx.x >> 2, sl_ = threadIdx.x & 3;
        const int pl_ = a_lo_ + (q_ < n_act_ ? q_ : 0);
        const uint32_t es = (uint32_t)g_glob * (uint32_t)Np + (uint32_t)pl_;
        mine_o = draw_block(p.seed, S_PART, 0, (uint64_t)iter_, es, (uint32_t)sl_);
        nzA_o = draw_block(p.seed, S_NOISE, 0, (uint64_t)iter_, es, DT == 8 ? (uint32_t)(sl_ >> 1) : (uint32_t)sl_);
        nzB_o = nzA_o;
        if (DT != 8 && 16 + 4 * sl_ < D) nzB_o = draw_block(p.seed, S_NOISE, 0, (uint64_t)iter_, es, (uint32_t)(sl_ + 4));
    };
    // (Round 4 also let wave 0 -- which forms select_base's cumulative weights while the other waves draw, and then draws for
    // its own particles -- make its draws of the next phase at the end of this one: no measurable change, dropped again.  Nor did
    // the three-round cumulative weights or the two-level base pick move the phase: it is a chain of dependent latencies at two
    // waves per SIMD with the vector pipe 0.39 busy, not a count of instructions.)
    if (STREAM) draw_phase(0, pre_mine, pre_nzA, pre_nzB);
#ifdef DEMC_EXPERIMENTS
    if constexpr (STREAM && OCC == 2) {
        // A/B only: the second workgroup of every CU (the upper half of the grid: other groups) starts p.st_rows shader cycles late,
        // so that the two workgroups of a CU run OUT of step -- one's matrix stage under the other's proposal / hand-over stages
        if (p.st_rows > 0 && blockIdx.x >= gridDim.x / 2) {
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)p.st_rows) __builtin_amdgcn_s_sleep(8);
        }
    }
#endif
    for (long long step = 0; step < n_steps; ++step) {
        DEMC_STAMP_RESET();
        const int ph = (int)(step & 1);
        const int it_rel = (int)(step >> 1);
        long long iter_o = p.iter + it_rel;
        // (HIST: with the trip count as the guard the compiler knows it_rel == 0 and hoists everything that depends on the
        // iteration alone -- Philox key schedules, row numbers -- out of the loop, live across both halves: +14 VGPRs, scratch in
        // instance 3.  Opaque instead.)
        if constexpr (HIST) asm volatile("" : "+s"(iter_o));
        const long long iter = iter_o;
        const int a_lo = ph ? half : 0, n_act = ph ? Np - half : half;
        const int pool_lo = HIST ? 0 : (ph ? 0 : half), pool_n = HIST ? Np : (ph ? half : Np - half);
        const long long store_row = (p.hist && iter - 1 < p.n_rows) ? iter - 1 : -1;
        const bool is_mut = s_mut[it_rel] != 0;
        const bool use_base = HIST_ != 1 && !is_mut && iter <= p.burnin;  // crossover.jl:164 (HIST_ == 1 is launched past burn-in only)
        constexpr bool HSNK = HIST_ == 3;
        const bool valid = q < n_act;
        const int pl = a_lo + (valid ? q : 0);
        const size_t slot = (size_t)g * Np + pl;
        const uint32_t eslot = (uint32_t)g_glob * (uint32_t)Np + (uint32_t)pl;

        // ---- select_base's cumulative weights (wave 0) while the other waves draw ----
        // (HIST: once for both halves -- the weights of the iteration's start, which nothing has written yet: the first half's
        // stores are held back behind the second half's loads)
        if (use_base && wave == 0 && (!HIST || ph == 0)) {
            const double* pw = HIST ? gw : w_s + pool_lo;
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
        DEMC_STAMP(0);  // wave 0: cumulative weights in LDS
        // ---- per-particle scalars: lane sl draws PART block sl, the quad shares them (crossover.jl:156-166, utilities.jl:57) ----
        const U4 mine = STREAM ? pre_mine : draw_block(p.seed, S_PART, 0, (uint64_t)iter, eslot, (uint32_t)sl);
        const U4 r0 = bcast_u4<0>(mine, 4, 0), ri = bcast_u4<1>(mine, 4, 0), rg = bcast_u4<2>(mine, 4, 0), ra = bcast_u4<3>(mine, 4, 0);
        const double u_base = u53(r0.z, r0.w), u_acc = u53(ra.x, ra.y);
        uint32_t ia = 0, ib = 0;
        const double *Pa_h = nullptr, *Pb_h = nullptr, *Pc_h = nullptr;
        double w_h = 0.0;
        const bool snk = HSNK && !is_mut && u53(r0.x, r0.y) <= p.theta_snooker;  // crossover.jl:31 (this particle: a snooker update)
        const bool base_p = use_base && !snk;                                     // (a snooker update reads no base particle)
        if constexpr (HIST) {
            hist_rows(iter, eslot, snk, Pa_h, Pb_h, Pc_h);
            w_h = gw[pl];  // (asked for here, needed at the decision)
        } else
            pick_pair(ri.x, ri.y, (uint32_t)pool_n, ia, ib);  // two_colour: the pool is the resting half, self is not in it
        const double g1 = snk ? 1.2 + (2.2 - 1.2) * u53(rg.x, rg.y) : 0.5 + (1.0 - 0.5) * u53(rg.x, rg.y);  // crossover.jl:249 / 162
        const double g2 = base_p ? 0.5 + (1.0 - 0.5) * u53(rg.z, rg.w) : 0.0;
        // noise: block sl (scalars jA..jA+3) and block sl + 4 (scalars jB..jB+3)
        const U4 nzA = STREAM ? pre_nzA : draw_block(p.seed, S_NOISE, 0, (uint64_t)iter, eslot, nbA);
        U4 nzB = STREAM ? pre_nzB : nzA;
        if (!STREAM && jB < D) nzB = draw_block(p.seed, S_NOISE, 0, (uint64_t)iter, eslot, (uint32_t)(sl + 4));
        const bool hi2 = DT == 8 && (sl & 1);  // (DT == 8: the lane's two scalars are the second half of the block)
        const uint32_t nw[8] = {hi2 ? nzA.z : nzA.x, hi2 ? nzA.w : nzA.y, nzA.z, nzA.w, nzB.x, nzB.y, nzB.z, nzB.w};

        DEMC_STAMP(1);  // particle and noise blocks drawn
        int ibase = 0;
        if (use_base) {
            if (!HIST || ph == 0) lds_barrier();  // cdf visible
            const double total = cdf[pool_n - 1];
            if (!(total > 0.0) || !(total < INFINITY)) {
                ibase = (int)(u_base * pool_n);
                ibase = ibase < pool_n ? ibase : pool_n - 1;
            } else {
                // first i with cdf[i] >= t, else last = the number of entries below t (cdf is monotone), counted on two
                // levels by the particle's four lanes: the chunk (16 entries) from the chunk ENDS -- at most four reads
                // per lane, all in flight together -- then the position inside that chunk, four entries per lane: two
                // LDS round trips where the binary search made log2(pool) dependent ones (round 3: ~1.4 k cycles of a phase)
                const double t = u_base * total;
                const int n_chunk = (pool_n + 15) >> 4;
                int below = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = sl + 4 * k;  // pools of up to 256: sixteen chunks
                    const int last = 16 * c + 15 < pool_n ? 16 * c + 15 : pool_n - 1;
                    const double ce = cdf[c < n_chunk ? last : 0];
                    below += (c < n_chunk && ce < t) ? 1 : 0;
                }
                below = subgroup_sum(below, 4);
                const int c0 = 16 * (below < n_chunk ? below : n_chunk - 1);
                int cnt = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = c0 + 4 * sl + k;
                    const double ci = cdf[i < pool_n ? i : pool_n - 1];
                    cnt += (i < pool_n && ci < t) ? 1 : 0;
                }
                cnt = subgroup_sum(cnt, 4);
                ibase = c0 + cnt;
                ibase = ibase < pool_n ? ibase : pool_n - 1;
            }
        }
        DEMC_STAMP(4);  // base picked
        // ---- proposal of the lane's 8 scalars, bounds, prior ----
        const double* pt = HIST ? grows + (size_t)pl * D : tile + (size_t)pl * D;
        const double* Pa = HIST ? Pa_h : tile + (size_t)(pool_lo + (int)ia) * D;
        const double* Pb = HIST ? Pb_h : tile + (size_t)(pool_lo + (int)ib) * D;
        // (HIST_ == 2: the base row's loads are issued whether or not the iteration is still inside burn-in -- the same loads in the
        // same order every time, so that the compiler can wait for each where it is needed)
        const double* Pc = HIST ? (snk ? Pc_h : base_p ? grows + (size_t)ibase * D : pt) : tile + (size_t)(pool_lo + ibase) * D;
        double t8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // HIST: the lane's scalars of the current row (a rejected particle's history row), kept from the proposal
        double v8[8];
        int oob = 0;
        double prior = 0.0;
        // ISO with D compiled in (launch_lean_hist takes that instance only for the reference's own table: ONE entry for the means, one
        // for sigma): the means share the wave-uniform entry 0 like any one-segment row, and the single sigma scalar -- scalar d, a fixed
        // slot (second block, third scalar) of the quad's last lane -- reads entry 1.  The general per-scalar lookup (a table pointer, six
        // LDS reads and a kind switch per scalar) was what these instances spilled on.
        constexpr bool ISO2 = ISO && DT > 0;
        const bool one_seg = (DT > 0 && !ISO) || ISO2 || p.n_seg == 1;  // the usual case for this family: every scalar shares one table entry
        // ... which is then wave-uniform: through readfirstlane its fields sit in SGPRs and the switch on the prior kind is a
        // scalar branch (as a per-lane value it compiles to one exec-masked region per prior kind and scalar)
        auto uni = [](double x) {
            return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
        };
        DimTab tb1 = s_seg[0].t;
        tb1.lo = uni(tb1.lo); tb1.hi = uni(tb1.hi); tb1.a = uni(tb1.a); tb1.b = uni(tb1.b); tb1.c = uni(tb1.c);
        tb1.kind = __builtin_amdgcn_readfirstlane(tb1.kind);
        auto load4 = [&](const double* row, int j0, double (&o)[4]) {  // four (DT == 8: two) consecutive scalars of a tile row
            if constexpr (DT == 8) {
                const double2 a = *reinterpret_cast<const double2*>(row + j0);
                o[0] = a.x; o[1] = a.y;
            } else if (even && j0 + 3 < D) {
                const double2 a = *reinterpret_cast<const double2*>(row + j0), b = *reinterpret_cast<const double2*>(row + j0 + 2);
                o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
            } else if (HIST && j0 + 3 < D) {
                // an odd row length (rows 8-byte aligned only): still two 16-byte loads -- global memory takes them at any dword
                // alignment -- instead of four 8-byte ones (half the load instructions and address registers of the ISO instances)
                typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
                const d2u a = *reinterpret_cast<const d2u*>(row + j0), b = *reinterpret_cast<const d2u*>(row + j0 + 2);
                o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
            } else
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = j0 + e < D ? row[j0 + e] : 0.0;
        };
        double zz[8];  // mutation sweeps only: the standard normals of the lane's scalars
        if (is_mut) {
#pragma unroll
            for (int pe = 0; pe < (DT == 8 ? 1 : 4); ++pe) {
                const double2 z = box_muller_outofline(nw[2 * pe], nw[2 * pe + 1]);
                zz[2 * pe] = z.x;
                zz[2 * pe + 1] = z.y;
            }
        }
        // HIST_ == 3, snooker: project(Pm, Pd), project(Pn, Pd) with Pd = Pt - Pz need whole-row dot products first (utilities.jl:
        // 239-246); the rows are read again by the proposal loop below (they sit in the vector cache by then)
        double cm = 0.0, cn = 0.0, s1 = 0.0, s2 = 0.0;
        if constexpr (HSNK) {
            if (__ballot(snk) != 0ull) {
                double vm = 0.0, vn = 0.0, vd = 0.0;
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const int j0 = blk ? jB : jA;
                    if (j0 >= D) continue;
                    double tt[4], aa[4], bb[4], cc[4];
                    load4(pt, j0, tt); load4(Pa, j0, aa); load4(Pb, j0, bb); load4(Pc, j0, cc);
#pragma unroll
                    for (int e4 = 0; e4 < SPL; ++e4)
                        if (j0 + e4 < D) {
                            const double dj = tt[e4] - aa[e4];
                            vm += bb[e4] * dj; vn += cc[e4] * dj; vd += dj * dj;
                        }
                }
                vm = subgroup_sum(vm, 4); vn = subgroup_sum(vn, 4); vd = subgroup_sum(vd, 4);
                cm = vm / vd; cn = vn / vd;
            }
        }
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int j0 = blk ? jB : jA;
#pragma unroll
            for (int e = 0; e < 4; ++e) v8[4 * blk + e] = 0.0;
            if (j0 >= D) continue;
            double tt[4], aa[4], bb[4], cc[4];
            load4(pt, j0, tt);
            if constexpr (HIST) {
#pragma unroll
                for (int e4 = 0; e4 < SPL; ++e4) t8[4 * blk + e4] = tt[e4];
            }
            if (!is_mut) {
                load4(Pa, j0, aa);
                load4(Pb, j0, bb);
                if (HIST_ >= 2 || use_base) load4(Pc, j0, cc);
            }
#pragma unroll
            for (int e4 = 0; e4 < SPL; ++e4) {
                const int e = 4 * blk + e4, j = j0 + e4;
                if (j < D) {
                    const double tj = tt[e4];
                    double v;
                    if (is_mut)  // pt + Normal(0, sigma)  mutation.jl:15-18
                        v = tj + p.sigma * zz[e];
                    else if (HSNK && snk) {  // (Pt + gamma*(Pr1 - Pr2)) + b  crossover.jl:253
                        const double dj = tj - aa[e4];
                        const double t1 = dj * cm - dj * cn;
                        v = (tj + t1 * g1) + (-eps + eps2 * u32unit(nw[e]));
                        const double a0 = v - aa[e4];  // adjust_loglike's norms (crossover.jl:268-273)
                        s1 += a0 * a0; s2 += dj * dj;
                    } else {  // ((Pt + g1*(Pm-Pn)) + g2*(Pb-Pt)) + b  crossover.jl:168
                        const double t1 = aa[e4] - bb[e4];
                        double t6 = tj + t1 * g1;
                        if (HSNK ? base_p : use_base) {
                            const double t4 = cc[e4] - tj;
                            t6 = t6 + t4 * g2;
                        }
                        v = t6 + (-eps + eps2 * u32unit(nw[e]));
                    }
                    v8[e] = v;
                    if (ISO2 && blk == 1 && e4 == (DT - 1) % 4 && j == DT - 1) {  // sigma: its own table entry (Cauchy+ in the reference's test)
                        const DimTab* tb = &s_seg[1].t;
                        oob |= !(v >= tb->lo && v <= tb->hi);
                        if (tb->kind != PR_FLAT) prior += prior_term_outofline(tb, v);
                    } else if (one_seg) {
                        oob |= !(v >= tb1.lo && v <= tb1.hi);  // in_bounds utilities.jl:70-78
                        if (tb1.kind == PR_NORMAL) {
                            const double z = (v - tb1.a) * tb1.b;
                            prior += tb1.c - 0.5 * (z * z);
                        } else if (tb1.kind != PR_FLAT)
                            prior += prior_term_outofline(&s_seg[0].t, v);  // (the entry in LDS: a pointer to the register copy would spill it)
                    } else {
                        const DimTab* tb = &s_seg[(segs >> (4 * e)) & 15u].t;
                        oob |= !(v >= tb->lo && v <= tb->hi);
                        if (tb->kind == PR_NORMAL) {  // (the bulk of any row: inline; a call per scalar is ~100 instructions and its spills)
                            const double z = (v - tb->a) * tb->b;
                            prior += tb->c - 0.5 * (z * z);
                        } else if (tb->kind != PR_FLAT)
                            prior += prior_term_outofline(tb, v);
                    }
                }
            }
        }
        DEMC_STAMP(5);  // proposal, bounds, prior of the lane's scalars
        if constexpr (HIST_ >= 2) {
            // Inside burn-in a particle's base row is ANY row of the group's current population: a row another WAVE is about to
            // write (its first-half particle's accepted proposal, held back until now, or -- at the end of this phase -- its
            // second-half particle's).  Every wave must therefore have its base rows in registers before any wave stores
            // anything: one workgroup barrier per launch, here.  (Found by the parity test of cfg3's shape the moment the kernel
            // got faster: 197 of 39 k decisions differed; with the slower register allocation the race had never fired.)
            // The base rows are GLOBAL loads: the barrier must also wait for them (lds_barrier() orders LDS traffic only, and
            // the race would then be closed by the register allocation of the day, not by the program).
            if (ph == 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
        }
        if constexpr (HIST) {
            if (pend) {  // the first half's writes, now that this half's rows are in registers
                if constexpr (PARK) unpark8(pend_l, pend_x);
                hist_store(0, pend_acc, pend_wp, pend_w, store_row, pend_x);
                pend = false;
            }
        }
        prior = subgroup_sum(prior, 4);
        oob = subgroup_sum(oob, 4);
        // ---- y = A^-1 (theta' - xbar) on the matrix cores: centred rows through LDS into operand order ----
        if constexpr (!DIRECT8 && !ISO) {
            double* row = scr + (size_t)q * scr_stride;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = (e < 4 ? jA : jB - 4) + e;
                if (j < D) row[j] = v8[e] - xb[e];
            }
        }
        double aux, S = 0.0, sg_iso = 1.0;
        if constexpr (ISO) {
            // A = I: the lane's share of |mu~|^2 and mu~ . sum_i x~_i, and sigma from the lane that holds scalar d
            double a_ = 0.0, s_ = 0.0, sg_ = 0.0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = (e < 4 ? jA : jB - 4) + e;
                const int jl = j < 32 ? j : 0;
                const double c = v8[e] - iso_l[jl];
                if (j < d) {
                    a_ = fma(c, c, a_);
                    s_ = fma(c, iso_l[32 + jl], s_);
                }
                if (j == d) sg_ = v8[e];
            }
            aux = subgroup_sum(a_, 4);
            S = subgroup_sum(s_, 4);
            sg_iso = subgroup_sum(sg_, 4);  // (three zeros and sigma)
        } else if constexpr (DIRECT8) {
            // d = 8: the 8 x 8 product on the vector pipe inside the quad -- lane m holds the centred scalars 2 m and 2 m + 1,
            // quad_perm hands them round, lane sl forms columns 2 sl and 2 sl + 1 (16 FMAs) and its share of theta~.y.
            // The MFMA route (LDS transposition, two matrix instructions, four 16-lane reductions, LDS again) is a chain
            // of ~2.3 k cycles for 64 FMAs per particle.
            const double c0_ = v8[0] - xb[0], c1_ = v8[1] - xb[1];
            double c[8];
            c[0] = dpp_mov<0x00>(c0_); c[1] = dpp_mov<0x00>(c1_);  // quad_perm:[0,0,0,0]
            c[2] = dpp_mov<0x55>(c0_); c[3] = dpp_mov<0x55>(c1_);  // quad_perm:[1,1,1,1]
            c[4] = dpp_mov<0xAA>(c0_); c[5] = dpp_mov<0xAA>(c1_);  // quad_perm:[2,2,2,2]
            c[6] = dpp_mov<0xFF>(c0_); c[7] = dpp_mov<0xFF>(c1_);  // quad_perm:[3,3,3,3]
            double y0 = 0.0, y1 = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                y0 = fma(c[k], ai8[0][k], y0);
                y1 = fma(c[k], ai8[1][k], y1);
            }
            aux = subgroup_sum(fma(c1_, y1, c0_ * y0), 4);
            if (!STREAM)
                S = subgroup_sum(fma(y1, sx8[1], y0 * sx8[0]), 4);
            else
                *reinterpret_cast<double2*>(ybuf + (size_t)q * p.dpad + 2 * sl) = make_double2(y0, y1);
        } else {
            wave_lds_sync();
            const int kq = lane >> 4, rowi = lane & 15;
            const int wrow0 = wave * 16;  // the wave's 16 particles are rows 0..15 of the A operand
            const double* trow = scr + (size_t)(wrow0 + rowi) * scr_stride;
            d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
            const bool two_tiles = d > 16;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                if (4 * ks < d) {
                    const int k = 4 * ks + kq;
                    const double a = k < d ? trow[k] : 0.0;
                    const double b0 = BF_LDS ? bf_l[ks * 64 + lane] : bfrag[0][ks];
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc0, 0, 0, 0);
                    if (two_tiles) {
                        const double b1 = BF_LDS ? bf_l[(8 + ks) * 64 + lane] : bfrag[1][ks];
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc1, 0, 0, 0);
                    }
                }
            double a4[4], s4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pr = kq + 4 * r;
                double* tr = scr + (size_t)(wrow0 + pr) * scr_stride;
                const double y0 = mc0 < d ? acc0[r] : 0.0, y1 = mc1 < d ? acc1[r] : 0.0;
                double a_ = 0.0, s_ = 0.0;
                if (mc0 < d) a_ = fma(tr[mc0], y0, a_);
                if (mc1 < d) a_ = fma(tr[mc1], y1, a_);
                if (!STREAM)
                    s_ = fma(y1, sx1, y0 * sx0);
                else {
                    double* yrow = ybuf + (size_t)(wrow0 + pr) * p.dpad;
                    if (mc0 < d) yrow[mc0] = y0;
                    if (mc1 < d) yrow[mc1] = y1;
                }
                a4[r] = a_;
                s4[r] = s_;
            }
            {  // the four rows' dot products summed over their 16 columns: one fold each (fold4_over16), not four sums each
                const double af = fold4_over16(a4[0], a4[1], a4[2], a4[3], lane);
                const double sf = STREAM ? 0.0 : fold4_over16(s4[0], s4[1], s4[2], s4[3], lane);
                if ((lane & 12) == 0) {
                    double* tr = scr + (size_t)(wrow0 + kq + 4 * fold4_row(lane)) * scr_stride;
                    tr[D] = af;
                    tr[D + 1] = sf;
                }
            }
            wave_lds_sync();
            aux = scr[(size_t)q * scr_stride + D];
            if (!STREAM) S = scr[(size_t)q * scr_stride + D + 1];
        }
        DEMC_STAMP(7);  // A^-1 product and its dot products
        if (STREAM) {
            // ---- the observation stream: cross terms of the phase's proposals against this workgroup's chunk of tiles ----
            lds_barrier();  // (LDS only: see the end of the phase)
            DEMC_STAMP(16);  // every proposal of the phase prepared
            if constexpr (DIR) {
                // sum_i |z_i - m|^2 over the chunk's rows for every proposal of the phase.  A thread holds HALF a row's worth of m for PB
                // proposals in registers and walks every n-th row of the chunk: what it reads of a row from LDS serves PB proposals,
                // and the two halves of a row are read by neighbouring lane groups -- a wave's eight groups then touch FOUR consecutive
                // rows (64 bytes apart: all 64 banks once) instead of eight (rows r and r + 4 on the same banks: every read a 2-way
                // conflict, the LDS pipe as busy as the vector pipe and the two serialised).  One proposal per thread and whole rows
                // (the first form) left the LDS pipe four times as busy: 17 us of a 21 us phase at BASELINE cfg2.
                constexpr int PB = 4, HD = DT / 2;                                           // proposals per thread, dims per half row
                const int n_pg = (n_act + PB - 1) / PB;                                      // groups of PB proposals
                const int pg_pad = n_pg <= 4 ? 4 : n_pg <= 8 ? 8 : 16;                       // (wave-uniform; n_act <= 64)
                const int pgi = tid & (pg_pad - 1), hs = (tid / pg_pad) & 1, slice = tid / (2 * pg_pad), n_slices = WG / (2 * pg_pad);
                double m[PB][HD];
#pragma unroll
                for (int j = 0; j < PB; ++j)
#pragma unroll
                    for (int k = 0; k < HD; ++k)  // (proposals pgi, pgi + pg_pad, ...: neighbouring lanes read neighbouring m rows -- with
                                                  // proposals pgi PB .. pgi PB + PB - 1 the rows of a wave's lanes sat PB x 64 bytes apart, all on the same banks)
                        m[j][k] = pgi + j * pg_pad < n_act ? ybuf[(size_t)(pgi + j * pg_pad) * DT + hs * HD + k] : 0.0;  // (dpad == DT)
                DEMC_STAMP(20);  // m in registers
                const long long o_hi = (long long)xt_hi * 16 < p.N ? (long long)xt_hi * 16 : p.N;
                const int n_obs = xt_lo < xt_hi ? (int)(o_hi - (long long)xt_lo * 16) : 0;
                double a2[PB][2];
#pragma unroll
                for (int j = 0; j < PB; ++j) a2[j][0] = a2[j][1] = 0.0;
                const lds_cptr zs = (lds_cptr)xs + hs * HD;  // (xs went through an integer round-up: without the cast the row reads are FLAT loads)
                // UR rows per trip, all of them read before the first is used: the compiler puts a trip's reads at its top and waits for
                // each where it is consumed, so the LDS round trip is paid once per UR rows (with one row per trip the younger of a
                // SIMD's two waves ended 4.5 k cycles after the older one: a 10 k-cycle loop, 15 k until the barrier let go)
                constexpr int UR = HD <= 4 ? 4 : 1;  // (D = 32: a row half is sixteen doubles -- one row per trip keeps the instance off scratch)
                auto accumulate = [&](const double (&zc)[HD]) {
#pragma unroll
                    for (int k = 0; k < HD; k += 2) {
                        double r[PB][2];  // (the 2 x PB residuals of a dim pair before the 2 x PB FMAs that consume them)
#pragma unroll
                        for (int j = 0; j < PB; ++j) {
                            r[j][0] = zc[k] - m[j][k];
                            r[j][1] = zc[k + 1] - m[j][k + 1];
                        }
#pragma unroll
                        for (int j = 0; j < PB; ++j) {
                            a2[j][0] = fma(r[j][0], r[j][0], a2[j][0]);
                            a2[j][1] = fma(r[j][1], r[j][1], a2[j][1]);
                        }
                    }
                };
                int o = slice;
                for (; o + (UR - 1) * n_slices < n_obs; o += UR * n_slices) {
                    double zr[UR][HD];
#pragma unroll
                    for (int u = 0; u < UR; ++u) {
                        const lds_cptr z = zs + (size_t)(o + u * n_slices) * DT;
#pragma unroll
                        for (int k = 0; k < HD; ++k) zr[u][k] = z[k];
                    }
#pragma unroll
                    for (int u = 0; u < UR; ++u) accumulate(zr[u]);
                }
                for (; o < n_obs; o += n_slices) {  // (the last rows of the slice)
                    const lds_cptr z = zs + (size_t)o * DT;
                    double zc[HD];
#pragma unroll
                    for (int k = 0; k < HD; ++k) zc[k] = z[k];
                    accumulate(zc);
                }
                DEMC_STAMP(21);  // residual loop done
                // The lanes of a 16-lane row that hold the same proposals (the two halves of the rows, the row's slices) are added on
                // the DPP network; every ROW then leaves its partial sums in LDS and the hand-over below adds the WG / 16 of them in a
                // fixed order.  (Summed across the wave with __shfl_xor -- sixteen dependent ds_bpermute round trips a thread -- this
                // step took 6 k cycles of a 25 k-cycle phase.)
#pragma unroll
                for (int j = 0; j < PB; ++j) {
                    double v = a2[j][0] + a2[j][1];
                    if (pg_pad <= 8) v += dpp_mov<0x128>(v);  // row_ror:8
                    if (pg_pad <= 4) v += dpp_mov<0x124>(v);  // row_ror:4
                    if ((lane & 15) < pg_pad && pgi + j * pg_pad < n_act) part_l[(size_t)(tid >> 4) * nact_max + pgi + j * pg_pad] = v;
                }
                DEMC_STAMP(15);  // row partials in LDS
            } else {
                const int nw_ = WG / 64;
                const int nt = xt_hi - xt_lo, per_w = (nt + nw_ - 1) / nw_;
                const int t_lo = wave * per_w < nt ? wave * per_w : nt, t_hi = t_lo + per_w < nt ? t_lo + per_w : nt;
                lds_ptr outw = (lds_ptr)(part_l + (size_t)wave * nact_max);
                if constexpr (DT == 8 || DT == 32) {  // dpad = DT: the k-step count is a constant of the instance
                    if (p.st_x_lds)
                        cross_ks<DT / 4, lds_cptr, true>((lds_cptr)ybuf, DT, n_act, (lds_cptr)xs, t_lo, t_hi, p.st_chunk_tiles, outw, lane);
                    else
                        cross_ks<DT / 4, glb_cptr>((lds_cptr)ybuf, DT, n_act, (glb_cptr)(p.Xf + (size_t)xt_lo * (DT >> 2) * 64), t_lo, t_hi,
                                                   p.n_tiles - xt_lo, outw, lane);
                } else if (p.st_x_lds)
                    cross_stage<lds_cptr>((lds_cptr)ybuf, p.dpad, n_act, (lds_cptr)xs, t_lo, t_hi, p.st_chunk_tiles, outw, lane);
                else
                    cross_stage<glb_cptr>((lds_cptr)ybuf, p.dpad, n_act, (glb_cptr)(p.Xf + (size_t)xt_lo * (p.dpad >> 2) * 64), t_lo, t_hi,
                                          p.n_tiles - xt_lo, outw, lane);
            }
            lds_barrier();  // (LDS only: see the end of the phase)
            DEMC_STAMP(17);  // cross terms of this workgroup's chunk done
            const unsigned epoch = (unsigned)(step + 1);
            unsigned long long* gran = p.st_gran + (((size_t)(step & 1) * p.n_groups + gi) * p.st_C) * nact_max * 2;
            if constexpr (DIR) {
                // a partial per 16-lane row (WG / 16 of them): LPS lanes per proposal add LPS-strided shares of them, the shares meet on
                // the DPP network -- a fixed tree, the same bits in every workgroup of the group -- and the proposal's first lane stores
                // the granules.  (One lane per proposal adding all 32 in turn was 2.3 k cycles of the phase.)
                constexpr int LPS = WG / 64;  // 8 lanes per proposal at 512 threads, 4 at 256: 64 proposals at most
                const int pq8 = tid / LPS, r8 = tid % LPS;
                double v = 0.0;
                if (pq8 < n_act) {
#pragma unroll
                    for (int wv = 0; wv < WG / 16; wv += LPS) v += part_l[(size_t)(wv + r8) * nact_max + pq8];
                }
                v = subgroup_sum(v, LPS);
                if (pq8 < n_act && r8 == 0) {
                    unsigned long long* mine_g = gran + ((size_t)c_idx * nact_max + pq8) * 2;
                    store_granule(mine_g, epoch, (unsigned)__double2loint(v));
                    store_granule(mine_g + 1, epoch, (unsigned)__double2hiint(v));
                }
            } else if (tid < n_act) {
                double v = 0.0;
                for (int wv = 0; wv < WG / 64; ++wv) v += part_l[(size_t)wv * nact_max + tid];
                unsigned long long* mine_g = gran + ((size_t)c_idx * nact_max + tid) * 2;
                store_granule(mine_g, epoch, (unsigned)__double2loint(v));
                store_granule(mine_g + 1, epoch, (unsigned)__double2hiint(v));
            }
            DEMC_STAMP(18);  // granules stored
            DEMC_STAMP_AT(22, 0, (double)(__builtin_amdgcn_s_memrealtime() & 0xffffffffull));  // ... on the 100 MHz clock all CUs share (hand-over skew: tools/k1_stamps.py)
            if (step + 1 < n_steps) draw_phase(step + 1, pre_mine, pre_nzA, pre_nzB);  // while the granules travel
            DEMC_STAMP(11);  // next phase's blocks drawn
            // every lane collects what ITS particle needs and nothing else: lane sl of the quad polls chunks sl, sl + 4, ...
            // (both granules of a double, all of the lane's chunks in flight together) until their tags carry this epoch -- no
            // staging in LDS, no workgroup barrier inside the wait, and the waves without a moving particle do not wait at
            // all.  The quad then adds its four partial sums as a fixed tree: the same bits in every workgroup of the group.
            if (valid) {
                double part = 0.0;
                unsigned spins = 0;
                auto collect = [&](auto ns_c, int c0) {  // NS chunks of this lane (c0, c0 + 4, ...): 2 NS granules in flight
                    constexpr int NS = decltype(ns_c)::value;
                    unsigned long long x[NS][2];
                    for (;;) {
                        bool ok = true;
#pragma unroll
                        for (int i = 0; i < NS; ++i) {
                            const int cc = c0 + 4 * i < p.st_C ? c0 + 4 * i : c0;
                            const unsigned long long* gq = gran + ((size_t)cc * nact_max + q) * 2;
                            x[i][0] = load_granule(gq);
                            x[i][1] = load_granule(gq + 1);
                        }
#pragma unroll
                        for (int i = 0; i < NS; ++i) ok &= (unsigned)(x[i][0] >> 32) == epoch && (unsigned)(x[i][1] >> 32) == epoch;
                        if (ok) break;
                        if (++spins > (1u << 22)) {  // cannot happen with co-resident workgroups; never spin unbounded
                            *p.st_err = 1u;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);  // (without it: the same)
                    }
#pragma unroll
                    for (int i = 0; i < NS; ++i)
                        if (c0 + 4 * i < p.st_C) part += __hiloint2double((int)(unsigned)x[i][1], (int)(unsigned)x[i][0]);
                };
                for (int c0 = sl; c0 < p.st_C; c0 += 16) {
                    if (p.st_C <= 8) collect(std::integral_constant<int, 2>(), c0);  // (cfg2: two chunks per lane)
                    else collect(std::integral_constant<int, 4>(), c0);
                }
                S = subgroup_sum(part, 4);
            }
        }
        DEMC_STAMP(8);  // (STREAM: cross terms streamed and handed over)
        DEMC_STAMP(19);
        DEMC_STAMP_AT(23, 0, (double)(__builtin_amdgcn_s_memrealtime() & 0xffffffffull));
        // ---- compute_posterior! + mh_update! + store_samples! (utilities.jl:92-99, 55-58, 201-210, 161-180) ----
        const double w = HIST ? w_h : w_s[pl];
        double wp;
        if constexpr (ISO) {  // loglike_from_stats' MVN_ISO form (c1 = sum_i |x~_i|^2)
            const double nd = (double)p.N * (double)d;
            wp = oob ? -INFINITY : prior + (-0.5 * nd * kLog2Pi - nd * log(sg_iso) - 0.5 * (p.c1 - 2.0 * S + (double)p.N * aux) / (sg_iso * sg_iso));
        } else if constexpr (DIR)
            wp = oob ? -INFINITY : prior + (p.c0 - 0.5 * S);  // loglike_from_stats' DIRECT form: S = sum_i |L^-1 (x_i - mu)|^2
        else
            wp = oob ? -INFINITY : prior + (p.c0 - 0.5 * (p.c1 - 2.0 * S + (double)p.N * aux));
        double adj = 0.0;
        if constexpr (HSNK) {
            if (__ballot(snk) != 0ull) {
                s1 = subgroup_sum(s1, 4); s2 = subgroup_sum(s2, 4);
                if (snk) adj = (double)(D - 1) * (0.5 * log(s1) - 0.5 * log(s2));  // adjust_loglike, in the form that does not overflow
            }
        }
        const double ex = HSNK ? exp(wp - w + adj) : exp(wp - w);
        const int acc = (ex >= 1.0) || (u_acc <= ex);
        if constexpr (HIST) {
            double x8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x8[e] = acc ? v8[e] : t8[e];
            if (step + 1 < n_steps) {
                if constexpr (PARK)
                    park8(pend_l, x8);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) pend_x[e] = x8[e];
                }
                pend_acc = acc; pend_wp = wp; pend_w = w; pend = true;
            } else
                hist_store(ph, acc, wp, w, store_row, x8);
        } else if (valid) {
            if (sl == 0) {
                if (acc) {
                    if (wr_hbm) p.weight[slot] = wp;
                    w_s[pl] = wp;
                }
                if (store_row >= 0 && wr_hbm) {
                    const size_t hrow = (size_t)store_row * p.P + slot;
                    p.acc_hist[hrow] = (unsigned char)acc;
                    p.lp_hist[hrow] = acc ? wp : w;
                    p.id_hist[hrow] = ph ? id_hi : id_lo;
                }
            }
            double* trow = p.theta + slot * D;
            double* lrow = tile + (size_t)pl * D;
            double* hrow = (store_row >= 0 && wr_hbm) ? p.hist + ((size_t)store_row * p.P + slot) * p.hist_ld : nullptr;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const int j = blk ? jB : jA;
                if (j >= D) continue;
                if (!acc && !hrow) continue;
                if constexpr (DT == 8) {  // the lane's two scalars: one 16-byte access each way
                    const double2 a01 = acc ? make_double2(v8[0], v8[1]) : *reinterpret_cast<const double2*>(lrow + j);
                    if (acc) {
                        *reinterpret_cast<double2*>(lrow + j) = a01;
                        if (wr_hbm) *reinterpret_cast<double2*>(trow + j) = a01;
                    }
                    if (hrow) *reinterpret_cast<double2*>(hrow + j) = a01;
                } else if (j + 3 < D && (D & 3) == 0) {  // whole block: 32-byte accesses (a rejected particle's row comes from the tile)
                    const double2 a01 = acc ? make_double2(v8[4 * blk], v8[4 * blk + 1]) : *reinterpret_cast<const double2*>(lrow + j);
                    const double2 a23 = acc ? make_double2(v8[4 * blk + 2], v8[4 * blk + 3]) : *reinterpret_cast<const double2*>(lrow + j + 2);
                    if (acc) {
                        *reinterpret_cast<double2*>(lrow + j) = a01; *reinterpret_cast<double2*>(lrow + j + 2) = a23;
                        if (wr_hbm) { *reinterpret_cast<double2*>(trow + j) = a01; *reinterpret_cast<double2*>(trow + j + 2) = a23; }
                    }
                    if (hrow) { *reinterpret_cast<double2*>(hrow + j) = a01; *reinterpret_cast<double2*>(hrow + j + 2) = a23; }
                } else
                    for (int e = 0; e < 4 && j + e < D; ++e) {
                        const double x = acc ? v8[4 * blk + e] : lrow[j + e];
                        if (acc) {
                            lrow[j + e] = x;
                            if (wr_hbm) trow[j + e] = x;
                        }
                        if (hrow) hrow[j + e] = x;
                    }
            }
        }
        DEMC_STAMP(9);  // accept + row moves
        // LDS only.  __syncthreads() also waits for this phase's stores to HBM (state, weights, history) to be acknowledged,
        // which nothing in the kernel reads back: that was 1.7 k cycles of every phase of the SUFFSTAT form, and in the
        // streaming form it made workgroup 0 of each group -- the one that writes HBM -- 0.4 us late for the next hand-over,
        // with its seven peers waiting (tools/k1_stamps.py, slots 22 / 23).  Stores to one address stay in issue order.
        if constexpr (!HIST) lds_barrier();  // the other colour reads what this phase wrote (rows, weights)
        DEMC_STAMP(10);
    }
}

// Every instance the runtime launches, in one list: demc_resmvn.cpp instantiates them (a translation unit of its own, compiled
// beside the others), demc_hip.cpp declares them extern (DEMC_RESMVN_EXTERN).  <WG, STREAM, DT, HIST, OCC, ISO>
#define DEMC_RESMVN_INSTANCES(X) \
    X(256, false, 0, 0, 1, false) \
    X(256, false, 0, 1, 1, false) \
    X(256, false, 0, 2, 1, false) \
    X(256, false, 0, 3, 1, false) \
    X(256, false, 8, 0, 1, false) \
    X(256, false, 8, 1, 1, false) \
    X(256, false, 8, 2, 1, false) \
    X(256, false, 8, 3, 1, false) \
    X(256, false, 32, 0, 1, false) \
    X(256, false, 32, 1, 1, false) \
    X(256, false, 32, 2, 1, false) \
    X(256, false, 32, 3, 1, false) \
    X(256, false, 0, 1, 1, true) \
    X(256, false, 0, 2, 1, true) \
    X(256, false, 0, 3, 1, true) \
    X(256, false, 31, 1, 1, true) \
    X(256, false, 31, 2, 1, true) \
    X(256, false, 31, 3, 1, true) \
    X(512, false, 0, 0, 1, false) \
    X(512, false, 0, 1, 1, false) \
    X(512, false, 0, 2, 1, false) \
    X(512, false, 0, 3, 1, false) \
    X(512, false, 8, 0, 1, false) \
    X(512, false, 8, 1, 1, false) \
    X(512, false, 8, 2, 1, false) \
    X(512, false, 8, 3, 1, false) \
    X(512, false, 32, 0, 1, false) \
    X(512, false, 32, 1, 1, false) \
    X(512, false, 32, 2, 1, false) \
    X(512, false, 32, 3, 1, false) \
    X(512, false, 0, 1, 1, true) \
    X(512, false, 0, 2, 1, true) \
    X(512, false, 0, 3, 1, true) \
    X(512, false, 31, 1, 1, true) \
    X(512, false, 31, 2, 1, true) \
    X(512, false, 31, 3, 1, true) \
    X(256, true, 0, 0, 1, false) \
    X(256, true, 8, 0, 1, false) \
    X(256, true, 32, 0, 1, false)
#define DEMC_RESMVN_INSTANCES_DIR(X) X(512, true, 8, 0, 1, false, true) X(256, true, 32, 0, 1, false, true)  // the DIRECT streaming-resident form
#ifdef DEMC_EXPERIMENTS
#define DEMC_RESMVN_INSTANCES_EXP(X) X(256, true, 8, 0, 2, false)  // A/B only: two streaming workgroups per CU
#else
#define DEMC_RESMVN_INSTANCES_EXP(X)
#endif
#ifdef DEMC_RESMVN_EXTERN
#define DEMC_X_(...) extern template __global__ void k_res_mvn<__VA_ARGS__>(KParams);
DEMC_RESMVN_INSTANCES(DEMC_X_)
DEMC_RESMVN_INSTANCES_DIR(DEMC_X_)
DEMC_RESMVN_INSTANCES_EXP(DEMC_X_)
#undef DEMC_X_
#endif

}  // namespace demc
