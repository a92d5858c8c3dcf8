// demc_longrow.hpp -- K1 for LONG rows (D in the thousands: hierarchical models with one scalar per subject, BASELINE cfg4).
//
// One workgroup of WG threads per moving particle, the whole update in one launch (proposal, bounds, prior, the
// hierarchical likelihood, Metropolis accept, state and history write-back), like k_propose's fused tail -- but built for
// rows that stream from HBM/L2 instead of an LDS tile:
//   * ONE pass over the row.  The proposal of scalar j, its prior term AND the likelihood term of the subject behind it are
//     formed together; the few scalars every term needs (the hyper-parameters: theta'[0], theta'[ref], the observation sd)
//     are proposed first by every lane for itself.  k_propose makes two passes (proposal, then the subjects from an LDS copy).
//   * The row loads of a lane's NEXT dim pair (current row, partners, base row, subject data) are issued before the Philox
//     rounds and the arithmetic of the current pair, so a lane always has one round trip to L2/HBM in flight.
//   * The per-particle scalars (coins, partner indices, gammas, accept uniform) are drawn by four lanes of every wave and
//     shared through SGPRs -- no LDS, no barrier; the base pick (burn-in only) is the one step that waits for the group's
//     cumulative weights, and the first batch of row loads and noise draws goes out before that wait.
//   * Scalars outside the block of a block sweep (reset!, crossover.jl:336-352) cost neither noise draws nor partner loads.
// Same addressed draws, same arithmetic and the same lane -> dim-pair mapping as k_propose with a workgroup per particle,
// so proposals, priors and decisions are the ones that kernel produces (tests/test_gpu_parity.py::test_longrow_*).
//
// Reference: crossover!/snooker_update!/mutation!/recombination!/reset!/in_bounds/compute_posterior!/mh_update!/
// store_samples! (crossover.jl:30-99,154-257,301-352; mutation.jl:13-25; utilities.jl:70-99,161-180,201-226).
#pragma once
#include "demc_kernels.hpp"

namespace demc {

// value held by lane `src` of the wave, as a wave-uniform scalar
__device__ inline uint32_t wave_get(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }

template <int WG>
__global__ __launch_bounds__(WG, 1) void k_longrow(KParams p) {
    extern __shared__ double lds[];  // theta' of the particle [D (+1 if odd)] | cumulative pool weights [pool_n + chunks]
    __shared__ double s_red[5][WG / 64];
    __shared__ int s_redi[WG / 64];
    __shared__ DimSeg s_seg[kMaxDimSeg];
    __shared__ int s_base;
    __shared__ double s_ref[2][kMaxDimSeg];
    DEMC_STAMP_INIT();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = p.D, Np = p.Np;
    for (int i = tid; i < p.n_seg * (int)(sizeof(DimSeg) / sizeof(double)); i += WG)
        reinterpret_cast<double*>(s_seg)[i] = reinterpret_cast<const double*>(p.dimseg)[i];

    // particle of this workgroup; the particles of a group share an XCD (their partner rows then share its L2)
    int g, qg;
    if ((p.n_groups & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        qg = j % p.n_act;
        g = (j / p.n_act) * 8 + xcd;
    } else {
        g = blockIdx.x / p.n_act;
        qg = blockIdx.x % p.n_act;
    }
    if (p.glist) g = p.glist[g];
    const int pl = p.a_lo + qg;
    const size_t slot = (size_t)g * Np + pl;
    const int g_glob = p.group_offset + g;
    const uint32_t eslot = (uint32_t)g_glob * (uint32_t)Np + (uint32_t)pl;
    const double* grows = p.theta + (size_t)g * Np * D;
    const double* gw = p.weight + (size_t)g * Np;
    const double* pt = grows + (size_t)pl * D;
    double* scr = lds;
    double* cdf = lds + ((D + 1) & ~1);
    const bool even = (D & 1) == 0;
    const double w_cur = gw[pl];

    // ---- per-particle scalars: lanes 0..5 of every wave evaluate one Philox block each, the words travel through SGPRs ----
    bool is_mut = false;
    {
        const U4 r = draw_block(p.seed, S_GROUP, p.sweep, (uint64_t)p.iter, (uint32_t)g_glob, 0);
        is_mut = p.mode == MODE_STEP && u53(r.x, r.y) <= p.beta;  // mutate_or_crossover! main.jl:199-207
    }
    const bool hist_partners = p.partner_kind == 1;
    const U4 mine = draw_block(p.seed, S_PART, p.sweep, (uint64_t)p.iter, eslot, (uint32_t)(lane < 6 ? lane : 0));
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
                Pa = p.hist + ((a % ub) * (uint64_t)p.P + a / ub) * (uint64_t)D;
                Pb2 = p.hist + ((b % ub) * (uint64_t)p.P + b / ub) * (uint64_t)D;
                Pc = p.hist + ((c % ub) * (uint64_t)p.P + c / ub) * (uint64_t)D;
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

    DEMC_STAMP(12);  // per-particle scalars drawn
    // ---- select_base (crossover.jl:282-289) over the partner pool: stabilised softmax, cumulative weights in the fixed
    // two-level order shared with the oracle and k_propose (chunks of 16, sequential inside and over the chunks).  Wave 0
    // alone; the other waves go straight to their first row loads and noise draws and meet it at the barrier below.
    const int n_cdf = p.pool_n;
    if (use_base && wave == 0) {
        const double* pw = gw + p.pool_lo;
        double m = -INFINITY;
        for (int i = lane; i < n_cdf; i += 64) m = fmax(m, pw[i]);
        m = wave_max(m);
        int b;
        if (n_cdf <= 256) {  // the usual case: the pool sits in this wave's registers, no LDS round trips (wave_cdf)
            double e[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) e[r] = (lane + 64 * r < n_cdf) ? exp(pw[lane + 64 * r] - m) : 0.0;
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
            double pre = 0.0;
            const int e1 = (c * 16 + 16 < n_cdf) ? c * 16 + 16 : n_cdf;
            for (int i = c * 16; i < e1; ++i) {
                pre += cdf[i];
                cdf[i] = pre;
            }
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
    auto load_pair = [&](int k, bool with_base) -> PairIn {
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
        const bool frozen = kind == 0 && r.keep0 && (r.keep1 || !has1);
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
    // proposal of both scalars of pair k from loaded rows (recombination! and reset! applied)
    auto propose_pair = [&](int k, const PairIn& in, bool base_on, double& v0, double& v1) {
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

    // This lane's dim pairs, in the order it visits them: both pairs of noise block tid, then of block tid + WG, ...
    auto pair_at = [&](int i) -> int { return 2 * (tid + (i >> 1) * WG) + (i & 1); };
    // first batch of row loads before anything waits: this lane's first pair (the base row follows once the pick is known)
    const int k_first = 2 * tid;
    const bool any = 2 * k_first < D;
    PairIn cur = {};
    if (any) cur = load_pair(k_first, false);

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
    __syncthreads();  // base pick (wave 0), snooker partial sums, segment table
    DEMC_STAMP(1);
    if (kind == 1) {
        double vm = 0.0, vn = 0.0, vd = 0.0;
        // the tree k_propose's group_sum uses: pairs of waves, then pairs of pairs
        auto tree = [&](const double* s) {
            const double lo = (s[0] + s[1]) + (s[2] + s[3]);
            return WG > 256 ? lo + ((s[4] + s[5]) + (s[6] + s[7])) : lo;
        };
        vm = tree(s_red[0]); vn = tree(s_red[1]); vd = tree(s_red[2]);
        cm = vm / vd; cn = vn / vd;
        __syncthreads();  // s_red is reused by the final reduction
    }
    bool base_on = false;
    if (use_base) {
        i2 = s_base;
        Pbase = grows + (size_t)i2 * D;
        base_on = true;
        if (any) {  // the one load that had to wait for the pick
            const int j0 = 2 * k_first;
            cur.c = even ? *reinterpret_cast<const double2*>(Pbase + j0) : make_double2(Pbase[j0], j0 + 1 < D ? Pbase[j0 + 1] : 0.0);
        }
    }

    // the hyper-parameter scalars: value of scalar j under the proposal (any lane may ask; pairs are cheap to re-propose)
    auto value_of = [&](int j) -> double {
        const PairIn in = load_pair(j >> 1, base_on);
        double a0, a1;
        propose_pair(j >> 1, in, base_on, a0, a1);
        return (j & 1) ? a1 : a0;
    };
    // family constants of the fused likelihood term
    double mu0 = 0.0, sg_obs = 1.0, lsg_obs = 0.0, isg_obs = 1.0;
    if (hier_b || hier_g) mu0 = value_of(0);
    if (hier_g) {
        sg_obs = value_of(2 + (int)S);
        lsg_obs = log(sg_obs);
        isg_obs = 1.0 / sg_obs;
    }
    const double n_bin = p.c0;
    DEMC_STAMP(4);  // hyper-parameter scalars proposed

    // Normal(a, theta'[ref]) priors (hierarchical scale): 1/scale and log scale per table segment, once
    if (tid < p.n_seg && s_seg[tid].t.kind == PR_NORMAL_REF) {
        const double sref = value_of(s_seg[tid].t.ref);
        s_ref[0][tid] = 1.0 / sref;
        s_ref[1][tid] = log(sref);
    }
    __syncthreads();
    int oob = 0;
    double prior = 0.0, like = 0.0, s1 = 0.0, s2 = 0.0;
    unsigned long long wbits = 0;  // which of this lane's pairs lie (partly) inside the block of the sweep
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
        propose_pair(k, cur, base_on, v0, v1);
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
                    like += -n_bin * softplus_fast(-eta) - (n_bin - kk) * eta;
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
    // the rows of this lane's NEXT pair go out before the current pair's noise is drawn and its terms are formed.  (Deeper
    // prefetch -- three pairs in flight -- was measured and gains nothing: the pass is bound by VALU issue, Philox rounds
    // and the softplus of the likelihood, not by memory latency)
    // (a block's second pair is missing only in the very last block of an odd-ish row, which ends its lane's sequence)
    for (int i = 0; 2 * pair_at(i) < D; ++i) {
        const int k = pair_at(i), kn = pair_at(i + 1);
        PairIn nxt = cur;
        if (2 * kn < D) nxt = load_pair(kn, base_on);
        process(k, cur);
        cur = nxt;
    }
    DEMC_STAMP(5);  // the pass over the row done
    // ---- one reduction for everything: waves on the DPP network, then the fixed tree over the waves through LDS ----
    prior = subgroup_sum(prior, 64); like = subgroup_sum(like, 64); oob = subgroup_sum(oob, 64);
    if (kind == 1) { s1 = subgroup_sum(s1, 64); s2 = subgroup_sum(s2, 64); }
    if (lane == 0) {
        s_red[0][wave] = prior; s_red[1][wave] = like; s_red[2][wave] = s1; s_red[3][wave] = s2;
        s_redi[wave] = oob;
    }
    __syncthreads();
    auto tree = [&](const double* s) {
        const double lo = (s[0] + s[1]) + (s[2] + s[3]);
        return WG > 256 ? lo + ((s[4] + s[5]) + (s[6] + s[7])) : lo;
    };
    prior = tree(s_red[0]); like = tree(s_red[1]);
    if (hier_b) like = p.c2 + like;  // + sum_s log C(n, k_s): data-only, summed once at demc_set_model
    oob = 0;
    for (int i = 0; i < WG / 64; ++i) oob |= s_redi[i];
    double adj = 0.0;
    if (kind == 1) adj = (double)(D - 1) * (0.5 * log(tree(s_red[2])) - 0.5 * log(tree(s_red[3])));

    DEMC_STAMP(6);  // reductions done
    // ---- compute_posterior! + mh_update! + store_samples! ----
    double wp;
    if (p.fitness_kind == 1)
        wp = oob ? (p.update_kind == 1 ? -INFINITY : INFINITY) : like;
    else
        wp = oob ? -INFINITY : prior + like;
    const int acc = decide<false>(p, u_acc, wp, w_cur, adj);
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
    double* trow = p.theta + slot * D;
    double* hrow = (p.store_row >= 0) ? p.hist + ((size_t)p.store_row * p.P + slot) * D : nullptr;
    if (acc || hrow) {
        // in a block sweep an accepted crossover proposal differs from the row only inside the block: write only those
        const bool masked = acc && p.mask && kind != 2 && kind != 3;
        for (int it = 0; 2 * pair_at(it) < D; ++it) {  // the lane's pairs in the order the pass visited them (wbits)
            const int k = pair_at(it);
            const int j0 = 2 * k;
            const bool has1 = j0 + 1 < D;
            double v0, v1;
            if (acc) {
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
            const bool in_block = it < 64 ? ((wbits >> it) & 1ull) != 0 : (p.mask[j0] || (has1 && p.mask[j0 + 1]));
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
    DEMC_STAMP(9);   // accept + row moves done
    DEMC_STAMP(10);
}

}  // namespace demc
