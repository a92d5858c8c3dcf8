// demc_device.hpp -- device-side building blocks of the gfx950 DE-MCMC path:
// counter-based Philox4x32-10, draw addressing, and the registered log-densities.
//
// RNG contract (DESIGN.md "Randomness"): the reference consumes Julia's task-local Xoshiro
// stream in a fixed order.  A fused kernel cannot share one sequential
// stream, so every draw is addressed instead:
//     Philox4x32-10( key = seed,
//                    ctr = (block, entity, iteration, stream<<24 | sweep) )
// entity = global group index or global slot index, so results do not depend on how groups
// are sharded over GPUs.  One block = 4 x u32 = two 53-bit uniforms.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>

#include "demc_logphi_table.hpp"
#include "demc_phi_table.hpp"
#include "demc_softplus_table.hpp"

namespace demc {

constexpr double kLog2Pi = 1.8378770664093454835606594728112;
constexpr double kLogPi = 1.1447298858494001741434273513531;
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double kInvSqrt2 = 0.70710678118654752440;
constexpr double kInvSqrt2Pi = 0.39894228040143267794;

enum Stream : uint32_t { S_STEP = 1, S_GROUP = 2, S_PART = 3, S_NOISE = 4, S_RECOMB = 5, S_MIG = 6 };

struct U4 {
    uint32_t x, y, z, w;
};

__host__ __device__ inline U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
        U4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

__host__ __device__ inline U4 draw_block(uint64_t seed, uint32_t stream, uint32_t sweep, uint64_t iter, uint32_t entity,
                                         uint32_t block) {
    U4 c;
    c.x = block;
    c.y = entity;
    c.z = (uint32_t)iter;
    c.w = (stream << 24) | (sweep & 0xFFFFu);
    return philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// 53-bit uniform in [0,1), the resolution of Julia's rand(Float64)
__host__ __device__ inline double u53(uint32_t lo, uint32_t hi) {
    const uint64_t x = ((uint64_t)hi << 32) | lo;
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}
// 32-bit uniform in (0,1) from ONE word: the per-scalar draws (crossover noise b ~ U(-eps, eps), recombination coins, the
// Box-Muller inputs of the mutation noise) -- four scalars per Philox block: block m of the NOISE / RECOMB streams covers
// scalars 4m..4m+3 (word 2(k&1)+e for scalar e of dim pair k).  2^-32 resolution is ample for a jitter of scale eps.
__host__ __device__ inline double u32unit(uint32_t w) { return ((double)w + 0.5) * (1.0 / 4294967296.0); }
__host__ __device__ inline uint32_t mulhi32(uint32_t x, uint32_t m) { return (uint32_t)(((uint64_t)x * m) >> 32); }
__device__ inline uint64_t mulhi64(uint64_t x, uint64_t m) { return __umul64hi(x, m); }

// StatsBase.samplepair: i1 = rand(1:m); i2 = rand(1:m-1); i2 == i1 ? m : i2
__host__ __device__ inline void pick_pair(uint32_t r0, uint32_t r1, uint32_t m, uint32_t& i1, uint32_t& i2) {
    i1 = mulhi32(r0, m);
    uint32_t b = mulhi32(r1, m - 1);
    if (b == i1) b = m - 1;
    i2 = b;
}
// ordered 3-of-m without replacement (rank shift)
__host__ __device__ inline void pick_triple(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t m, uint32_t& i1, uint32_t& i2,
                                            uint32_t& i3) {
    const uint32_t a = mulhi32(r0, m);
    uint32_t b = mulhi32(r1, m - 1);
    if (b >= a) ++b;
    uint32_t c = mulhi32(r2, m - 2);
    const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
    if (c >= lo) ++c;
    if (c >= hi) ++c;
    i1 = a;
    i2 = b;
    i3 = c;
}

// ---- log-densities of the registered family (Distributions.jl forms) ----
__device__ inline double norm_logpdf(double x, double m, double s) {
    const double z = (x - m) / s;
    return -(z * z + kLog2Pi) / 2.0 - log(s);
}
__device__ inline double Phi(double x) { return 0.5 * erfc(-x * kInvSqrt2); }
__device__ inline double phi(double x) { return exp(-0.5 * x * x) * kInvSqrt2Pi; }
__device__ inline double softplus(double x) { return x > 0 ? x + log1p(exp(-x)) : log1p(exp(x)); }
// softplus with a short log1p: for t = exp(-|x|) in (0, 1], log1p(t) = 2 atanh(z), z = t / (2 + t) in (0, 1/3], whose odd
// series in z converges as 9^-n (17 terms reach 1e-17 relative); the quotient by reciprocal + two Newton steps (the
// denominator lies in [2, 3]: no scaling needed).  About a third of the instructions of the library's log1p on top of
// exp; agrees with softplus() to ~2e-16 relative (tests/test_gpu_parity.py compares the hierarchical families with the
// oracle's libm forms at 1e-9).  The hierarchical likelihoods spend most of their instructions here.
// exp(a) for a <= 0 without a quarter-rate instruction: the library's exp spends v_rndne_f64, v_cvt_i32_f64 and v_ldexp_f64 (16
// cycles a wave each, against 4 for an FP64 add) on range reduction and scaling.  Here: n = round(a log2 e) by adding and
// subtracting 1.5 * 2^52 (n sits in the low word of the sum), r = a - n ln 2 in two FMAs (hi / lo split), e^r by its degree-13
// Taylor polynomial (|r| <= 0.347: truncation 4e-18), and 2^n built in the exponent field by integer adds.  a is clamped at
// -700 (e^-700 ~ 1e-304: still a normal number; below it the callers' results do not change).  Relative error < 2.5e-16.
__device__ __forceinline__ double exp_nonpos(double a) {
    a = fmax(a, -700.0);
    const double kMagic = 6755399441055744.0;  // 1.5 * 2^52
    const double tn = fma(a, 1.4426950408889634074, kMagic);
    const double nf = tn - kMagic;
    const int n = __double2loint(tn);           // low word of the sum = n (two's complement, -1010 <= n <= 0)
    double r = fma(nf, -6.93147180369123816490e-01, a);
    r = fma(nf, -1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;              // 1/13!
    p = fma(p, r, 1.0 / 479001600.0); p = fma(p, r, 1.0 / 39916800.0); p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0); p = fma(p, r, 1.0 / 40320.0); p = fma(p, r, 1.0 / 5040.0); p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0); p = fma(p, r, 1.0 / 24.0); p = fma(p, r, 1.0 / 6.0); p = fma(p, r, 0.5);
    p = fma(p, r, 1.0); p = fma(p, r, 1.0);
    const double two_n = __hiloint2double((n + 1023) << 20, 0);  // 2^n, n + 1023 >= 13
    return p * two_n;
}
__device__ inline double softplus_fast(double x) {
    const double t = exp_nonpos(-fabs(x));
    const double den = 2.0 + t;
    double r = __builtin_amdgcn_rcp(den);
    r = fma(fma(-den, r, 1.0), r, r);
    r = fma(fma(-den, r, 1.0), r, r);
    double z = t * r;
    z = fma(fma(-den, z, t), r, z);
    const double w = z * z;
    double s = 1.0 / 33.0;
    s = fma(s, w, 1.0 / 31.0); s = fma(s, w, 1.0 / 29.0); s = fma(s, w, 1.0 / 27.0); s = fma(s, w, 1.0 / 25.0);
    s = fma(s, w, 1.0 / 23.0); s = fma(s, w, 1.0 / 21.0); s = fma(s, w, 1.0 / 19.0); s = fma(s, w, 1.0 / 17.0);
    s = fma(s, w, 1.0 / 15.0); s = fma(s, w, 1.0 / 13.0); s = fma(s, w, 1.0 / 11.0); s = fma(s, w, 1.0 / 9.0);
    s = fma(s, w, 1.0 / 7.0); s = fma(s, w, 1.0 / 5.0); s = fma(s, w, 1.0 / 3.0); s = fma(s, w, 1.0);
    return fmax(x, 0.0) + 2.0 * z * s;
}

// softplus from two small LDS tables (tools/gen_softplus_table.py: 2 576 bytes, copied in by load_softplus_table): about a third of
// the instructions of softplus_fast -- a hierarchical Binomial sweep is bound by the vector pipe's issue rate and spends most of its
// instructions here (profiles/r05/NOTES.md: the frozen loop ran 106 instructions a scalar, 66 of them FP64).
//   exp(a), a = -|x|:  a = k ln2/64 + r (k by adding 1.5 * 2^52: no quarter-rate conversion), e^r by its degree-5 polynomial
//                      (|r| <= ln2/128: truncation 3e-17), times E[k & 63] = 2^((k & 63)/64), times 2^(k >> 6) built in the exponent field
//   log1p(t):          j = round(128 t), f = (t - j/128) / c_j with c_j = 1 + j/128 (the difference is exact, 1/c_j from the table),
//                      log1p(t) = log c_j + f Q6(f), |f| <= 2^-8 (truncation f^7/8: 2e-18 relative to f)
// Relative error < 6e-16 (the generator checks a dense grid against 50-digit arithmetic); not bit-identical to softplus_fast.
__device__ __forceinline__ void load_softplus_table(double* s_tab, int tid, int n_threads) {
    for (int i = tid; i < kSpDoubles; i += n_threads) s_tab[i] = kSpTable[i];
}
// (the constants as wave-uniform values: the compiler keeps them in SGPR pairs and feeds v_fma_f64 directly, where a literal costs a
// v_mov_b64 into the accumulator in front of every v_fmac)
struct SoftplusTab {
    const double* tab;
    double magic, magic7, c, ln2hi, ln2lo, e5, e4, e3, q7, q6, q5, q4, q3, floor_;
};
__device__ __forceinline__ SoftplusTab softplus_tab_consts(const double* s_tab) {
    auto u = [](double x) {
        asm volatile("" : "+s"(x));
        return x;
    };
    // (an FMA takes ONE scalar operand: where a step has two constants the multiplicand lives in a vector register pair and the
    // addend in a scalar one -- left alone the compiler parks the ADDEND in vector registers and copies it in front of a v_fmac)
    auto v = [](double x) {
        asm volatile("" : "+v"(x));
        return x;
    };
    SoftplusTab c;
    c.tab = s_tab;
    c.magic = u(6755399441055744.0);   // 1.5 * 2^52: the sum's last place is 1
    c.magic7 = u(52776558133248.0);    // 1.5 * 2^45: the sum's last place is 2^-7
    c.c = v(kSpC); c.ln2hi = u(-kSpLn2Hi); c.ln2lo = u(-kSpLn2Lo);
    c.e5 = v(1.0 / 120.0); c.e4 = u(1.0 / 24.0); c.e3 = u(1.0 / 6.0);
    c.q7 = v(1.0 / 7.0); c.q6 = u(-1.0 / 6.0); c.q5 = u(1.0 / 5.0); c.q4 = u(-1.0 / 4.0); c.q3 = u(1.0 / 3.0);
    c.floor_ = u(-700.0);
    return c;
}
__device__ __forceinline__ double softplus_tab(double x, const SoftplusTab& c) {
    const double a = fmax(-fabs(x), c.floor_);
    const double tk = fma(a, c.c, c.magic);
    const double kf = tk - c.magic;
    const int k = __double2loint(tk);  // low word of the sum = k (two's complement, -64 700 <= k <= 0)
    double r = fma(kf, c.ln2hi, a);
    r = fma(kf, c.ln2lo, r);
    double pe = fma(c.e5, r, c.e4);
    pe = fma(pe, r, c.e3); pe = fma(pe, r, 0.5); pe = fma(pe, r, 1.0); pe = fma(pe, r, 1.0);
    const double tp = c.tab[k & (kSpExpN - 1)] * pe;  // in [0.99, 2): times 2^(k >> 6), k >> 6 >= -1011, by an add in the exponent field
    const double t = __hiloint2double(__double2hiint(tp) + ((k << 14) & (int)0xfff00000), __double2loint(tp));
    const double tj = t + c.magic7;
    const double jd = tj - c.magic7;   // round(128 t) / 128
    const int j = __double2loint(tj);  // round(128 t): 0 .. 128
    const double2 row = *reinterpret_cast<const double2*>(c.tab + kSpExpN + 2 * j);  // (log c_j, 1 / c_j)
    const double f = (t - jd) * row.y;
    double q = fma(c.q7, f, c.q6);
    q = fma(q, f, c.q5); q = fma(q, f, c.q4); q = fma(q, f, c.q3); q = fma(q, f, -0.5); q = fma(q, f, 1.0);
    return fmax(x, 0.0) + fma(f, q, row.x);
}

enum Prior : int {
    PR_FLAT = 0, PR_NORMAL = 1, PR_HALFCAUCHY = 2, PR_UNIFORM = 3, PR_BETA = 4, PR_NORMAL_REF = 5, PR_GAMMA = 6,
    PR_EXPONENTIAL = 7, PR_LOGNORMAL = 8, PR_CAUCHY = 9
};

// x-independent part of a scalar's log-prior, computed on the host (keeps lgamma / atan out of the kernels)
inline double prior_const(int kind, double a, double b) {
    switch (kind) {
        case PR_NORMAL:
            return -0.5 * kLog2Pi - std::log(b);
        case PR_HALFCAUCHY:
            return -kLogPi - std::log(b) - std::log(1.0 - (std::atan((0.0 - a) / b) / kPi + 0.5));
        case PR_UNIFORM:
            return -std::log(b - a);
        case PR_BETA:
            return -(std::lgamma(a) + std::lgamma(b) - std::lgamma(a + b));
        case PR_GAMMA:
            return -a * std::log(b) - std::lgamma(a);
        case PR_EXPONENTIAL:
            return -std::log(b);
        case PR_LOGNORMAL:
            return -std::log(b) - 0.5 * kLog2Pi;
        case PR_CAUCHY:
            return -kLogPi - std::log(b);
        default:
            return 0.0;
    }
}

// Phi(z) and phi(z)/S from the generated piecewise polynomial of Phi (tools/gen_phi_table.py: absolute error 1.1e-16 for Phi
// and for the phi recovered from it; |z| clamped to 8.5, beyond which they are 0 / 1 in double precision).  S = kPhiPerUnit
// rows per unit of z.  The argument arrives SCALED, zs = S z: row i = round(zs + 8.5 S) of the table is the polynomial of Phi
// around z_i = -8.5 + i/S in u = zs + 8.5 S - i in [-1/2, 1/2]; value and derivative (dPhi/du = phi/S) come out of one Horner
// recurrence (2 deg - 1 FMAs, deg + 1 coefficients), no exp.  Row index and u without a conversion instruction: adding
// 1.5 * 2^52 + 8.5 S rounds zs + 8.5 S to an integer that sits in the low word of the sum; subtracting the constant again
// gives i - 8.5 S exactly.  (v_cvt_i32_f64, v_fract_f64, v_mul_lo_u32 and v_ldexp_f64 run at a quarter of the FP64 add rate;
// the row address is a 24-bit multiply, which runs at full rate.  Rows are NOT padded to a power of two: with a 128-byte
// stride every row starts on LDS bank 0 and lanes that read different rows serialise -- measured 2.7x slower than the
// 80-byte stride.)  For sums of order one -- the LBA density and distribution function -- where absolute accuracy is what
// counts; the log-survival of the LNR has a table of its own (log_Phi_neg_table below: relative accuracy in the tail).  tab
// points at a copy of kPhiTable.  The clamp is v_max / v_min: a NaN argument is read as -8.5 (callers that must turn a NaN into -Inf test their
// own inputs, see lba_trial).
constexpr double kPhiS = (double)kPhiPerUnit;
constexpr double kPhiHalf = kPhiZmax * kPhiPerUnit;              // 8.5 S: an integer
constexpr double kPhiMagic = 6755399441055744.0 + kPhiHalf;      // 1.5 * 2^52 + 8.5 S (exactly representable)
static_assert(kPhiZmax == 8.5 && (kPhiPerUnit & 1) == 0 && kPhiRow >= kPhiDeg + 1 && (kPhiRow & 1) == 0, "table shape");
// RS = doubles from one row of the LDS copy to the next: kPhiRow (rows packed, any lane may read any copy) or 16 (rows of
// 128 bytes in the eight-copy layout of k_lba_loglike, where `tab` is the LANE'S OWN copy -- see that kernel).
template <int RS = kPhiRow>
__device__ __forceinline__ void phiS_Phi_table(const double* tab, double zs, double& phS, double& Ph) {
    const double zc = fmin(fmax(zs, -kPhiHalf), kPhiHalf);
    const double t = zc + kPhiMagic;              // = round(zc + 8.5 S) + 1.5 * 2^52
    const double u = zc - (t - kPhiMagic);        // t - magic = row - 8.5 S, exactly
    const int row = __double2loint(t);            // low word of the sum = the row
    const double* a = RS == 16 ? tab + (row << 4) : tab + __mul24(row, RS);
    double P = a[kPhiDeg], dP = a[kPhiDeg];
    P = fma(P, u, a[kPhiDeg - 1]);
#pragma unroll
    for (int k = kPhiDeg - 2; k >= 0; --k) {
        dP = fma(dP, u, P);
        P = fma(P, u, a[k]);
    }
    Ph = P;
    phS = dP;
}
// log Phi(-z), the log-survival of a standard normal, finite far into the tail,
// from its own table (tools/gen_logphi_table.py: one degree-9 polynomial per interval of 1/8 on [-8.5, 38.5], error
// 2.2e-16 relative to max(1, |g|)): nine FMAs and ten coefficients per value -- no erfcx, no exp, no log.  Row and u as for the
// Phi table (1.5 * 2^52 added and subtracted).  Below -8.5 the value is 0 to double precision (clamp); beyond 38.5 --
// survival probabilities under 1e-324 -- Mills' ratio: log Phi(-z) = -z^2/2 - log z - log sqrt(2 pi) + log(1 - 1/z^2 + 3/z^4
// - 15/z^6) (relative error < 1e-11 there), out of line.  tab points at a copy of kLogPhiTable.
__device__ __attribute__((noinline)) double log_Phi_neg_far(double z) {
    const double i2 = 1.0 / (z * z);
    return -0.5 * z * z - log(z) - 0.5 * kLog2Pi + log1p(i2 * (-1.0 + i2 * (3.0 - 15.0 * i2)));
}
static_assert(kLogPhiZlo == -8.5 && kLogPhiPerUnit == 8 && kLogPhiRow == 10 && kLogPhiDeg == 9, "log_Phi_neg_table is written for this table shape");
__device__ __forceinline__ double log_Phi_neg_table(const double* tab, double z) {
    if (z > kLogPhiZhi) return log_Phi_neg_far(z);  // (NaN falls through to the clamp and reads row 0)
    const double zc = fmin(fmax(8.0 * z, -68.0), 8.0 * kLogPhiZhi);
    const double kM = 6755399441055744.0 + 68.0;
    const double t = zc + kM;
    const double u = zc - (t - kM);
    const double* a = tab + __mul24(__double2loint(t), kLogPhiRow);
    double P = a[9];
#pragma unroll
    for (int k = 8; k >= 0; --k) P = fma(P, u, a[k]);
    return P;
}

// LBA (Examples/Run_LBA.jl:33-37; SequentialSamplingModels conventions: b = A + k, sigma = 1,
// normalised by 1 - P(all drifts <= 0), density floored at 1e-10).
// One accumulator at decision time t, n1 = (b - A - t v)/t, n2 = (b - t v)/t, sharing the four phi / Phi values:
//   density      f = (1/A) [ v (Phi(n2) - Phi(n1)) + (phi(n1) - phi(n2)) ]                       (the winner's factor)
//   survival 1-F = (t/A) [ (n2 Phi(n2) - n1 Phi(n1)) - (phi(n1) - phi(n2)) ]                      (a loser's factor)
// Only ONE of the two is needed per accumulator and trial, and which one is wave-uniform (the lanes of a wave are
// proposals at the same trial): `win` is a scalar branch.  Everything arrives scaled by S (the table's argument scale):
// vS = S v, c1 = S k/t, c2 = S b/t, so m1 = S n1, m2 = S n2; the table returns q = phi/S.  With dq = q1 - q2:
//   f = (1/A) [ v (P2 - P1) + S dq ],      1 - F = (t / S A) [ (m2 P2 - m1 P1) - S^2 dq ].
template <int RS = kPhiRow>
__device__ __forceinline__ double lba_factor(const double* tab, bool win, double v, double vS, double c1, double c2, double inv_A,
                                             double t_inv_SA) {
    const double m1 = c1 - vS, m2 = c2 - vS;
    double q1, P1, q2, P2;
    phiS_Phi_table<RS>(tab, m1, q1, P1);
    phiS_Phi_table<RS>(tab, m2, q2, P2);
    const double dq = q1 - q2;
    if (win) return inv_A * fma(kPhiS, dq, v * (P2 - P1));
    return t_inv_SA * fma(-(kPhiS * kPhiS), dq, fma(m2, P2, -(m1 * P1)));
}
// 1/x for x > 0 to the last bit or two: hardware reciprocal + two Newton steps (the IEEE division is ~25 instructions)
__device__ __forceinline__ double recip_pos(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
// the factor a trial contributes to the likelihood PRODUCT: max(density, 1e-10), or 0 where the reference's log-density is
// -Inf (decision time not after tau, NaN density) -- the caller takes ONE log of the product of several trials.
// nuS = S nu, kS = S k, bS = S b, inv_SA = 1 / (S A)
// (C = int: the winner as a wave-uniform index, k_obs_loglike; C = double: as it lies in the data, one per lane -- k_lba_wave --
// compared without a conversion instruction)
template <int NA, int RS = kPhiRow, typename C = int>
__device__ __forceinline__ double lba_trial(const double* tab, int na_rt, const double* nu, const double* nuS, double kS, double bS,
                                            double tau, double inv_A, double inv_SA, double inv_norm, C c, double rt) {
    const int na = NA > 0 ? NA : na_rt;
    const double t = rt - tau;
    const double inv_t = recip_pos(t), c1 = kS * inv_t, c2 = bS * inv_t, t_inv_SA = t * inv_SA;
    double den = inv_norm;
#pragma unroll
    for (int a = 0; a < (NA > 0 ? NA : 8); ++a)
        if (a < na) den *= lba_factor<RS>(tab, (C)(a + 1) == c, nu[a], nuS[a], c1, c2, inv_A, t_inv_SA);
    const double floored = fmax(den, 1e-10);  // (a NaN density is caught below)
    // The table clamps its argument with v_max / v_min, which read a NaN as -8.5: a NaN ARGUMENT would no longer reach `den`.
    // The arguments can only become NaN through 1/t of a denormal t (rcp = Inf, 0 * Inf in the Newton steps); decision times
    // and tau are numbers of order one, whose difference is 0 or at least 1e-17, so `t > 1e-300` costs nothing, changes no
    // reachable case and keeps every argument a number.
    return (t > 1e-300 && den == den) ? floored : 0.0;
}

}  // namespace demc
