// longrow_traffic_probe.hip -- what HBM rate can a kernel with k_longrow's TRAFFIC SHAPE and OCCUPANCY reach on this chip?
//
// k_longrow (demc_longrow.hpp; BASELINE cfg4: 128 groups x 32 particles, D = 10 002 doubles = an 80 KB row) moves, per
// particle and block sweep: its own row in (HBM), two partner rows of its group in (L2 / Infinity Cache when the group's
// particles run in step on one XCD), theta' parked in LDS until the Metropolis decision, then the accepted row out and --
// once per iteration -- one history row out.  VERDICT r4 #2a: the 0.29 of HBM that kernel reaches should ALSO be quoted
// against the rate a bare kernel of this shape reaches, because a streaming copy's 6.3 TB/s is not the roof of a kernel
// that takes 80 KB rows one workgroup at a time at 8 waves per CU.
//
// This probe does the row traffic and (optionally) a stand-in for the arithmetic, nothing else: no RNG, no prior, no
// likelihood.  Forms (all persistent, particle -> workgroup -> XCD mapping as launch_phase / k_longrow):
//   WG 256 (two workgroups per CU) or 512 (one), theta' parked in LDS or not (not: the stores re-form theta' from re-read rows),
//   subject-sweep shape (own + 2 partner rows in; accepted row + history row out) or hyper-sweep shape (own row in, nothing out),
//   `work` dependent FP64 FMAs per scalar (0: pure traffic; ~126: the span loops' instruction count, profiles/r04/NOTES.md),
//   `ahead` blocks requested before the one being worked on (k_longrow: 1).
// Build / run (GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/longrow_traffic_probe.hip -o /tmp/lrprobe && /tmp/lrprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                              \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                          \
        }                                                                                     \
    } while (0)

struct Args {
    const double* theta;  // [G][Np][D]
    double* theta_out;    // the same buffer (accepted rows are written in place, as the kernel does)
    double* hist;         // [P][D] one history row of the iteration
    double* sums;         // [P] per-particle result (keeps the loads alive)
    int G, Np, D, a_lo, n_act;  // moving particles of a group: a_lo .. a_lo + n_act - 1 (a colour of two_colour)
    int work;                   // dependent FMAs per scalar
    int accept_per_256;         // accepted fraction x 256
    int subject;                // 1: subject-sweep shape, 0: hyper-sweep shape
    unsigned seed;
};

__device__ inline unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__device__ inline double grind(double v, int work) {  // `work` dependent FMAs (one chain per scalar; four chains per lane and block)
    for (int i = 0; i < work; ++i) v = __builtin_fma(v, 0.999999, 1e-9);
    return v;
}

template <int WG, bool PARK, int AHEAD>
__global__ __launch_bounds__(WG, 512 / WG) void probe(Args p) {
    extern __shared__ double lds[];  // theta' [D] when parked
    __shared__ double s_red[WG / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = p.D, Np = p.Np;
    const int n_prop = p.G * p.n_act;
    const int n_blocks = (D + 3) >> 2;             // a lane's unit: four consecutive scalars (two 16-byte loads per row)
    const int rounds = (n_blocks + WG - 1) / WG;
    for (int vb = blockIdx.x; vb < n_prop; vb += gridDim.x) {
        int g, qg;
        if ((p.G & 7) == 0) {  // the particles of a group share an XCD (blocks b and b + 8 do)
            const int xcd = vb & 7, j = vb >> 3;
            qg = j % p.n_act;
            g = (j / p.n_act) * 8 + xcd;
        } else {
            g = vb / p.n_act;
            qg = vb % p.n_act;
        }
        const int pl = p.a_lo + qg;
        const size_t slot = (size_t)g * Np + pl;
        const double* grows = p.theta + (size_t)g * Np * D;
        const double* pt = grows + (size_t)pl * D;
        const unsigned h = hash32(p.seed ^ (unsigned)slot * 2654435761u);
        // partners: two distinct rows of the group other than the particle's own (pick_pair's shape)
        // (of the colour that does not move in this launch, as two_colour's pool: rows nobody writes meanwhile)
        const unsigned pool_n = (unsigned)(Np - p.n_act), pool_lo = p.a_lo == 0 ? (unsigned)p.n_act : 0u;
        unsigned a = h % pool_n, b = (h >> 12) % (pool_n - 1);
        if (b >= a) ++b;
        a += pool_lo; b += pool_lo;
        const double* Pa = p.subject ? grows + (size_t)a * D : pt;
        const double* Pb = p.subject ? grows + (size_t)b * D : pt;
        const bool accept = p.subject && (int)((h >> 24) & 255u) < p.accept_per_256;

        auto ld = [&](const double* row, int m, int half) -> double2 {  // scalars 4m + 2 half, + 1 (D is even: a pair is in or out)
            const int j = 4 * m + 2 * half;
            return j < D ? *reinterpret_cast<const double2*>(row + j) : make_double2(0.0, 0.0);
        };
        struct Blk { double2 t0, t1, a0, a1, b0, b1; };
        auto fetch = [&](int r) -> Blk {
            Blk k;
            const int m = tid + r * WG;
            k.t0 = ld(pt, m, 0); k.t1 = ld(pt, m, 1);
            if (p.subject) {
                k.a0 = ld(Pa, m, 0); k.a1 = ld(Pa, m, 1);
                k.b0 = ld(Pb, m, 0); k.b1 = ld(Pb, m, 1);
            } else {
                k.a0 = k.t0; k.a1 = k.t1; k.b0 = k.t0; k.b1 = k.t1;
            }
            return k;
        };
        double acc = 0.0;
        Blk q[AHEAD + 1];
#pragma unroll
        for (int i = 0; i < AHEAD; ++i) q[i] = fetch(i < rounds ? i : rounds - 1);
        for (int r = 0; r < rounds; ++r) {
            const int rn = r + AHEAD < rounds ? r + AHEAD : rounds - 1;  // (the tail re-requests the last block: static loop shape)
            q[AHEAD] = fetch(rn);
            const Blk k = q[0];
            const int m = tid + r * WG;
            double v[4];
            v[0] = grind(k.t0.x + 0.7 * (k.a0.x - k.b0.x), p.work);
            v[1] = grind(k.t0.y + 0.7 * (k.a0.y - k.b0.y), p.work);
            v[2] = grind(k.t1.x + 0.7 * (k.a1.x - k.b1.x), p.work);
            v[3] = grind(k.t1.y + 0.7 * (k.a1.y - k.b1.y), p.work);
            acc += (v[0] + v[1]) + (v[2] + v[3]);
            if (PARK && p.subject) {
                if (4 * m < D) *reinterpret_cast<double2*>(lds + 4 * m) = make_double2(v[0], v[1]);
                if (4 * m + 2 < D) *reinterpret_cast<double2*>(lds + 4 * m + 2) = make_double2(v[2], v[3]);
            }
#pragma unroll
            for (int i = 0; i < AHEAD; ++i) q[i] = q[i + 1];
        }
        // the decision: a workgroup reduction, as the kernel's (one barrier pair)
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) s_red[wave] = acc;
        __syncthreads();
        double tot = 0.0;
        for (int w = 0; w < WG / 64; ++w) tot += s_red[w];
        if (tid == 0) p.sums[slot] = tot;
        if (p.subject) {
            // row moves: the accepted theta' (or the current row) to the history row; an accepted theta' to the state row
            double* hrow = p.hist + slot * (size_t)D;
            double* trow = p.theta_out + slot * (size_t)D;
            for (int r = 0; r < rounds; ++r) {
                const int m = tid + r * WG;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int j = 4 * m + 2 * half;
                    if (j >= D) continue;
                    double2 v;
                    if (!accept)
                        v = *reinterpret_cast<const double2*>(pt + j);
                    else if (PARK)
                        v = *reinterpret_cast<const double2*>(lds + j);
                    else {  // theta' formed again from the rows (L2 by now)
                        const double2 t = *reinterpret_cast<const double2*>(pt + j), x = *reinterpret_cast<const double2*>(Pa + j),
                                      y = *reinterpret_cast<const double2*>(Pb + j);
                        v = make_double2(grind(t.x + 0.7 * (x.x - y.x), p.work), grind(t.y + 0.7 * (x.y - y.y), p.work));
                    }
                    *reinterpret_cast<double2*>(hrow + j) = v;
                    if (accept) *reinterpret_cast<double2*>(trow + j) = v;
                }
            }
        }
        __syncthreads();  // (s_red and the LDS row are reused by the next particle)
    }
}

template <int WG, bool PARK, int AHEAD>
static void run(const char* label, Args a, int n_cus, int reps) {
    const size_t lds = PARK ? sizeof(double) * (size_t)((a.D + 3) & ~3) : 0;
    CHECK(hipFuncSetAttribute((const void*)probe<WG, PARK, AHEAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, probe<WG, PARK, AHEAD>, WG, lds));
    if (per_cu > 512 / WG * 2) per_cu = 512 / WG * 2;  // (at most 16 waves per CU: what a kernel with ~128 VGPRs could hold)
    const int n_prop = a.G * a.n_act;
    int grid = per_cu * n_cus;
    if (grid > n_prop) grid = n_prop;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 4; ++i) {  // warm-up: both colours
        a.a_lo = (i & 1) * a.n_act;
        probe<WG, PARK, AHEAD><<<grid, WG, lds>>>(a);
    }
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) {
        a.a_lo = (i & 1) * a.n_act;
        a.seed += 17;
        probe<WG, PARK, AHEAD><<<grid, WG, lds>>>(a);
    }
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, row = 8.0 * a.D;
    const double acc = a.accept_per_256 / 256.0;
    // bytes that must cross the HBM interface per launch: own row in; history row and accepted row out
    const double hbm = n_prop * row * (a.subject ? 1.0 + 1.0 + acc : 1.0);
    const double l2 = a.subject ? n_prop * row * 2.0 : 0.0;  // partner rows (another particle's own row: L2 / Infinity Cache / HBM)
    printf("%-44s WG %3d x %d/CU grid %4d  work %3d ahead %d | %7.1f us/launch | necessary %6.1f MB -> %5.2f TB/s (%.3f of 8) | with partner rows %6.1f MB -> %5.2f TB/s\n",
           label, WG, per_cu, grid, a.work, AHEAD, us, hbm / 1e6, hbm / us / 1e6, hbm / us / 1e6 / 8.0, (hbm + l2) / 1e6, (hbm + l2) / us / 1e6);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
    int G = 128, Np = 32, D = 10002, reps = 40;
    if (argc > 1) G = atoi(argv[1]);
    if (argc > 2) D = atoi(argv[2]);
    if (argc > 3) reps = atoi(argv[3]);
    if (D % 2 || D < 8 || Np < 4 || G < 1) {
        fprintf(stderr, "need an even D >= 8\n");
        return 1;
    }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cus = prop.multiProcessorCount;
    const size_t P = (size_t)G * Np, n = P * (size_t)D;
    double *theta, *hist, *sums;
    CHECK(hipMalloc(&theta, n * sizeof(double)));
    CHECK(hipMalloc(&hist, n * sizeof(double)));
    CHECK(hipMalloc(&sums, P * sizeof(double)));
    {
        std::vector<double> h(n);
        for (size_t i = 0; i < n; ++i) h[i] = 1e-3 * (double)(i % 1009);
        CHECK(hipMemcpy(theta, h.data(), n * sizeof(double), hipMemcpyHostToDevice));
        CHECK(hipMemset(hist, 0, n * sizeof(double)));
    }
    printf("longrow traffic probe: %s, %d CUs; G %d x Np %d, D %d (row %.1f KB, population %.1f MB), %d moving particles per launch\n", prop.name, n_cus, G, Np,
           D, 8.0 * D / 1e3, n * 8.0 / 1e6, G * (Np / 2));
    Args a{theta, theta, hist, sums, G, Np, D, 0, Np / 2, 0, 64, 1, 12345u};
    // a plain device-to-device copy of the population, for the box's streaming rate
    {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        CHECK(hipMemcpyAsync(hist, theta, n * sizeof(double), hipMemcpyDeviceToDevice));
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) CHECK(hipMemcpyAsync(hist, theta, n * sizeof(double), hipMemcpyDeviceToDevice));
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("hipMemcpy D2D of the population: %.2f TB/s (read + write)\n", 2.0 * n * 8.0 * 10 / (ms * 1e-3) / 1e12);
    }
    for (int subject = 1; subject >= 0; --subject) {
        a.subject = subject;
        for (int work : {0, 32, 126}) {
            a.work = work;
            const char* s = subject ? "subject sweep (3 rows in, 1.25 out)" : "hyper sweep (own row in)";
            run<256, true, 1>(s, a, n_cus, reps);
            run<512, true, 1>(s, a, n_cus, reps);
            if (work == 0) {
                run<256, true, 2>(s, a, n_cus, reps);
                run<256, true, 3>(s, a, n_cus, reps);
                run<512, true, 3>(s, a, n_cus, reps);
            }
            if (subject) {
                run<256, false, 1>("  theta' not parked (re-formed for the stores)", a, n_cus, reps);
                if (work == 0) run<256, false, 3>("  theta' not parked (re-formed for the stores)", a, n_cus, reps);
            }
        }
    }
    CHECK(hipFree(theta));
    CHECK(hipFree(hist));
    CHECK(hipFree(sums));
    return 0;
}
