// How fast does ONE wave per SIMD issue v_mfma_f64_16x16x4_f64, as a function of how its accumulation chains are
// interleaved?  (k_res_mvn's observation stage: one wave per SIMD, two particle tiles = two chains.)  The timed loop is ONE
// asm statement (eight MFMAs + the loop counter), so the accumulators provably stay in AGPRs across the back edge -- with
// "+a" operands on separate statements the compiler kept the loop-carried values in VGPRs and copied them in and out.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_chain.hip -o /tmp/mfma_chain && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define M(acc) "v_mfma_f64_16x16x4_f64 %" #acc ", %5, %6, %" #acc "\n\t"
#define N1 "s_nop 1\n\t"
#define LOOP(body) "1:\n\t" body "s_sub_u32 %4, %4, 1\n\ts_cmp_lg_u32 %4, 0\n\ts_cbranch_scc1 1b"
template <int PAT, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k(double* out, long long* cyc, int iters, double a0, double b0) {
    d4 c0 = {0.0, 0.0, 0.0, 0.0}, c1 = c0, c2 = c0, c3 = c0;
    double a = a0 + threadIdx.x, b = b0;
    int n = iters;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
#define RUNPAT(body) asm volatile(LOOP(body) : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3), "+s"(n) : "v"(a), "v"(b) : "scc")
    if (PAT == 0) RUNPAT(M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0));                          // one chain
    if (PAT == 1) RUNPAT(M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1));                          // two chains, alternating
    if (PAT == 2) RUNPAT(M(0) M(0) M(1) M(1) M(0) M(0) M(1) M(1));                          // two chains, runs of two
    if (PAT == 3) RUNPAT(M(0) M(0) M(0) M(0) M(1) M(1) M(1) M(1));                          // two chains, runs of four
    if (PAT == 4) RUNPAT(M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3));                          // four chains, alternating
    if (PAT == 5) RUNPAT(M(0) M(0) M(1) M(1) M(2) M(2) M(3) M(3));                          // four chains, runs of two
    if (PAT == 6) RUNPAT(N1 M(0) N1 M(1) N1 M(0) N1 M(1) N1 M(0) N1 M(1) N1 M(0) N1 M(1));  // two chains alternating, s_nop 1 each
    // what else may sit between the MFMAs of a tile loop (the loaded values are not used: issue cost only)
    __shared__ double pad[2048];
    pad[threadIdx.x] = a0;
    unsigned la = threadIdx.x * 8;
    double l0, l1, l2, l3;
    double __attribute__((ext_vector_type(2))) l4;
#define RUNLDS(body) asm volatile(LOOP(body) "\n\ts_waitcnt lgkmcnt(0)" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3), "+s"(n), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3), "=&v"(l4), "+v"(la) : "v"(a), "v"(b) : "scc", "memory")
#undef M
#define M(acc) "v_mfma_f64_16x16x4_f64 %" #acc ", %11, %12, %" #acc "\n\t"
    if (PAT == 7) RUNLDS(M(0) M(1) "ds_read_b64 %5, %10\n\t" M(0) M(1) "ds_read_b64 %6, %10 offset:512\n\t" M(0) M(1) "ds_read_b64 %7, %10 offset:1024\n\t" M(0) M(1) "ds_read_b64 %8, %10 offset:1536\n\t");
    if (PAT == 8) RUNLDS(M(0) M(1) M(0) M(1) "ds_read2st64_b64 %9, %10 offset1:1\n\t" M(0) M(1) M(0) M(1));
    if (PAT == 9) RUNLDS(M(0) M(1) M(0) M(1) "v_add_u32 %10, 0, %10\n\t" M(0) M(1) M(0) M(1));
    if (PAT == 10) RUNLDS(M(0) M(1) "v_add_u32 %10, 0, %10\n\t" M(0) M(1) "v_add_u32 %10, 0, %10\n\t" M(0) M(1) "v_add_u32 %10, 0, %10\n\t" M(0) M(1) "v_add_u32 %10, 0, %10\n\t");
    if (PAT == 11) RUNLDS(M(0) M(1) M(0) M(1) "s_waitcnt lgkmcnt(0)\n\t" M(0) M(1) M(0) M(1) "s_waitcnt lgkmcnt(0)\n\t");
    if (PAT == 12) RUNLDS(M(0) M(1) M(0) M(1) "ds_read_b64 %5, %10\n\tds_read_b64 %6, %10 offset:512\n\ts_waitcnt lgkmcnt(2)\n\t" M(0) M(1) M(0) M(1) "ds_read_b64 %7, %10 offset:1024\n\tds_read_b64 %8, %10 offset:1536\n\ts_waitcnt lgkmcnt(2)\n\t");
    asm volatile("s_nop 15\n\ts_nop 7" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3));
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + (PAT >= 7 ? l0 + l1 + l2 + l3 + l4[0] + pad[threadIdx.x ^ 1] + la : 0.0);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int PAT, int WAVES>
void run(const char* name, int blocks) {
    double* out; long long* cyc;
    (void)hipMalloc(&out, sizeof(double) * blocks * 64 * WAVES);
    (void)hipMalloc(&cyc, sizeof(long long) * blocks);
    const int iters = 1000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<PAT, WAVES><<<blocks, 64 * WAVES>>>(out, cyc, iters, 1.0, 2.0);
    (void)hipEventRecord(e0);
    k<PAT, WAVES><<<blocks, 64 * WAVES>>>(out, cyc, iters, 1.0, 2.0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long c; (void)hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    printf("%-52s %d waves/SIMD: %6.1f s_memtime ticks per MFMA per wave, %5.1f ns (events)\n", name, WAVES / 4, (double)c / (iters * 8.0), ms * 1e6 / (iters * 8.0));
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    const int blocks = 256;
    run<0, 4>("one chain", blocks);
    run<1, 4>("two chains, alternating", blocks);
    run<2, 4>("two chains, runs of two", blocks);
    run<3, 4>("two chains, runs of four", blocks);
    run<4, 4>("four chains, alternating", blocks);
    run<5, 4>("four chains, runs of two", blocks);
    run<6, 4>("two chains alternating, s_nop 1 before each", blocks);
    run<7, 4>("two chains + 4 ds_read_b64 per 8 MFMAs", blocks);
    run<8, 4>("two chains + 1 ds_read2st64_b64 per 8 MFMAs", blocks);
    run<9, 4>("two chains + 1 v_add_u32 per 8 MFMAs", blocks);
    run<10, 4>("two chains + 4 v_add_u32 per 8 MFMAs", blocks);
    run<11, 4>("two chains + 2 s_waitcnt per 8 MFMAs", blocks);
    run<12, 4>("the tile loop's shape: 2 x (2 ds_read_b64 + wait(2))", blocks);
    run<0, 8>("one chain", blocks);
    run<1, 8>("two chains, alternating", blocks);
    run<4, 8>("four chains, alternating", blocks);
    return 0;
}
