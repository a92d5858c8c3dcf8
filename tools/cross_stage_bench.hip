// cross_tiles<KS = 2, MT = 2> (the observation stage of k_res_mvn<256,true,8>, BASELINE cfg2) in isolation: four waves,
// one per SIMD, 20 tiles each out of an LDS copy, as in the kernel -- cycles for the whole call and per MFMA.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I differentialevolutionmcmc.jl_amd/csrc tools/cross_stage_bench.hip -o /tmp/csb && /tmp/csb
#include "demc_kernels.hpp"
#include <cstdio>
using namespace demc;
template <int KS, int MT>
__global__ __launch_bounds__(256, 1) void kb(double* outg, long long* cyc, int tiles_per_wave, int reps) {
    extern __shared__ double lds[];
    double* ybuf = lds;                       // [64][4 KS]
    double* part = ybuf + 64 * 4 * KS;        // [4][64]
    double* xs = part + 4 * 64;               // [4 * tiles_per_wave + 1][KS][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = 4 * tiles_per_wave;
    for (int i = tid; i < 64 * 4 * KS; i += 256) ybuf[i] = 1.0 / (1 + i);
    for (int i = tid; i < (nt + 1) * KS * 64; i += 256) xs[i] = i < nt * KS * 64 ? 1.0 / (3 + i) : 0.0;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        cross_tiles<KS, MT, lds_cptr, true>((lds_cptr)ybuf, 4 * KS, 16 * MT, 0, (lds_cptr)xs, KS, wave * tiles_per_wave, (wave + 1) * tiles_per_wave, nt,
                                      (lds_ptr)(part + wave * 64), lane);
        __syncthreads();
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    outg[blockIdx.x * 256 + tid] = part[tid];
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KS, int MT>
void run(int tiles_per_wave) {
    double* out; long long* cyc;
    const int blocks = 256, reps = 100;
    (void)hipMalloc(&out, sizeof(double) * blocks * 256);
    (void)hipMalloc(&cyc, sizeof(long long) * blocks);
    const size_t lds = sizeof(double) * (64 * 4 * KS + 256 + (size_t)(4 * tiles_per_wave + 1) * KS * 64);
    (void)hipFuncSetAttribute((const void*)kb<KS, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    kb<KS, MT><<<blocks, 256, lds>>>(out, cyc, tiles_per_wave, reps);
    kb<KS, MT><<<blocks, 256, lds>>>(out, cyc, tiles_per_wave, reps);
    (void)hipDeviceSynchronize();
    long long c; (void)hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    const double per_call = (double)c / reps, mf = (double)tiles_per_wave * KS * MT;
    printf("KS %d MT %d, %2d tiles per wave: %7.0f cycles per call (+ barrier), %5.1f per MFMA, matrix work %5.0f\n", KS, MT, tiles_per_wave, per_call,
           per_call / mf, 64.0 * mf);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run<2, 2>(20);
    run<2, 2>(19);
    run<2, 2>(0);
    run<2, 4>(20);
    run<2, 1>(20);
    run<8, 2>(8);
    return 0;
}
