// fp64_pipes.hip -- microbenchmark: do the FP64 vector pipe (v_fma_f64) and the FP64 matrix pipe
// (v_mfma_f64_16x16x4_f64) of gfx950 run concurrently, and what does each sustain alone?
// Build: hipcc --offload-arch=gfx950 -O3 tools/fp64_pipes.hip -o tools/fp64_pipes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

// mode bit0: waves with (wave&1)==0 do MFMA; bit1: waves with (wave&1)==1 do VALU; mode 4: all MFMA; 8: all VALU
__global__ __launch_bounds__(256) void k(double* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = (mode == 4) || ((mode & 1) && (wave & 1) == 0);
    const bool do_valu = (mode == 8) || ((mode & 2) && (wave & 1) == 1);
    double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
    if (do_mfma) {
        d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c3, 0, 0, 0);
            }
        }
        out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (do_valu) {
        double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                x0 = fma(x0, b, a); x1 = fma(x1, b, a); x2 = fma(x2, b, a); x3 = fma(x3, b, a);
                x4 = fma(x4, b, a); x5 = fma(x5, b, a); x6 = fma(x6, b, a); x7 = fma(x7, b, a);
            }
        }
        out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
}

int main() {
    double* out;
    hipMalloc(&out, 4096 * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 40000;
    for (int blocks_per_cu : {2, 4}) {
        const int grid = 256 * blocks_per_cu;
        for (int mode : {4, 8, 3, 1, 2}) {
            k<<<grid, 256>>>(out, 10, mode);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            k<<<grid, 256>>>(out, iters, mode);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double waves = grid * 4.0;
            double mfma_waves = mode == 4 ? waves : (mode & 1) ? waves / 2 : 0;
            double valu_waves = mode == 8 ? waves : (mode & 2) ? waves / 2 : 0;
            const double f_mfma = mfma_waves * iters * 32.0 * (16 * 16 * 4 * 2);
            const double f_valu = valu_waves * iters * 256.0 * 64 * 2;
            printf("blocks/CU %d mode %d: %.3f ms  mfma %.1f TF  valu %.1f TF  total %.1f TF\n", blocks_per_cu, mode, ms,
                   f_mfma / ms / 1e9, f_valu / ms / 1e9, (f_mfma + f_valu) / ms / 1e9);
        }
    }
    return 0;
}
