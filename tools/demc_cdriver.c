/* demc_cdriver.c -- plain-C caller of the C-ABI (include/demc.h), the way a non-Python host (the Julia @ccall shim of
 * julia/DEMCHIP.jl, or any FFI) drives the library: Examples/Gaussian_Example.jl end to end.
 * Build: gcc -O2 -I include tools/demc_cdriver.c -o tools/demc_cdriver -L differentialevolutionmcmc.jl_amd -ldemc_hip -lm
 * Run  : LD_LIBRARY_PATH=differentialevolutionmcmc.jl_amd ./tools/demc_cdriver      (needs an MI355X) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "demc.h"

#define CK(call)                                                                 \
    do {                                                                         \
        int32_t rc_ = (call);                                                    \
        if (rc_ != DEMC_OK) {                                                    \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, demc_last_error(h));   \
            return 1;                                                            \
        }                                                                        \
    } while (0)

static double lcg(unsigned long long* s) { /* host-side prior draws only */
    *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
    return (double)(*s >> 11) / 9007199254740992.0;
}

int main(void) {
    enum { N = 50, G = 4, NP = 6, P = G * NP, D = 2, N_ITER = 3000, BURN = 1500 };
    unsigned long long s = 50514;
    double data[N];
    for (int i = 0; i < N; ++i) data[i] = sqrt(-2.0 * log(1.0 - lcg(&s))) * cos(6.283185307179586 * lcg(&s));
    demc_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.n_groups = G; cfg.Np = NP; cfg.D = D; cfg.burnin = BURN; cfg.n_rows = N_ITER;
    cfg.alpha = 0.1; cfg.beta = 0.1; cfg.eps = 0.001; cfg.sigma = 0.05; cfg.kappa = 1.0; cfg.theta_snooker = 0.0;
    cfg.schedule = DEMC_SCHED_TWO_COLOUR; cfg.store_history = 1; cfg.n_groups_total = G; cfg.seed = 20261003;
    demc_handle* h = NULL;
    CK(demc_create(&cfg, &h));
    const int64_t dims[1] = {N};
    CK(demc_set_model(h, DEMC_FAM_GAUSSIAN, data, dims, 1, NULL, 0));
    const int32_t kind[D] = {DEMC_PRIOR_NORMAL, DEMC_PRIOR_HALFCAUCHY};
    const double a[D] = {0.0, 0.0}, b[D] = {1.0, 1.0};
    const int32_t ref[D] = {0, 0};
    CK(demc_set_priors(h, kind, a, b, ref));
    const double lo[D] = {-INFINITY, 0.0}, hi[D] = {INFINITY, INFINITY};
    CK(demc_set_bounds(h, lo, hi));
    double theta[P * D];
    for (int p = 0; p < P; ++p) {
        theta[p * D] = 2.0 * lcg(&s) - 1.0;
        theta[p * D + 1] = 0.2 + 2.0 * lcg(&s);
    }
    CK(demc_set_state(h, theta, NULL, NULL)); /* weights evaluated on device */
    CK(demc_step(h, 1, N_ITER));
    const int keep = N_ITER - BURN;
    double* hist = (double*)malloc(sizeof(double) * (size_t)keep * P * D);
    uint8_t* acc = (uint8_t*)malloc((size_t)keep * P);
    CK(demc_get_history(h, BURN, N_ITER, hist, acc, NULL, NULL));
    double m[D] = {0, 0}, xbar = 0, ar = 0;
    for (int i = 0; i < N; ++i) xbar += data[i] / N;
    for (long long i = 0; i < (long long)keep * P; ++i) {
        m[0] += hist[i * D];
        m[1] += hist[i * D + 1];
        ar += acc[i];
    }
    m[0] /= (double)keep * P; m[1] /= (double)keep * P; ar /= (double)keep * P;
    printf("demc version %d: posterior mean mu=%.4f (data mean %.4f) sigma=%.4f accept=%.3f\n", demc_version(), m[0], xbar, m[1], ar);
    const int ok = fabs(m[0] - xbar) < 0.1 && m[1] > 0.6 && m[1] < 1.6 && ar > 0.1;
    free(hist); free(acc);
    demc_destroy(h);
    puts(ok ? "C-ABI driver OK" : "C-ABI driver FAILED");
    return ok ? 0 : 2;
}
