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
static double lcg(unsigned long long* s);

static double lcg(unsigned long long* s) { /* host-side prior draws only */
    *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
    return (double)(*s >> 11) / 9007199254740992.0;
}

/* The call sequence of julia/DEMCHIP.jl's `sample` method for a model with NESTED Theta and block updates whose
 * blocking_on(de) changes from iteration to iteration (src/main.jl:137,162): Theta = [mu::Vector{3}, sigma] (the shape of
 * test/multivariate_normal_tests.jl:19), blocks = [[mu-block], [sigma-block]], blocking on even iterations.
 *   create -> set_model -> set_priors -> set_bounds -> set_state(theta, weight = device-evaluated, ids)
 *   -> per run of equal blocking_on answers: set_blocks(masks | none) + demc_step(first, count)
 *   -> export_chains(layout 0 = Julia Array{Float64,3}(n, D+2, P)) -> un-flatten per top-level parameter. */
static int nested_blocked_sequence(void) {
    enum { N = 200, d = 3, D = d + 1, G = 4, NP = 8, P = G * NP, N_ITER = 600, BURN = 300 };
    unsigned long long s = 777;
    static double X[N * d];
    const double mu_true[d] = {0.5, -1.0, 2.0};
    for (int i = 0; i < N; ++i)
        for (int k = 0; k < d; ++k)
            X[i * d + k] = mu_true[k] + sqrt(-2.0 * log(1.0 - lcg(&s))) * cos(6.283185307179586 * lcg(&s));
    demc_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.n_groups = G; cfg.Np = NP; cfg.D = D; cfg.burnin = BURN; cfg.n_rows = N_ITER;
    cfg.alpha = 0.1; cfg.beta = 0.1; cfg.eps = 0.001; cfg.sigma = 0.05; cfg.kappa = 1.0;
    cfg.schedule = DEMC_SCHED_TWO_COLOUR; cfg.store_history = 1; cfg.n_groups_total = G; cfg.seed = 99;
    demc_handle* h = NULL;
    CK(demc_create(&cfg, &h));
    const int64_t dims[2] = {N, d};
    CK(demc_set_model(h, DEMC_FAM_MVN_ISO, X, dims, 2, NULL, 0));
    const int32_t kind[D] = {DEMC_PRIOR_NORMAL, DEMC_PRIOR_NORMAL, DEMC_PRIOR_NORMAL, DEMC_PRIOR_HALFCAUCHY};
    const double a[D] = {0, 0, 0, 0}, b[D] = {10, 10, 10, 1};
    const int32_t ref[D] = {0, 0, 0, 0};
    CK(demc_set_priors(h, kind, a, b, ref));
    /* de.bounds has one entry per TOP-LEVEL parameter: ((-Inf,Inf),(0,Inf)) -> per scalar */
    const double lo[D] = {-INFINITY, -INFINITY, -INFINITY, 0.0}, hi[D] = {INFINITY, INFINITY, INFINITY, INFINITY};
    CK(demc_set_bounds(h, lo, hi));
    /* de.blocks = [[[true,true,true], false], [[false,false,false], true]] flattened like Theta */
    const uint8_t masks[2 * D] = {1, 1, 1, 0, 0, 0, 0, 1};
    double theta[P * D];
    int64_t ids[P];
    for (int p = 0; p < P; ++p) {
        for (int k = 0; k < d; ++k) theta[p * D + k] = 4.0 * lcg(&s) - 2.0;
        theta[p * D + d] = 0.3 + 2.0 * lcg(&s);
        ids[p] = p; /* 0-based across the ABI (p.id - 1) */
    }
    CK(demc_set_state(h, theta, NULL, ids));
    int n_calls = 0;
    for (int first = 1; first <= N_ITER;) { /* blocking_runs(de, n_iter): blocking_on = iteration is even -> runs of one */
        const int on = (first % 2 == 0);
        int count = 1;
        while (first + count <= N_ITER && ((first + count) % 2 == 0) == on) ++count;
        CK(demc_set_blocks(h, masks, on ? 2 : 0));
        CK(demc_step(h, first, count));
        first += count;
        ++n_calls;
    }
    /* bundle_samples' gather on the device, Julia layout: v[row + n*(j + (D+2)*id)] */
    const size_t n = N_ITER;
    double* v = (double*)malloc(sizeof(double) * n * (D + 2) * P);
    CK(demc_export_chains(h, 0, N_ITER, 0, v));
    /* the same rows through the raw history + the id that sat in each slot */
    double* hist = (double*)malloc(sizeof(double) * n * P * D);
    uint8_t* acc = (uint8_t*)malloc(n * P);
    double* lp = (double*)malloc(sizeof(double) * n * P);
    int64_t* idh = (int64_t*)malloc(sizeof(int64_t) * n * P);
    CK(demc_get_history(h, 0, N_ITER, hist, acc, lp, idh));
    int bad = 0;
    for (size_t r = 0; r < n; ++r)
        for (int sl = 0; sl < P; ++sl) {
            const size_t id = (size_t)idh[r * P + sl];
            for (int j = 0; j < D; ++j) bad += v[r + n * (j + (size_t)(D + 2) * id)] != hist[(r * P + sl) * D + j];
            bad += v[r + n * (D + (size_t)(D + 2) * id)] != (double)acc[r * P + sl];
            bad += v[r + n * (D + 1 + (size_t)(D + 2) * id)] != lp[r * P + sl];
        }
    /* un-flatten per top-level parameter: mu = columns 0..2 (a Vector{Float64}), sigma = column 3 */
    double m[D] = {0, 0, 0, 0};
    for (size_t id = 0; id < P; ++id)
        for (size_t r = BURN; r < n; ++r)
            for (int j = 0; j < D; ++j) m[j] += v[r + n * (j + (size_t)(D + 2) * id)];
    int ok = (bad == 0) && n_calls == N_ITER;
    for (int j = 0; j < D; ++j) m[j] /= (double)(n - BURN) * P;
    double xbar[d] = {0, 0, 0};
    for (int i = 0; i < N; ++i)
        for (int k = 0; k < d; ++k) xbar[k] += X[i * d + k] / N;
    for (int k = 0; k < d; ++k) ok = ok && fabs(m[k] - xbar[k]) < 0.05;
    ok = ok && m[d] > 0.85 && m[d] < 1.15;
    printf("nested/blocked sequence: %d step calls, export mismatches %d, mu=(%.3f %.3f %.3f) vs data mean (%.3f %.3f %.3f), sigma=%.3f\n",
           n_calls, bad, m[0], m[1], m[2], xbar[0], xbar[1], xbar[2], m[d]);
    free(v); free(hist); free(acc); free(lp); free(idh);
    demc_destroy(h);
    return ok ? 0 : 3;
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
    if (!ok) {
        puts("C-ABI driver FAILED");
        return 2;
    }
    const int rc2 = nested_blocked_sequence();
    puts(rc2 == 0 ? "C-ABI driver OK" : "C-ABI driver FAILED (nested / blocked sequence)");
    return rc2;
}
