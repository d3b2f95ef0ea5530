/* A plain-C host that drives SEVERAL GPUs from one process through the C ABI (include/fcamd_multi.h, "one process, several
 * GPUs"): the single assembler of BASELINE.json's north_star.  VonMises3D on N points in this process's own arrays;
 * every device context evaluates its slice in place (fcamd_multi_evaluate_host) -- compared bit for bit with the
 * single-device entry -- and then a two-increment Newton-style loop on the device-resident state (fcamd_multi_state_*).
 *
 *   gcc -std=c99 -I include examples/c_caller_multi.c -o c_caller_multi -L fenics-constitutive_amd/lib -lfcamd -lm
 *   ./c_caller_multi [n_contexts]      (contexts are dealt round-robin to the visible GPUs: 2 contexts on a 1-GPU box share it)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fcamd_multi.h"

#define CHECK(call)                                                                 \
    do {                                                                            \
        int rc_ = (call);                                                           \
        if (rc_ != FCAMD_OK) {                                                      \
            fprintf(stderr, "%s -> %s: %s\n", #call, fcamd_status_string(rc_), fcamd_last_error()); \
            return 1;                                                               \
        }                                                                           \
    } while (0)

static double frand(unsigned* s) { /* xorshift: the same numbers everywhere */
    *s ^= *s << 13, *s ^= *s >> 17, *s ^= *s << 5;
    return (double)(*s % 2000001u) / 1000000.0 - 1.0;
}

int main(int argc, char** argv) {
    enum { N = 100003 }; /* ragged: the last slice ends inside a 64-point tile */
    const int n_ctx = argc > 1 ? atoi(argv[1]) : 2;
    const double params[5] = {175000.0, 80769.0, 1200.0, 2500.0, 200.0}; /* p_ka, p_mu, p_y0, p_y00, p_w (tests/models/test_plasticity.py:19-25) */
    int devices[FCAMD_MULTI_MAX_DEVICES];
    if (n_ctx < 1 || n_ctx > FCAMD_MULTI_MAX_DEVICES) return 9;
    /* the ordinals repeat when there are fewer GPUs than contexts: slot k -> device k mod (visible GPUs) is the caller's choice */
    for (int k = 0; k < n_ctx; ++k) devices[k] = 0;

    double *grad = malloc(9 * N * sizeof(double)), *s1 = calloc(6 * N, sizeof(double)), *s2 = calloc(6 * N, sizeof(double));
    double *t1 = malloc(36 * N * sizeof(double)), *t2 = malloc(36 * N * sizeof(double));
    double *e1 = calloc(6 * N, sizeof(double)), *e2 = calloc(6 * N, sizeof(double)), *a1 = calloc(N, sizeof(double)), *a2 = calloc(N, sizeof(double));
    unsigned seed = 12345u;
    for (int i = 0; i < N; ++i) {
        const double scale = (i % 3 == 0) ? 1e-2 : 1e-4; /* a third of the points yields */
        for (int j = 0; j < 9; ++j) grad[9 * i + j] = scale * frand(&seed);
    }

    /* (1) single device: the entry a one-GPU host binds */
    fcamd_context* ctx = NULL;
    fcamd_model* law = NULL;
    fcamd_stats st1, st2;
    CHECK(fcamd_context_create(0, NULL, &ctx));
    CHECK(fcamd_model_create(ctx, FCAMD_VON_MISES_3D, FCAMD_FULL, params, 5, &law));
    double* h1[2] = {e1, a1};
    CHECK(fcamd_evaluate_host(law, 0.0, 1.0, N, grad, s1, t1, h1, 2, &st1));

    /* (2) the same call spread over n_ctx device contexts of this process */
    fcamd_multi* mg = NULL;
    CHECK(fcamd_multi_create(devices, n_ctx, FCAMD_VON_MISES_3D, FCAMD_FULL, params, 5, &mg));
    int used = 0, mode = 0;
    CHECK(fcamd_multi_plan(mg, N, 0, &used, NULL, NULL));
    double* h2[2] = {e2, a2};
    CHECK(fcamd_multi_evaluate_host(mg, 0.0, 1.0, N, grad, s2, t2, h2, 2, &st2));
    CHECK(fcamd_multi_last_host_mode(mg, &mode, &used));
    int64_t lo = 0, hi = 0, covered = 0;
    for (int k = 0; k < n_ctx; ++k) {
        CHECK(fcamd_multi_bounds(mg, N, k, &lo, &hi));
        covered += hi - lo;
    }
    if (covered != N) return 4;
    if (memcmp(s1, s2, 6 * N * sizeof(double)) || memcmp(t1, t2, 36 * N * sizeof(double)) || memcmp(e1, e2, 6 * N * sizeof(double)) ||
        memcmp(a1, a2, N * sizeof(double)) || st1.n_plastic != st2.n_plastic || st1.n_newton_iters != st2.n_newton_iters) {
        fprintf(stderr, "multi-device result differs from the single-device result\n");
        return 2;
    }

    /* (3) device-resident state: two increments of two Newton iterations; only grad goes up, stress / tangent come down */
    fcamd_multi_state* state = NULL;
    CHECK(fcamd_multi_state_create(mg, N, 0, &state));
    CHECK(fcamd_multi_state_set(state, NULL, NULL, 0)); /* zero initial state */
    for (int inc = 0; inc < 2; ++inc) {
        for (int it = 0; it < 2; ++it) {
            const int flags = (inc + it > 0) ? FCAMD_EVAL_SPARSE_TANGENT : 0; /* t2 holds the previous call's tangent from the second call on */
            CHECK(fcamd_multi_state_evaluate(state, (double)inc, 1.0, grad, s2, t2, flags, &st2));
        }
        CHECK(fcamd_multi_state_commit(state));
    }
    /* the first increment from the zero state is the in-place call above: committed history after increment 1 == e1 / a1 ... */
    memset(s2, 0, 6 * N * sizeof(double));
    CHECK(fcamd_multi_state_get(state, 0, s2, h2, 2));
    /* ... after increment 2 the plastic strain has grown further */
    double grew = 0.0;
    for (int i = 0; i < N; ++i) grew = fmax(grew, a2[i] - a1[i]);
    printf("fcamd v%d: %d points on %d device contexts (%d used, host mode flags %d): multi == single bit for bit, "
           "%llu plastic points; resident state: alpha grew by up to %.3e in increment 2\n",
           fcamd_version(), N, n_ctx, used, mode, (unsigned long long)st1.n_plastic, grew);
    CHECK(fcamd_multi_state_destroy(state));
    CHECK(fcamd_multi_destroy(mg));
    CHECK(fcamd_model_destroy(law));
    CHECK(fcamd_context_destroy(ctx));
    free(grad), free(s1), free(s2), free(t1), free(t2), free(e1), free(e2), free(a1), free(a2);
    return (st1.n_plastic > 0 && grew > 0.0) ? 0 : 3;
}
