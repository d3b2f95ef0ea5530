/* A plain-C host runs the laws of one form() as ONE call (fcamd_evaluate_batch, include/fcamd.h): two materials on interleaved
 * cells of one mesh -- linear elasticity on the even cells, VonMises3D (still elastic at this strain) on the odd ones -- writing their
 * rows of the SHARED stress / tangent arrays through parent_rows, the way the reference's form() calls one LawOnSubMesh.evaluate per
 * material (solver/_solver.py:143-144, solver/maps.py:82-123).  Device memory, copies and the launch all go through the C ABI: no HIP
 * runtime in this file.  The second call -- same arrays, new gradient values: a Newton iteration -- replays the kept table.
 *
 *   gcc -std=c99 -I include examples/c_caller_batch.c -o c_caller_batch -L fenics-constitutive_amd/lib -lfcamd -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "fcamd.h"

#define CHECK(call)                                                                 \
    do {                                                                            \
        int rc_ = (call);                                                           \
        if (rc_ != FCAMD_OK) {                                                      \
            fprintf(stderr, "%s -> %s: %s\n", #call, fcamd_status_string(rc_), fcamd_last_error()); \
            return 1;                                                               \
        }                                                                           \
    } while (0)

enum { Q = 4, CELLS = 1501, N = Q * CELLS, N0 = Q * ((CELLS + 1) / 2), N1 = N - N0 }; /* ragged on purpose: N0, N1 are no multiples of 64 */

int main(void) {
    const double le_p[2] = {42.0, 0.3}, vm_p[5] = {175000.0, 80769.0, 1200.0, 2500.0, 200.0};
    fcamd_context* ctx = NULL;
    fcamd_model* laws[2] = {NULL, NULL};
    CHECK(fcamd_context_create(0, NULL, &ctx));
    CHECK(fcamd_model_create(ctx, FCAMD_LINEAR_ELASTICITY, FCAMD_FULL, le_p, 2, &laws[0]));
    CHECK(fcamd_model_create(ctx, FCAMD_VON_MISES_3D, FCAMD_FULL, vm_p, 5, &laws[1]));

    /* host side: the parent rows of each law (cells of Q points, dealt alternately) and the gradients */
    int32_t* rows[2] = {malloc(N0 * sizeof(int32_t)), malloc(N1 * sizeof(int32_t))};
    double* grad[2] = {calloc(9 * N0, sizeof(double)), calloc(9 * N1, sizeof(double))};
    int k0 = 0, k1 = 0;
    for (int c = 0; c < CELLS; ++c)
        for (int q = 0; q < Q; ++q)
            if (c % 2 == 0) rows[0][k0++] = Q * c + q; else rows[1][k1++] = Q * c + q;
    if (k0 != N0 || k1 != N1) return 3;

    /* device side, through the C ABI: the shared parent arrays, each law's gradient / rows / history (two copies) */
    const size_t nb[2] = {N0, N1};
    size_t bytes[16] = {6 * N * 8, 6 * N * 8, 36 * N * 8, 9 * N0 * 8, 9 * N1 * 8, N0 * 4, N1 * 4, 6 * N1 * 8, 6 * N1 * 8, N1 * 8, N1 * 8};
    void* d[11];
    CHECK(fcamd_device_alloc_set(ctx, 11, bytes, 0, FCAMD_ALLOC_SEQUENTIAL, d));
    double *stress_prev = d[0], *stress = d[1], *tangent = d[2], *g_dev[2] = {d[3], d[4]};
    double* zeros = calloc(6 * N, sizeof(double));
    CHECK(fcamd_copy_to_device(ctx, stress_prev, zeros, 6 * N * 8));
    for (int k = 7; k < 11; ++k) CHECK(fcamd_copy_to_device(ctx, d[k], zeros, bytes[k])); /* eps_n, alpha: committed and trial */
    for (int k = 0; k < 2; ++k) CHECK(fcamd_copy_to_device(ctx, d[5 + k], rows[k], nb[k] * 4));
    const double* hist_prev[2] = {d[7], d[9]};
    double* hist[2] = {d[8], d[10]};

    fcamd_eval_args args[2] = {FCAMD_ZERO_INIT, FCAMD_ZERO_INIT};
    for (int k = 0; k < 2; ++k) {
        args[k].grad_del_u = g_dev[k], args[k].stress_prev = stress_prev, args[k].stress = stress, args[k].tangent = tangent;
        args[k].parent_rows = d[5 + k];
    }
    args[1].history_prev = hist_prev, args[1].history = hist, args[1].n_hist = 2;
    const int64_t n[2] = {N0, N1};

    double* s_host = malloc(6 * N * 8);
    double* t_host = malloc(36 * N * 8);
    double worst = 0.0;
    for (int it = 1; it <= 2; ++it) { /* two Newton iterations: the same call, other gradient values */
        for (int k = 0; k < 2; ++k) {
            for (size_t i = 0; i < nb[k]; ++i) grad[k][9 * i] = 1e-6 * it * (double)(i % 97 + 1); /* eps_xx */
            CHECK(fcamd_copy_to_device(ctx, g_dev[k], grad[k], 9 * nb[k] * 8));
        }
        CHECK(fcamd_evaluate_batch(2, laws, n, args, 0.0, 1.0));
        CHECK(fcamd_context_synchronize(ctx));
        CHECK(fcamd_copy_to_host(ctx, s_host, stress, 6 * N * 8));
        CHECK(fcamd_copy_to_host(ctx, t_host, tangent, 36 * N * 8));
        for (int k = 0; k < 2; ++k) {
            /* LE: lam, mu from (E, nu); VonMises3D below yield: kappa, mu as given */
            const double mu = k == 0 ? 42.0 / (2.0 * 1.3) : 80769.0;
            const double lam = k == 0 ? 42.0 * 0.3 / (1.3 * 0.4) : 175000.0 - 2.0 * 80769.0 / 3.0;
            for (size_t i = 0; i < nb[k]; ++i) {
                const double e = grad[k][9 * i];
                const int r = rows[k][i];
                worst = fmax(worst, fabs(s_host[6 * r] - (2.0 * mu + lam) * e) / ((2.0 * mu + lam) * e));
                worst = fmax(worst, fabs(s_host[6 * r + 1] - lam * e) / (lam * e));
                worst = fmax(worst, fabs(t_host[36 * r + 21] - 2.0 * mu) / (2.0 * mu)); /* D[3][3] */
            }
        }
    }
    fcamd_stats st;
    CHECK(fcamd_model_last_stats(laws[1], &st));
    printf("fcamd v%d: 2 laws, %d + %d points in one call, max relative error %.3e, %llu plastic points\n", fcamd_version(), N0, N1, worst,
           (unsigned long long)st.n_plastic);
    for (int k = 0; k < 11; ++k) CHECK(fcamd_device_free(ctx, d[k]));
    CHECK(fcamd_model_destroy(laws[0]));
    CHECK(fcamd_model_destroy(laws[1]));
    CHECK(fcamd_context_destroy(ctx));
    return (worst < 1e-12 && st.n_plastic == 0) ? 0 : 2;
}
