/* A plain-C caller of the C ABI (include/fcamd.h): what a non-Python host (cf. the reference's
 * examples/elasticity_cpp/src/main.cpp:35-50) links against.  Evaluates LinearElasticityModel FULL
 * on a few points through the host (ndarray-style) entry and checks sigma = D : eps.
 *
 *   gcc -std=c99 -I include examples/c_caller.c -o c_caller -L fenics-constitutive_amd/lib -lfcamd -lm
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

int main(void) {
    enum { N = 1000 };
    const double E = 42.0, nu = 0.3, params[2] = {42.0, 0.3};
    const double mu = E / (2.0 * (1.0 + nu)), lam = E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu));
    double* grad = calloc(9 * N, sizeof(double));
    double* stress = calloc(6 * N, sizeof(double));
    double* tangent = calloc(36 * N, sizeof(double));
    for (int i = 0; i < N; ++i) grad[9 * i] = 1e-3 * (i + 1); /* uniaxial strain eps_xx */

    fcamd_context* ctx = NULL;
    fcamd_model* law = NULL;
    fcamd_stats stats;
    CHECK(fcamd_context_create(0, NULL, &ctx));
    CHECK(fcamd_model_create(ctx, FCAMD_LINEAR_ELASTICITY, FCAMD_FULL, params, 2, &law));
    /* the getters of the reference's native model classes (bindings/src/lib.rs:131-148): nothing is hard-coded */
    int constraint = 0, sd = 0, gdim = 0, n_hist = -1;
    CHECK(fcamd_model_constraint(law, &constraint));
    CHECK(fcamd_model_dims(law, &sd, &gdim));
    CHECK(fcamd_model_history_count(law, &n_hist));
    if (constraint != FCAMD_FULL || sd != 6 || gdim != 3 || n_hist != 0) return 3;
    /* the hot call, timed like the reference's Timer("constitutive-law-evaluation") (solver/_lawonsubmesh.py:86) */
    CHECK(fcamd_context_set_timing(ctx, 1));
    CHECK(fcamd_evaluate_host(law, 0.0, 1.0, N, grad, stress, tangent, NULL, 0, &stats));
    float ms = -1.0f;
    CHECK(fcamd_model_last_kernel_ms(law, &ms));

    double worst = 0.0;
    for (int i = 0; i < N; ++i) {
        const double e = grad[9 * i];
        worst = fmax(worst, fabs(stress[6 * i] - (2.0 * mu + lam) * e));
        worst = fmax(worst, fabs(stress[6 * i + 1] - lam * e));
        worst = fmax(worst, fabs(tangent[36 * i + 21] - 2.0 * mu)); /* D[3][3] */
    }
    printf("fcamd v%d: %d points, max error %.3e, call %.3f ms\n", fcamd_version(), N, worst, ms);
    CHECK(fcamd_model_destroy(law));
    CHECK(fcamd_context_destroy(ctx));
    free(grad), free(stress), free(tangent);
    return worst < 1e-12 ? 0 : 2;
}
