/*
 * oracle.c -- plain-C CPU restatement of the reference's quadrature-point updates.
 *
 * TEST INFRASTRUCTURE ONLY: this is the checker and the CPU baseline ("port"), never the
 * product.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * it.  Built by oracle/Makefile into oracle/_build/liboracle.so (gcc, -O2,
 * -ffp-contract=off so that the rounding is the one written here).
 *
 * Parity pin: tests/test_oracle_c.py checks every function against the golden vectors
 * captured from the imported reference (tests/golden/, oracle/gen_golden.py) and against
 * oracle/numpy_oracle.py.  The plastic values of the comfe-rs laws are not pinned by any
 * reference test of their own; they are pinned indirectly through outputs of the imported Python
 * VonMises3D; no output of the crate itself exists ("parity unpinned" in the literal sense; see
 * numpy_oracle.py).
 *
 * Serial loops over points mirror the reference's own structure:
 *   Python per-point loop   models/mises_plasticity_isotropic_hardening.py:74-175
 *   Rust evaluate_model     comfe-rs/src/interfaces.rs:441-455
 * NumPy's n x 6 . 6 x 6 product and 6-term np.dot are ascending-k FMA chains (verified
 * against OpenBLAS 0.3.29 Haswell in the build container) -> fma() below.
 *
 * Third-party arithmetic restated: nalgebra 0.32.6 (Cargo.lock:117-119) static 6x6 *
 * 6x1 product = column-wise axpy (y = A[:,0]*x0; y += A[:,j]*x_j, no FMA), norm_squared of
 * a 6-vector = sequential sum of squares.
 */
#include <math.h>
#include <stddef.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Threads used by the point loops.  Default 1 = the reference's serial loop (one MPI rank);
   bench.py also reports an all-cores figure.  Results do not depend on the thread count. */
static int g_threads = 1;
void oracle_set_num_threads(int n) { g_threads = n > 0 ? n : 1; }
int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
#define POINT_LOOP _Pragma("omp parallel for schedule(static) num_threads(g_threads)")
#define POINT_LOOP_COUNT(...) _Pragma("omp parallel for schedule(static) num_threads(g_threads) reduction(+ : bad, npl, nit)")

#define F_PY 0x1.6a09e667f3bccp-1 /* 1 / 2**0.5      (models/utils.py:202-204) */
#define F_RS 0x1.6a09e667f3bcdp-1 /* FRAC_1_SQRT_2   (comfe-rs/src/mandel.rs:147) */

static void strain6(const double* g, double f, double* e) {
    e[0] = g[0];
    e[1] = g[4];
    e[2] = g[8];
    e[3] = f * (g[1] + g[3]);
    e[4] = f * (g[2] + g[6]);
    e[5] = f * (g[5] + g[7]);
}

static void lame(double E, double nu, double* mu, double* lam) {
    *mu = E / (2.0 * (1.0 + nu));
    *lam = E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu));
}

static void tangent_full(double E, double nu, double* D) {
    double mu, lam;
    lame(E, nu, &mu, &lam);
    memset(D, 0, 36 * sizeof(double));
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) D[6 * i + j] = (i == j) ? 2.0 * mu + lam : lam;
    for (int i = 3; i < 6; ++i) D[6 * i + i] = 2.0 * mu;
}

/* y = x @ M as OpenBLAS does it: ascending-k FMA chain */
static void row_mat(const double* x, const double* M, double* y) {
    for (int i = 0; i < 6; ++i) {
        double acc = x[0] * M[i];
        for (int k = 1; k < 6; ++k) acc = fma(x[k], M[6 * k + i], acc);
        y[i] = acc;
    }
}

/* utils.py:132-208 (FULL); rust = 1 selects mandel.rs:143-171 */
void oracle_strain_from_grad_u(long long n, const double* grad, double* strain, int rust) {
    for (long long p = 0; p < n; ++p) strain6(grad + 9 * p, rust ? F_RS : F_PY, strain + 6 * p);
}

/* models/linear_elasticity_model.py:26-45 */
void oracle_linear_elasticity(double E, double nu, long long n, const double* grad, double* stress,
                              double* tangent) {
    double D[36];
    tangent_full(E, nu, D);
    POINT_LOOP
    for (long long p = 0; p < n; ++p) {
        double e[6], y[6];
        strain6(grad + 9 * p, F_PY, e);
        row_mat(e, D, y);
        for (int i = 0; i < 6; ++i) stress[6 * p + i] += y[i];
        if (tangent) memcpy(tangent + 36 * p, D, sizeof(D));
    }
}

/* models/mises_plasticity_isotropic_hardening.py:57-175.  Returns the number of points whose
   Newton iteration exceeded 100 steps (the reference raises RuntimeError at the first). */
long long oracle_von_mises_3d(double ka, double mu, double y0, double y00, double w, long long n,
                              const double* grad, double* stress, double* tangent, double* eps_n,
                              double* alpha, long long* n_plastic, long long* n_iter) {
    const double s23 = sqrt(2.0 / 3.0), two_mu = 2 * mu, dy = y00 - y0, mw = -w;
    const double m2mu = -2 * mu, c23 = (2.0 / 3.0) * dy * w, four_mu2 = 4 * mu * mu;
    long long bad = 0, npl = 0, nit = 0;
    POINT_LOOP_COUNT()
    for (long long p = 0; p < n; ++p) {
        double e[6], dsig[6], sigtr[6], N[6] = {0, 0, 0, 0, 0, 0};
        double* s = stress + 6 * p;
        strain6(grad + 9 * p, F_PY, e);
        const double tr_eps = (e[0] + e[1]) + e[2], tr_sig = (s[0] + s[1]) + s[2];
        for (int i = 0; i < 6; ++i) {
            const double ed = i < 3 ? e[i] - tr_eps / 3 : e[i];
            const double sd = i < 3 ? s[i] - tr_sig / 3 : s[i];
            dsig[i] = two_mu * ed;
            sigtr[i] = sd + dsig[i];
        }
        double nn = sigtr[0] * sigtr[0];
        for (int i = 1; i < 6; ++i) nn = fma(sigtr[i], sigtr[i], nn);
        const double sigtrn = sqrt(nn), a_n = alpha[p];
        const double phitr = sigtrn - s23 * (y0 + dy * (1 - exp(mw * a_n)));
        double gamma = 0, xc1 = 0, xc2 = 0;
        if (phitr > 0) {
            double g0 = 1, g1 = 0, xr = 1, xg;
            int it = 0;
            while (fabs(xr) > 1e-12 && fabs(g1 - g0) > 1e-8 * fabs(g1)) {
                g0 = g1;
                ++it;
                const double ex = exp(mw * (a_n + s23 * g0));
                xr = (sigtrn - two_mu * g0) - s23 * (y0 + dy * (1 - ex));
                xg = m2mu - c23 * ex;
                g1 = g0 - xr / xg;
                if (it > 100) {
                    ++bad;
                    break;
                }
            }
            xg = m2mu - c23 * exp(mw * (a_n + s23 * g1));
            xc1 = -1 / xg;
            xc2 = g1 / sigtrn;
            gamma = g1;
            for (int i = 0; i < 6; ++i) N[i] = sigtr[i] / sigtrn;
            ++npl;
            nit += it;
        }
        const double kt = ka * tr_eps, tmg = two_mu * gamma;
        for (int i = 0; i < 6; ++i) {
            eps_n[6 * p + i] += gamma * N[i];
            s[i] += ((i < 3 ? kt : kt * 0.0) + dsig[i]) - tmg * N[i];
        }
        alpha[p] += s23 * gamma;
        if (tangent) {
            const double B = two_mu * (1 - two_mu * xc2), C = four_mu2 * (xc2 - xc1);
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j) {
                    const double xioi = (i < 3 && j < 3) ? 1.0 : 0.0;
                    const double xpp = ((i == j) ? 1.0 : 0.0) - (1.0 / 3.0) * xioi;
                    tangent[36 * p + 6 * i + j] = (ka * xioi + B * xpp) + C * (N[i] * N[j]);
                }
        }
    }
    if (n_plastic) *n_plastic = npl;
    if (n_iter) *n_iter = nit;
    return bad;
}

/* models/spring_maxwell_model.py:40-88 */
void oracle_spring_maxwell(double E0, double E1, double tau, double nu, double del_t, long long n,
                           const double* grad, double* stress, double* tangent, double* ev,
                           double* en) {
    double D0[36], D1[36], D01[36], Dt[36], mu1, lam1;
    tangent_full(E0, nu, D0);
    tangent_full(E1, nu, D1);
    lame(E1, nu, &mu1, &lam1);
    const double factor = 1 / del_t + 1 / tau, inv_factor = 1 / factor;
    const double cA = 1 / (tau * 2 * mu1), cB = 1 / tau, c2mu = 2 * mu1, w1 = 1 - 1 / (tau * factor);
    for (int i = 0; i < 36; ++i) {
        D01[i] = D0[i] + D1[i];
        Dt[i] = D0[i] + w1 * D1[i];
    }
    POINT_LOOP
    for (long long p = 0; p < n; ++p) {
        double e[6], x[6], y[6], dv[6];
        strain6(grad + 9 * p, F_PY, e);
        for (int i = 0; i < 6; ++i) x[i] = cA * (en[6 * p + i] + e[i]);
        row_mat(x, D1, y);
        for (int i = 0; i < 6; ++i) dv[i] = inv_factor * (y[i] - cB * ev[6 * p + i]);
        row_mat(e, D01, y);
        for (int i = 0; i < 6; ++i) {
            stress[6 * p + i] += y[i] - c2mu * dv[i];
            ev[6 * p + i] += dv[i];
            en[6 * p + i] += e[i];
        }
        if (tangent) memcpy(tangent + 36 * p, Dt, sizeof(Dt));
    }
}

/* models/spring_kelvin_model.py:43-88 */
void oracle_spring_kelvin(double E0, double E1, double tau, double nu, double del_t, long long n,
                          const double* grad, double* stress, double* tangent, double* ev,
                          double* en) {
    double D0[36], Dt[36], mu0, lam0, mu1, lam1;
    tangent_full(E0, nu, D0);
    lame(E0, nu, &mu0, &lam0);
    lame(E1, nu, &mu1, &lam1);
    const double factor = 1 / del_t + 1 / tau + mu0 / (tau * mu1), inv_factor = 1 / factor;
    const double cA = 1 / (tau * 2 * mu1), cB = 1 / tau, cC = mu0 / (tau * mu1);
    const double cD = lam0 / (tau * 2 * mu1), c2mu = 2 * mu0, w0 = 1 - mu0 / (tau * mu1 * factor);
    for (int i = 0; i < 36; ++i) Dt[i] = w0 * D0[i];
    POINT_LOOP
    for (long long p = 0; p < n; ++p) {
        double e[6], y[6], dv[6];
        double* s = stress + 6 * p;
        strain6(grad + 9 * p, F_PY, e);
        const double ctr = cD * ((e[0] + e[1]) + e[2]);
        for (int i = 0; i < 6; ++i)
            dv[i] = inv_factor * (((cA * s[i] - cB * ev[6 * p + i]) + cC * e[i]) + (i < 3 ? ctr : ctr * 0.0));
        row_mat(e, D0, y);
        for (int i = 0; i < 6; ++i) {
            s[i] += y[i] - c2mu * dv[i];
            ev[6 * p + i] += dv[i];
            en[6 * p + i] += e[i];
        }
        if (tangent) memcpy(tangent + 36 * p, Dt, sizeof(Dt));
    }
}

/* comfe-rs/src/consts.rs:6-115 */
static void comfe_proj(double* soo, double* pvol, double* pdev) {
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            const double s = (i < 3 && j < 3) ? 1.0 : 0.0;
            soo[6 * i + j] = s;
            pvol[6 * i + j] = s * (1.0 / 3.0);
            pdev[6 * i + j] = ((i == j) ? 1.0 : 0.0) + pvol[6 * i + j] * -1.0;
        }
}

/* comfe-rs/src/linear_elasticity.rs:49-74 driven by interfaces.rs:441-455 */
void oracle_comfe_linear_elasticity(double mu, double kappa, long long n, const double* grad,
                                    double* stress, double* tangent) {
    double soo[36], pvol[36], pdev[36], Cm[36], Ct[36];
    comfe_proj(soo, pvol, pdev);
    for (int i = 0; i < 36; ++i) Cm[i] = (2.0 * mu) * pdev[i] + (3.0 * kappa) * pvol[i];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) Ct[6 * j + i] = Cm[6 * i + j];
    POINT_LOOP
    for (long long p = 0; p < n; ++p) {
        double e[6];
        strain6(grad + 9 * p, F_RS, e);
        for (int i = 0; i < 6; ++i) {
            double acc = Cm[6 * i] * e[0];
            for (int j = 1; j < 6; ++j) acc = Cm[6 * i + j] * e[j] + acc;
            stress[6 * p + i] += acc;
        }
        if (tangent) memcpy(tangent + 36 * p, Ct, sizeof(Ct));
    }
}

/* comfe-rs/src/mises_plasticity.rs:58-126 driven by interfaces.rs:441-455.
   hist: 7 doubles per point [alpha, plastic_strain(6)].  Returns the number of plastic points. */
long long oracle_comfe_mises(double mu, double kappa, double y_0, double h, long long n,
                             const double* grad, double* stress, double* tangent, double* hist) {
    double soo[36], pvol[36], pdev[36];
    comfe_proj(soo, pvol, pdev);
    const double two_mu = 2. * mu, den = 3. * mu + h, s32 = sqrt(3. / 2.), three_mu = 3. * mu;
    const double hfac = 1.0 / (1.0 + (h / (3.0 * mu)));
    long long npl = 0;
    _Pragma("omp parallel for schedule(static) num_threads(g_threads) reduction(+ : npl)")
    for (long long p = 0; p < n; ++p) {
        double e[6], s_tr[6], nv[6] = {0, 0, 0, 0, 0, 0};
        double* s = stress + 6 * p;
        double* hp = hist + 7 * p;
        strain6(grad + 9 * p, F_RS, e);
        const double alpha = hp[0];
        const double p_0 = ((s[0] + s[1]) + s[2]) / 3.0;
        const double eps_trace = (e[0] + e[1]) + e[2], eps_vol = eps_trace / 3.0;
        const double p_1 = p_0 + kappa * eps_trace;
        for (int i = 0; i < 6; ++i) {
            const double s0 = i < 3 ? s[i] + (-p_0) : s[i];
            const double ed = i < 3 ? e[i] + (-eps_vol) : e[i];
            s_tr[i] = s0 + two_mu * ed;
        }
        const double v3 = ((s_tr[0] + s_tr[1]) + s_tr[2]) / 3.0;
        double n2 = 0.0;
        for (int i = 0; i < 6; ++i) {
            const double d = i < 3 ? s_tr[i] + (-v3) : s_tr[i];
            n2 = i == 0 ? d * d : n2 + d * d;
        }
        const double q = sqrt(3.0 * (0.5 * n2)), sigma_y = y_0 + h * alpha;
        double theta = 1.0, B = two_mu, sc = 0.0;
        if (!(q < sigma_y)) {
            const double del_alpha = (q - sigma_y) / den, del_gamma = s32 * del_alpha;
            theta = 1.0 - (three_mu * del_alpha) / q;
            for (int i = 0; i < 6; ++i) {
                nv[i] = s_tr[i] / q;
                hp[1 + i] += del_gamma * nv[i];
            }
            hp[0] = alpha + del_alpha;
            B = two_mu * theta;
            sc = two_mu * (hfac - (1.0 - theta));
            ++npl;
        }
        for (int i = 0; i < 6; ++i) s[i] = i < 3 ? p_1 + theta * s_tr[i] : theta * s_tr[i];
        if (tangent)
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j) /* flat[6i+j] = element (row j, col i): column-major .data.0 */
                    tangent[36 * p + 6 * i + j] =
                        (kappa * soo[6 * j + i] + B * pdev[6 * j + i]) + (sc * nv[j]) * nv[i];
    }
    return npl;
}


/* ---------------------------------------------------------------------------------------------
   f4: comfe-rs general return mapping with the Drucker-Prager surfaces, per point as the Rust loop
   does it (comfe-rs/src/plasticity/general.rs:105-266; drucker_prager_classic.rs:62-116;
   drucker_prager_hyperbolic.rs:64-114; driven by interfaces.rs:441-455).  8 unknowns
   [sigma(6), del_lambda, alpha]; nalgebra's LU with partial pivoting is restated as Doolittle LU with
   row pivoting on the 8x8 matrix; the consistent tangent is (dres^-1)[0:6,0:6] . E, transposed and then
   stored column-major (= row-major of the product).  Pinned indirectly by outputs of the imported Python VonMises3D
   (tests/golden_util.py: dp_j2_cases, dp_pressure_cases, dp_volumetric_cases); PARITY UNPINNED in the literal sense (no
   output of the crate itself; see numpy_oracle.py): this is a second, independent restatement; tests compare the two.
   status (return value): number of plastic points; *flags |= 1 tip of the classic surface reached,
   |= 2 Newton did not converge, |= 4 singular system. */
typedef struct {
    double f, dfs[6], g[6], dgs[36], k, dks[6];
} dp_model;

static void dp_set_state(int hyper, double a_, double b, double bf, double dsq, const double* pdev,
                         const double* sig, dp_model* m, int* flags) {
    const double i_1 = (sig[0] + sig[1]) + sig[2];
    double s[6];
    for (int i = 0; i < 6; ++i) s[i] = i < 3 ? sig[i] + (-(i_1 / 3.0)) : sig[i];
    double j_2 = 0.0;
    for (int i = 0; i < 6; ++i) j_2 += s[i] * s[i];
    j_2 *= 0.5;
    double df_dj2, df_dj2j2, root;
    if (hyper) {
        root = sqrt(j_2 + dsq);
        df_dj2 = 0.5 * (1.0 / root);
        df_dj2j2 = -1.0 / 4.0 * pow(j_2 + dsq, -3.0 / 2.0);
    } else {
        if (!(i_1 < a_ / b)) *flags |= 1;
        root = sqrt(j_2);
        df_dj2 = 0.5 / root;
        df_dj2j2 = -0.25 / (j_2 * root);
    }
    m->f = root + b * i_1 - a_;
    double gn2 = 0.0;
    for (int i = 0; i < 6; ++i) {
        const double id = i < 3 ? 1.0 : 0.0;
        m->dfs[i] = b * id + df_dj2 * s[i];
        m->g[i] = bf * id + df_dj2 * s[i];
        gn2 += m->g[i] * m->g[i];
    }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) m->dgs[6 * i + j] = (s[i] * df_dj2j2) * s[j] + df_dj2 * pdev[6 * i + j];
    const double g_norm = sqrt(gn2), s23 = sqrt(2.0 / 3.0);
    m->k = s23 * g_norm;
    for (int j = 0; j < 6; ++j) {
        double acc = 0.0;
        for (int i = 0; i < 6; ++i) acc += m->g[i] * m->dgs[6 * i + j];
        m->dks[j] = (s23 / g_norm) * acc;
    }
}

static void dp_newton_matrix(const double* E, double dl, const dp_model* m, double* J) {
    for (int i = 0; i < 64; ++i) J[i] = 0.0;
    for (int i = 0; i < 6; ++i) {
        for (int j = 0; j < 6; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 6; ++k) acc += (E[6 * i + k] * dl) * m->dgs[6 * k + j];
            J[8 * i + j] = (i == j ? 1.0 : 0.0) + acc;
        }
        double eg = 0.0;
        for (int k = 0; k < 6; ++k) eg += E[6 * i + k] * m->g[k];
        J[8 * i + 6] = eg;
        J[8 * 6 + i] = m->dfs[i];
        J[8 * 7 + i] = (-dl) * m->dks[i];
    }
    J[8 * 7 + 6] = -m->k;
    J[8 * 7 + 7] = 1.0;
}

/* in-place LU with partial pivoting; returns 0 if singular */
static int lu8(double* A, int* piv) {
    for (int c = 0; c < 8; ++c) {
        int p = c;
        double best = fabs(A[8 * c + c]);
        for (int r = c + 1; r < 8; ++r)
            if (fabs(A[8 * r + c]) > best) best = fabs(A[8 * r + c]), p = r;
        if (best == 0.0) return 0;
        piv[c] = p;
        if (p != c)
            for (int j = 0; j < 8; ++j) {
                const double t = A[8 * c + j];
                A[8 * c + j] = A[8 * p + j];
                A[8 * p + j] = t;
            }
        for (int r = c + 1; r < 8; ++r) {
            const double l = A[8 * r + c] / A[8 * c + c];
            A[8 * r + c] = l;
            for (int j = c + 1; j < 8; ++j) A[8 * r + j] -= l * A[8 * c + j];
        }
    }
    return 1;
}

static void lu8_solve(const double* LU, const int* piv, double* x) {
    for (int c = 0; c < 8; ++c)
        if (piv[c] != c) {
            const double t = x[c];
            x[c] = x[piv[c]];
            x[piv[c]] = t;
        }
    for (int r = 1; r < 8; ++r)
        for (int j = 0; j < r; ++j) x[r] -= LU[8 * r + j] * x[j];
    for (int r = 7; r >= 0; --r) {
        for (int j = r + 1; j < 8; ++j) x[r] -= LU[8 * r + j] * x[j];
        x[r] /= LU[8 * r + r];
    }
}

long long oracle_comfe_drucker_prager(int hyper, double mu, double kappa, double a_, double b, double d,
                                      double b_flow, long long n, const double* grad, double* stress,
                                      double* tangent, double* hist, long long* n_iter, int* flags_out) {
    double soo[36], pvol[36], pdev[36], E[36], Einv[36];
    comfe_proj(soo, pvol, pdev);
    for (int i = 0; i < 36; ++i) {
        E[i] = (2.0 * mu) * pdev[i] + (3.0 * kappa) * pvol[i];
        Einv[i] = (2.0 * (1.0 / (4.0 * mu))) * pdev[i] + (3.0 * (1.0 / (9.0 * kappa))) * pvol[i];
    }
    const double dsq = d * d, atol = 1e-8, rtol = 1e-8;
    long long npl = 0, nit = 0;
    int flags = 0;
    _Pragma("omp parallel for schedule(static) num_threads(g_threads) reduction(+ : npl, nit) reduction(| : flags)")
    for (long long p = 0; p < n; ++p) {
        double e[6], sig_tr[6], sig0[6];
        double* s = stress + 6 * p;
        double* hp = hist + 7 * p;
        strain6(grad + 9 * p, F_RS, e);
        for (int i = 0; i < 6; ++i) {
            double acc = 0.0;
            for (int k = 0; k < 6; ++k) acc += E[6 * i + k] * e[k];
            sig0[i] = s[i];
            sig_tr[i] = acc + s[i];
        }
        dp_model m;
        int fl = 0;
        dp_set_state(hyper, a_, b, b_flow, dsq, pdev, sig_tr, &m, &fl);
        if (m.f <= 0.0) {
            for (int i = 0; i < 6; ++i) s[i] = sig_tr[i];
            if (tangent)
                for (int i = 0; i < 6; ++i)
                    for (int j = 0; j < 6; ++j) tangent[36 * p + 6 * i + j] = E[6 * j + i]; /* column-major .data.0 */
            flags |= fl;
            continue;
        }
        ++npl;
        const double alpha_0 = hp[0];
        double sol1[8], sol0[8], res[8], J[64], LU[64];
        int piv[8];
        for (int i = 0; i < 6; ++i) sol1[i] = sig_tr[i], res[i] = 0.0;
        sol1[6] = 0.0, sol1[7] = alpha_0;
        res[6] = m.f, res[7] = 0.0;
        dp_newton_matrix(E, 0.0, &m, J);
        int it = 0;
        for (;;) {
            ++nit;
            for (int i = 0; i < 8; ++i) sol0[i] = sol1[i];
            for (int i = 0; i < 64; ++i) LU[i] = J[i];
            double x[8];
            for (int i = 0; i < 8; ++i) x[i] = res[i];
            if (!lu8(LU, piv)) {
                fl |= 4;
                break;
            }
            lu8_solve(LU, piv, x);
            for (int i = 0; i < 8; ++i) sol1[i] = sol0[i] - x[i];
            const double dl = sol1[6];
            dp_set_state(hyper, a_, b, b_flow, dsq, pdev, sol1, &m, &fl);
            dp_newton_matrix(E, dl, &m, J);
            double rs2 = 0.0, ds2 = 0.0, s2 = 0.0;
            for (int i = 0; i < 6; ++i) {
                double eg = 0.0;
                for (int k = 0; k < 6; ++k) eg += E[6 * i + k] * m.g[k];
                res[i] = (sol1[i] - sig_tr[i]) + dl * eg;
                rs2 += res[i] * res[i];
                ds2 += (sol1[i] - sol0[i]) * (sol1[i] - sol0[i]);
                s2 += sol1[i] * sol1[i];
            }
            res[6] = m.f;
            res[7] = (sol1[7] - alpha_0) - m.k;
            const int conv_res = sqrt(rs2) < atol && fabs(res[7]) < atol && fabs(res[6]) < atol;
            const int conv_inc = sqrt(ds2) < atol + rtol * sqrt(s2) &&
                                 fabs(sol1[7] - sol0[7]) < atol + rtol * fabs(sol1[7]) &&
                                 fabs(dl - sol0[6]) < atol + rtol * fabs(dl);
            if (conv_res || conv_inc) break;
            if (it > 25) {
                fl |= 2;
                break;
            }
            ++it;
        }
        flags |= fl;
        for (int i = 0; i < 6; ++i) s[i] = sol1[i];
        hp[0] = sol1[7];
        for (int i = 0; i < 6; ++i) { /* plastic_strain += d_eps - E^-1 (sigma_1 - sigma_0) */
            double acc = 0.0;
            for (int k = 0; k < 6; ++k) acc += Einv[6 * i + k] * (sol1[k] - sig0[k]);
            hp[1 + i] += e[i] - acc;
        }
        if (tangent) {
            /* inverse of the last Jacobian, column by column */
            double inv[64];
            for (int i = 0; i < 64; ++i) LU[i] = J[i];
            if (!lu8(LU, piv)) {
                flags |= 4;
                continue;
            }
            for (int c = 0; c < 8; ++c) {
                double x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                x[c] = 1.0;
                lu8_solve(LU, piv, x);
                for (int r = 0; r < 8; ++r) inv[8 * r + c] = x[r];
            }
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j) {
                    double acc = 0.0;
                    for (int k = 0; k < 6; ++k) acc += inv[8 * i + k] * E[6 * k + j];
                    tangent[36 * p + 6 * i + j] = acc; /* transpose, then column-major = row-major of the product */
                }
        }
    }
    if (n_iter) *n_iter = nit;
    if (flags_out) *flags_out = flags;
    return npl;
}
