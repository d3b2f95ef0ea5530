// C-ABI layer of libfcamd: contexts, model handles, law constants and launches (host entries: fcamd_hostpath.cpp).
// Public contract: include/fcamd.h.  Device code: fcamd_kernels.hip.
//
// Everything that depends on material parameters is computed here, on the host, in plain
// IEEE double arithmetic and in the reference's own expression order (this file is built
// with -ffp-contract=off), so that the device kernels can reproduce NumPy bit for bit:
//   lame_parameters / get_elastic_tangent  src/fenics_constitutive/models/utils.py:18-51
//   VonMises3D.__init__                     models/mises_plasticity_isotropic_hardening.py:32-55
//   SpringMaxwellModel / SpringKelvinModel  models/spring_maxwell_model.py:24-38,72-86,
//                                           models/spring_kelvin_model.py:24-41,73-86
//   comfe-rs projections / tangents         comfe-rs/src/consts.rs:6-115, mandel.rs:126-128
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "fcamd_host.h"

using namespace fcamd;

namespace fcamd {

thread_local std::string g_last_error;

int fail(int status, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return status;
}

}  // namespace fcamd

namespace {

// Row-masked history access (fcamd_kernels.hip: tile_von_mises, history7_store) pays off while few rows
// of a tile are touched: every skipped row saves its bytes, but holes turn full-line writes into partial
// ones.  Tiles with more touched rows than this take the dense tile path.  Measured at 5e7 points
// (round-2 probe masked_threshold_probe.py (git history), random mixtures, sparse protocol; dense = 5.55 ms for VonMises3D):
// 5 % plastic 4.75 ms, 10 % 4.93, 19 % 5.23, 33 % 5.52 with thresholds 16-24, while "always masked" loses
// 4-6 % from 47 % plastic on; the 56-byte rows of the comfe-rs laws straddle chunks and turn earlier.
constexpr int kMaskedRowMaxVonMises = 20;
constexpr int kMaskedRowMaxRows7 = 16;

static_assert(fcamd::kCounterSlots == FCAMD_COUNTER_SLOTS, "public and internal counter layout differ");

// Python "1 / 2**0.5" and Rust FRAC_1_SQRT_2 differ by one ULP.
constexpr double kFactorPy = 0x1.6a09e667f3bccp-1;
constexpr double kFactorRs = 0x1.6a09e667f3bcdp-1;

bool law_info(int id, LawInfo* li) {
    switch (id) {
        case FCAMD_LINEAR_ELASTICITY: *li = {2, 0, {}, false}; return true;
        case FCAMD_VON_MISES_3D: *li = {5, 2, {{"eps_n", 6}, {"alpha", 1}}, false}; return true;
        case FCAMD_SPRING_MAXWELL: *li = {4, 2, {{"strain_visco", 6}, {"strain", 6}}, true}; return true;
        case FCAMD_SPRING_KELVIN: *li = {4, 2, {{"strain_visco", 6}, {"strain", 6}}, true}; return true;
        case FCAMD_COMFE_LINEAR_ELASTICITY: *li = {2, 0, {}, false}; return true;
        case FCAMD_COMFE_MISES_PLASTICITY: *li = {4, 1, {{"history", 7}}, false}; return true;
        case FCAMD_COMFE_DRUCKER_PRAGER: *li = {5, 1, {{"history", 7}}, false}; return true;
        case FCAMD_COMFE_DRUCKER_PRAGER_HYPERBOLIC: *li = {6, 1, {{"history", 7}}, false}; return true;
        default: return false;
    }
}

// ---- reference constants ------------------------------------------------------------------

void lame(double E, double nu, double* mu, double* lam) {
    *mu = E / (2.0 * (1.0 + nu));
    *lam = E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu));
}

void elastic_tangent_full(double E, double nu, double D[36]) {
    double mu, lam;
    lame(E, nu, &mu, &lam);
    for (int i = 0; i < 36; ++i) D[i] = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) D[6 * i + j] = (i == j) ? 2.0 * mu + lam : lam;
    for (int i = 3; i < 6; ++i) D[6 * i + i] = 2.0 * mu;
}

// stress_strain_dim / geometric_dim of a constraint (models/interfaces.py:30-73)
Dims dims_of(int constraint) {
    switch (constraint) {
        case FCAMD_UNIAXIAL_STRAIN:
        case FCAMD_UNIAXIAL_STRESS: return {1, 1, 1};
        case FCAMD_PLANE_STRAIN:
        case FCAMD_PLANE_STRESS: return {4, 4, 2};
        default: return {9, 6, 3};
    }
}

// get_elastic_tangent for every constraint (utils.py:25-93), compact row-major sd x sd
void elastic_tangent(double E, double nu, int constraint, double D[36]) {
    for (int i = 0; i < 36; ++i) D[i] = 0.0;
    double mu, lam;
    lame(E, nu, &mu, &lam);
    switch (constraint) {
        case FCAMD_FULL: elastic_tangent_full(E, nu, D); break;
        case FCAMD_PLANE_STRAIN:
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) D[4 * i + j] = (i == j) ? 2.0 * mu + lam : lam;
            D[15] = 2.0 * mu;
            break;
        case FCAMD_PLANE_STRESS: {
            const double c = E / (1 - std::pow(nu, 2.0));
            const double M[16] = {1.0, nu, 0.0, 0.0, nu, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, (1.0 - nu)};
            for (int i = 0; i < 16; ++i) D[i] = c * M[i];
            break;
        }
        case FCAMD_UNIAXIAL_STRAIN: D[0] = E * (1.0 - nu) / ((1.0 + nu) * (1.0 - 2.0 * nu)); break;
        case FCAMD_UNIAXIAL_STRESS: D[0] = E; break;
    }
}

// get_identity (utils.py:96-129)
void identity_of(int constraint, double I2[6]) {
    for (int i = 0; i < 6; ++i) I2[i] = 0.0;
    const int ones = (constraint == FCAMD_FULL || constraint == FCAMD_PLANE_STRAIN) ? 3
                     : (constraint == FCAMD_PLANE_STRESS)                           ? 2
                                                                                    : 1;
    for (int i = 0; i < ones; ++i) I2[i] = 1.0;
}

void comfe_projections(double soo[36], double pvol[36], double pdev[36]) {
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            const double s = (i < 3 && j < 3) ? 1.0 : 0.0;
            soo[6 * i + j] = s;
            pvol[6 * i + j] = s * (1.0 / 3.0);
            pdev[6 * i + j] = ((i == j) ? 1.0 : 0.0) + pvol[6 * i + j] * -1.0;
        }
}

}  // namespace

namespace {

// Scalars and tables of the law for one del_t (del_t enters the SLS constants only).
void fill_constants(const fcamd_model* m, double del_t, Scalars& sc, Tables& tb) {
    std::memset(&sc, 0, sizeof(sc));
    std::memset(&tb, 0, sizeof(tb));
    const double* p = m->params;
    switch (m->law) {
        case FCAMD_LINEAR_ELASTICITY: {
            sc.s[0] = kFactorPy;
            elastic_tangent(p[0], p[1], m->constraint, tb.a);
            std::memcpy(tb.c, tb.a, sizeof(tb.a));
            break;
        }
        case FCAMD_VON_MISES_3D: {
            const double ka = p[0], mu = p[1], y0 = p[2], y00 = p[3], w = p[4];
            sc.s[0] = kFactorPy;
            sc.s[1] = ka;
            sc.s[2] = 2 * mu;
            sc.s[3] = std::sqrt(2.0 / 3.0);
            sc.s[4] = y0;
            sc.s[5] = y00 - y0;
            sc.s[6] = -w;
            sc.s[7] = -2 * mu;
            sc.s[8] = (2.0 / 3.0) * (y00 - y0) * w;
            sc.s[9] = 4 * mu * mu;
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j) {
                    const double xioi = (i < 3 && j < 3) ? 1.0 : 0.0;
                    tb.a[6 * i + j] = ka * xioi;
                    tb.b[6 * i + j] = ((i == j) ? 1.0 : 0.0) - (1.0 / 3.0) * xioi;
                }
            break;
        }
        case FCAMD_SPRING_MAXWELL: {
            const double E0 = p[0], E1 = p[1], tau = p[2];
            const double nu = (m->constraint == FCAMD_UNIAXIAL_STRESS) ? 0.0 : p[3];  // spring_*_model.py:31-34
            double D0[36], D1[36], mu1, lam1;
            elastic_tangent(E0, nu, m->constraint, D0);
            elastic_tangent(E1, nu, m->constraint, D1);
            lame(E1, nu, &mu1, &lam1);
            const double factor = 1 / del_t + 1 / tau;
            sc.s[0] = kFactorPy;
            sc.s[1] = 1 / factor;
            sc.s[2] = 1 / (tau * 2 * mu1);
            sc.s[3] = 1 / tau;
            sc.s[4] = 2 * mu1;
            const double w1 = 1 - 1 / (tau * factor);
            for (int i = 0; i < 36; ++i) {
                tb.a[i] = D1[i];
                tb.b[i] = D0[i] + D1[i];
                tb.c[i] = D0[i] + w1 * D1[i];
            }
            break;
        }
        case FCAMD_SPRING_KELVIN: {
            const double E0 = p[0], E1 = p[1], tau = p[2];
            const double nu = (m->constraint == FCAMD_UNIAXIAL_STRESS) ? 0.0 : p[3];  // spring_*_model.py:31-34
            double D0[36], mu0, lam0, mu1, lam1, I2[6];
            elastic_tangent(E0, nu, m->constraint, D0);
            identity_of(m->constraint, I2);
            for (int i = 0; i < 6; ++i) sc.s[8 + i] = I2[i];
            lame(E0, nu, &mu0, &lam0);
            lame(E1, nu, &mu1, &lam1);
            const double factor = 1 / del_t + 1 / tau + mu0 / (tau * mu1);
            sc.s[0] = kFactorPy;
            sc.s[1] = 1 / factor;
            sc.s[2] = 1 / (tau * 2 * mu1);
            sc.s[3] = 1 / tau;
            sc.s[4] = 2 * mu0;
            sc.s[5] = mu0 / (tau * mu1);
            sc.s[6] = lam0 / (tau * 2 * mu1);
            const double w0 = 1 - mu0 / (tau * mu1 * factor);
            for (int i = 0; i < 36; ++i) {
                tb.a[i] = D0[i];
                tb.c[i] = w0 * D0[i];
            }
            break;
        }
        case FCAMD_COMFE_LINEAR_ELASTICITY: {
            const double mu = p[0], kappa = p[1];
            double soo[36], pvol[36], pdev[36];
            comfe_projections(soo, pvol, pdev);
            sc.s[0] = kFactorRs;
            for (int i = 0; i < 36; ++i) tb.a[i] = (2.0 * mu) * pdev[i] + (3.0 * kappa) * pvol[i];
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j) tb.c[6 * j + i] = tb.a[6 * i + j];  // column-major .data.0
            break;
        }
        case FCAMD_COMFE_MISES_PLASTICITY: {
            const double mu = p[0], kappa = p[1], y_0 = p[2], h = p[3];
            double soo[36], pvol[36], pdev[36];
            comfe_projections(soo, pvol, pdev);
            sc.s[0] = kFactorRs;
            sc.s[1] = mu;
            sc.s[2] = kappa;
            sc.s[3] = y_0;
            sc.s[4] = h;
            sc.s[5] = 2. * mu;
            sc.s[6] = 3. * mu + h;
            sc.s[7] = std::sqrt(3. / 2.);
            sc.s[8] = 3. * mu;
            sc.s[9] = 1.0 / (1.0 + (h / (3.0 * mu)));
            for (int i = 0; i < 36; ++i) {
                tb.a[i] = kappa * soo[i];
                tb.b[i] = pdev[i];
            }
            break;
        }
        case FCAMD_COMFE_DRUCKER_PRAGER:
        case FCAMD_COMFE_DRUCKER_PRAGER_HYPERBOLIC: {
            const bool hyper = m->law == FCAMD_COMFE_DRUCKER_PRAGER_HYPERBOLIC;
            const double mu = p[0], kappa = p[1], a_ = p[2], b = p[3];
            const double d = hyper ? p[4] : 0.0, b_flow = hyper ? p[5] : p[4];
            double soo[36], pvol[36], pdev[36];
            comfe_projections(soo, pvol, pdev);
            sc.s[0] = kFactorRs;
            sc.s[1] = mu;
            sc.s[2] = kappa;
            sc.s[3] = a_;
            sc.s[4] = b;
            sc.s[5] = b_flow;
            sc.s[6] = d * d;  // d.powi(2)
            sc.s[7] = 2.0 * mu;
            sc.s[8] = std::sqrt(2.0 / 3.0);
            sc.s[9] = 1.0 / (4.0 * mu);    // isotropic_elastic_tangent_inv, mandel.rs:130-141
            sc.s[10] = 1.0 / (9.0 * kappa);
            for (int i = 0; i < 36; ++i) {
                tb.a[i] = soo[i];
                tb.b[i] = pdev[i];
                tb.c[i] = (2.0 * mu) * pdev[i] + (3.0 * kappa) * pvol[i];  // E (symmetric)
            }
            break;
        }
    }
}

// FCAMD_* environment defaults of a new context: read here, once, never on the launch path
void options_from_env(Options* o) {
    auto geti = [](const char* name, long long dflt) {
        const char* e = getenv(name);
        return (e && *e) ? atoll(e) : dflt;
    };
    o->masked_max = (int)geti("FCAMD_MASKED_MAX", -1);
    o->batch_kernel = geti("FCAMD_BATCH_KERNEL", 1) != 0;
    o->host_chunk = geti("FCAMD_HOST_CHUNK", 0);
    o->host_slots = (int)std::min<long long>(fcamd_context::kSlots, std::max<long long>(1, geti("FCAMD_HOST_SLOTS", fcamd_context::kSlots)));
    o->zero_copy = geti("FCAMD_ZERO_COPY", 1) != 0;
    o->zero_copy_grad = geti("FCAMD_ZERO_COPY_GRAD", 1) != 0;
    o->bounce_max = std::max<long long>(0, geti("FCAMD_BOUNCE_MAX", 256 << 10));
    o->host_tangent_threads = (int)geti("FCAMD_HOST_TANGENT_THREADS", -1);
    o->host_tangent_min_points = std::max<long long>(0, geti("FCAMD_HOST_TANGENT_MIN", 1 << 16));
    o->host_tangent_chunk = std::max<long long>(0, geti("FCAMD_HOST_TANGENT_CHUNK", 0));
    o->host_tangent_streams = (int)std::max<long long>(1, std::min<long long>(fcamd_context::kSlots, geti("FCAMD_HOST_TANGENT_STREAMS", 1)));
}

// the law's host constants, recomputed only when del_t changes (SLS) -- not once per launch
void constants_for(fcamd_model* m, double del_t) {
    if (m->const_valid && (m->const_del_t == del_t || !m->info.needs_del_t)) return;
    fill_constants(m, del_t, m->sc, m->tb);
    m->const_del_t = del_t;
    m->const_valid = true;
}

bool law_counts(int law) { return law == FCAMD_VON_MISES_3D || law >= FCAMD_COMFE_MISES_PLASTICITY; }

// HIP-event bracket of the device entries (fcamd_context_set_timing; the analogue of the reference's
// Timer("constitutive-law-evaluation") around the hot call, solver/_lawonsubmesh.py:86).  The counters
// are reset before the timed window so that the events bracket the kernel(s) only.
int timing_begin(fcamd_model* m) {
    fcamd_context* c = m->ctx;
    m->timed = c->timing;
    m->host_ms = -1.0f;
    if (m->timed) {
        if (law_counts(m->law)) HIP_TRY(hipMemsetAsync(m->d_counters, 0, kCounterBytes, c->stream));
        HIP_TRY(hipEventRecord(m->ev0, c->stream));
    }
    return FCAMD_OK;
}
int timing_end(fcamd_model* m) {
    if (m->timed) HIP_TRY(hipEventRecord(m->ev1, m->ctx->stream));
    return FCAMD_OK;
}

}  // namespace

namespace fcamd {
int validate_call(const fcamd_model* m, double del_t, int64_t n, const void* grad,
                  const void* stress_prev, const void* stress, const void* const* hist_prev,
                  const void* const* hist, int n_hist, int flags) {
    if (!m) return fail(FCAMD_ERR_BAD_ARG, "model handle is NULL");
    if (n < 0) return fail(FCAMD_ERR_SIZE, "negative number of quadrature points");
    if ((flags & FCAMD_EVAL_SPLIT_HISTORY) && !has_split_history(m->law))
        return fail(FCAMD_ERR_UNSUPPORTED, "FCAMD_EVAL_SPLIT_HISTORY exists for the laws with one [scalar, eps_p(6)] history row per point");
    if (n_hist < 0 || n_hist > FCAMD_MAX_HISTORY)
        return fail(FCAMD_ERR_SIZE, "n_hist = %d: 0 .. %d history fields exist", n_hist, FCAMD_MAX_HISTORY);
    if (m->info.n_hist == 0 && n_hist != 0)
        return fail(FCAMD_ERR_SIZE, "law has no history, got %d history fields", n_hist);
    if (m->info.n_hist > 0) {
        const int expected = (flags & FCAMD_EVAL_SPLIT_HISTORY) ? 2 : m->info.n_hist;  // split: [scalar (n), rows (6 n)]
        if (!hist || !hist_prev || n_hist == 0)
            return fail(FCAMD_ERR_NULL_HISTORY, "history must not be None");
        if (n_hist != expected)
            return fail(FCAMD_ERR_SIZE, "law expects %d history fields, got %d", expected,
                        n_hist);
        for (int k = 0; k < n_hist; ++k)
            if (n > 0 && (!hist[k] || !hist_prev[k]))
                return fail(FCAMD_ERR_NULL_HISTORY, "history must not be None");
    }
    if (m->info.needs_del_t && !(del_t > 0.0))
        return fail(FCAMD_ERR_DEL_T, "Time step must be defined and positive.");
    if (n > 0 && (!grad || !stress || !stress_prev))
        return fail(FCAMD_ERR_BAD_ARG, "grad_del_u / stress pointer is NULL");
    return FCAMD_OK;
}
}  // namespace fcamd

namespace {

int grid_for(fcamd_model* m, int64_t n) {
    fcamd_context* c = m->ctx;
    int grid = c->grid_override;
    if (grid <= 0) {
        if (m->grid_auto <= 0) {
            m->grid_auto = default_grid(m->law, c->num_cu);
            // Drucker-Prager (3 waves per SIMD, the longest tile): a quarter of that -- 9.08 / 9.11 / 9.25 / 9.18 / 9.37 ms at 32768 /
            // 65536 / 131072 / 262144 / 390625 workgroups (round 5, tools/ab_lib.py path.so@GRID, 1e8 points, identical buffers)
            if (m->law == FCAMD_COMFE_DRUCKER_PRAGER || m->law == FCAMD_COMFE_DRUCKER_PRAGER_HYPERBOLIC) m->grid_auto /= 4;
        }
        grid = m->grid_auto;
    }
    const int64_t tiles = (n + 63) / 64;
    // an explicit grid is capped at one tile per wave; the automatic one keeps two tiles per wave (fcamd_kernels.hip: default_grid)
    const int64_t need = c->grid_override > 0 ? (tiles + 3) / 4 : (tiles + 7) / 8;
    if (need < grid) grid = (int)std::max<int64_t>(need, 1);
    return grid;
}

// enqueue one launch on `stream`; device pointers already validated
}  // namespace

namespace fcamd {
constexpr int kFlagExactTangentRows = 16;  // kernels/tangent_writers.h (library-internal bit of EvalArgs::flags)

// does the device address `p` lie in host memory this context has mapped (registered caller ranges, the page-locked scratch)?
static bool host_mapped(fcamd_context* c, const void* p) {
    const char* q = static_cast<const char*>(p);
    if (c->bounce_dev && q >= c->bounce_dev && q < c->bounce_dev + c->bounce_bytes) return true;
    std::lock_guard<std::recursive_mutex> lock(c->host_mu);
    for (const auto& kv : c->registered)
        if (kv.second.dev && q >= kv.second.dev && q < kv.second.dev + kv.second.bytes) return true;
    return false;
}

// the kernel arguments of one launch (device pointers already validated)
static void fill_args(fcamd_model* m, double del_t, int64_t n, const double* grad, const double* stress_prev,
                      double* stress, double* tangent, const double* const* hprev, double* const* hcur, const int* rows,
                      unsigned long long* hmask, int flags, double* stress2,
                      unsigned long long* counters, const unsigned long long* emask_prev, unsigned long long* emask, EvalArgs& a) {
    a.grad = grad;
    a.stress_in = stress_prev;
    a.stress_out = stress;
    a.stress_out2 = stress2;
    a.tangent = tangent;
    const bool split = (flags & FCAMD_EVAL_SPLIT_HISTORY) != 0 && has_split_history(m->law);
    const int nh = split ? 2 : m->info.n_hist;
    a.h0_in = nh > 0 ? hprev[0] : nullptr;
    a.h0_out = nh > 0 ? hcur[0] : nullptr;
    a.h1_in = nh > 1 ? hprev[1] : nullptr;
    a.h1_out = nh > 1 ? hcur[1] : nullptr;
    a.rows = rows;
    a.cache3d = nullptr;
    a.hmask = hmask;
    a.flags = split ? FCAMD_EVAL_SPLIT_HISTORY : 0;
    if (hmask) {
        if (tangent) a.flags |= flags & FCAMD_EVAL_SPARSE_TANGENT;  // needs an array that holds the previous tangent
        // the tangent is page-locked host memory written over PCIe (a registered range or the bounce scratch): bare rows, no granules
        if ((a.flags & FCAMD_EVAL_SPARSE_TANGENT) && host_mapped(m->ctx, tangent)) a.flags |= kFlagExactTangentRows;
        // (with parent_rows: VonMises3D only -- the indexed split-history kernels would need instantiations of their own)
        if ((m->law == FCAMD_VON_MISES_3D || (split && !rows)) && emask_prev && emask) a.flags |= flags & FCAMD_EVAL_PACKED_HISTORY;
    }
    // (measurement device, context option "twin_masks":) the synthetic twin of the packed sparse-protocol VonMises3D launch
    if (m->ctx->twin_masks && m->law == FCAMD_VON_MISES_3D && !rows && (a.flags & FCAMD_EVAL_PACKED_HISTORY)) {
        a.flags |= 64;  // kernels/tangent_writers.h: kFlagTwin
        a.cache3d = reinterpret_cast<double*>(m->ctx->twin_masks);
    }
    // (library-internal, host entries only:) `tangent` is the ring of 8 doubles per point, the host rebuilds the rows
    if (tangent && (flags & kFlagTangentParamsHost)) a.flags = (a.flags & ~FCAMD_EVAL_SPARSE_TANGENT) | kFlagTangentParamsHost;
    a.emask_in = emask_prev;
    a.emask_out = emask;
    a.n = n;
    a.counters = counters ? counters : m->d_counters;  // caller-owned counters are always reset by the launch
    const Options& o = m->ctx->opt;
    a.masked_max = o.masked_max >= 0 ? o.masked_max  // split history: 48-byte eps_p rows, as VonMises3D's eps_n
                                     : ((m->law == FCAMD_VON_MISES_3D || split) ? kMaskedRowMaxVonMises : kMaskedRowMaxRows7);
    constants_for(m, del_t);
    a.sc = m->sc;
    a.tb = m->tb;
}

int enqueue(fcamd_model* m, double del_t, int64_t n, const double* grad, const double* stress_prev,
            double* stress, double* tangent, const double* const* hprev, double* const* hcur,
            hipStream_t stream, bool reset_counters, const int* rows,
            unsigned long long* hmask, int flags, double* stress2,
            unsigned long long* counters, const unsigned long long* emask_prev, unsigned long long* emask) {
    EvalArgs a;
    fill_args(m, del_t, n, grad, stress_prev, stress, tangent, hprev, hcur, rows, hmask, flags, stress2, counters, emask_prev, emask, a);
    // only the plasticity laws count anything: skip the extra launch for the others
    if ((reset_counters || counters) && law_counts(m->law)) HIP_TRY(hipMemsetAsync(a.counters, 0, kCounterBytes, stream));
    if (n == 0) return FCAMD_OK;
    const int grid = grid_for(m, n);
    // the launchers report hipGetLastError(): drop whatever an earlier, unrelated call of this thread
    // (ours, the caller's, torch's) left behind, so that a failure reported here is this launch's
    (void)hipGetLastError();
    HIP_TRY(launch_evaluate(m->law, m->dims.gdim, a, grid, stream));
    return FCAMD_OK;
}
}  // namespace fcamd

namespace fcamd {
void constants_for_call(fcamd_model* m, double del_t) { constants_for(m, del_t); }

void sum_counters(const fcamd_model* m, fcamd_stats* out) {
    out->n_nonconverged = out->n_plastic = out->n_newton_iters = out->n_domain = 0;
    out->kernel_ms = -1.0;  // fcamd_model_last_stats fills it
    for (int s = 0; s < fcamd::kCounterSlots; ++s) {
        out->n_nonconverged += m->h_counters[4 * s + 0];
        out->n_plastic += m->h_counters[4 * s + 1];
        out->n_newton_iters += m->h_counters[4 * s + 2];
        out->n_domain += m->h_counters[4 * s + 3];
    }
}

int enqueue_counters_download(fcamd_model* m, hipStream_t stream) {
    HIP_TRY(hipMemcpyAsync(m->h_counters, m->d_counters, kCounterBytes, hipMemcpyDeviceToHost, stream));
    return FCAMD_OK;
}

int read_stats(fcamd_model* m, hipStream_t stream, fcamd_stats* out) {
    int st = enqueue_counters_download(m, stream);
    if (st != FCAMD_OK) return st;
    HIP_TRY(hipStreamSynchronize(stream));
    if (out) sum_counters(m, out);
    return FCAMD_OK;
}
}  // namespace fcamd

static int evaluate_wrapped(fcamd_model* m, int wrapper_constraint, double del_t, int64_t n, const double* grad_lo, double* stress_lo,
                            double* tangent_lo, double* stress_3d, double* const* hist, int n_hist);

extern "C" {

int fcamd_version(void) { return FCAMD_VERSION_MAJOR * 1000 + FCAMD_VERSION_MINOR; }

int fcamd_device_count(int* count) {
    if (!count) return fail(FCAMD_ERR_BAD_ARG, "NULL argument");
    *count = 0;
    if (hipGetDeviceCount(count) != hipSuccess) {
        (void)hipGetLastError();
        *count = 0;
    }
    return FCAMD_OK;
}

const char* fcamd_last_error(void) { return g_last_error.c_str(); }

int fcamd_context_create(int device, void* stream, fcamd_context** out) {
    if (!out) return fail(FCAMD_ERR_BAD_ARG, "out is NULL");
    *out = nullptr;
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count)
        return fail(FCAMD_ERR_BAD_ARG, "device %d out of range (%d devices)", device, count);
    HIP_TRY(hipSetDevice(device));
    fcamd_context* c = new (std::nothrow) fcamd_context();
    if (!c) return fail(FCAMD_ERR_BAD_ARG, "out of host memory");
    c->device = device;
    options_from_env(&c->opt);
    const int st = [&]() -> int {  // any failure below must not leak the context
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (stream) {
            c->stream = static_cast<hipStream_t>(stream);
        } else {
            HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
            c->owns_stream = true;
        }
        return FCAMD_OK;
    }();
    if (st != FCAMD_OK) {
        delete c;
        return st;
    }
    *out = c;
    return FCAMD_OK;
}

static int context_trim(fcamd_context* c) {
    if (!c) return fail(FCAMD_ERR_BAD_ARG, "context is NULL");
    std::lock_guard<std::recursive_mutex> lock(c->host_mu);
    HIP_TRY(hipSetDevice(c->device));
    for (int i = 0; i < fcamd_context::kSlots; ++i)
        if (c->hstream[i]) HIP_TRY(hipStreamSynchronize(c->hstream[i]));
    free_host_staging(c);
    fcamd::host_tangent_release(c);
    return FCAMD_OK;
}

int fcamd_context_set_option(fcamd_context* c, const char* name, long long value) {
    if (!c || !name) return fail(FCAMD_ERR_BAD_ARG, "NULL argument");
    std::lock_guard<std::recursive_mutex> lock(c->host_mu);
    const std::string k(name);
    Options& o = c->opt;
    if (k == "masked_max") o.masked_max = (int)value;
    else if (k == "batch_kernel") o.batch_kernel = value != 0;
    else if (k == "host_chunk") o.host_chunk = value;
    else if (k == "host_slots") o.host_slots = (int)std::min<long long>(fcamd_context::kSlots, std::max<long long>(1, value));
    else if (k == "zero_copy") o.zero_copy = value != 0;
    else if (k == "zero_copy_grad") o.zero_copy_grad = value != 0;
    else if (k == "bounce_max") o.bounce_max = std::max<long long>(0, value);
    else if (k == "host_tangent_threads") o.host_tangent_threads = (int)std::max<long long>(-1, std::min<long long>(value, 256));
    else if (k == "host_tangent_min_points") o.host_tangent_min_points = std::max<long long>(0, value);
    else if (k == "host_tangent_chunk") o.host_tangent_chunk = std::max<long long>(0, (value / 64) * 64);
    else if (k == "host_tangent_streams") o.host_tangent_streams = (int)std::max<long long>(1, std::min<long long>(fcamd_context::kSlots, value));
    else if (k == "twin_masks") c->twin_masks = (unsigned long long)value;  // device address of one recorded ballot word per tile; 0: off
    else if (k == "grid") c->grid_override = value > 0 ? (int)value : 0;
    else if (k == "timing") c->timing = value != 0;
    else if (k == "trim") return context_trim(c);
    else if (k == "peer_access") return fcamd::enable_peer_access(c, (int)value);
    else return fail(FCAMD_ERR_BAD_ARG, "unknown option '%s'", name);
    return FCAMD_OK;
}

int fcamd_context_get_option(fcamd_context* c, const char* name, long long* value) {
    if (!c || !name || !value) return fail(FCAMD_ERR_BAD_ARG, "NULL argument");
    const std::string k(name);
    const Options& o = c->opt;
    if (k == "masked_max") *value = o.masked_max;
    else if (k == "batch_kernel") *value = o.batch_kernel ? 1 : 0;
    else if (k == "host_chunk") *value = o.host_chunk;
    else if (k == "host_slots") *value = o.host_slots;
    else if (k == "zero_copy") *value = o.zero_copy;
    else if (k == "zero_copy_grad") *value = o.zero_copy_grad;
    else if (k == "bounce_max") *value = o.bounce_max;
    else if (k == "host_tangent_threads") *value = fcamd::host_tangent_threads(c);  // resolved (-1 -> the automatic count)
    else if (k == "host_tangent_min_points") *value = o.host_tangent_min_points;
    else if (k == "host_tangent_chunk") *value = o.host_tangent_chunk;
    else if (k == "host_tangent_streams") *value = o.host_tangent_streams;
    else if (k == "last_host_tangent_cpu_us") *value = c->last_host_tangent_cpu_us;
    else if (k == "last_host_tangent_threads") *value = c->last_host_tangent_threads;
    else if (k == "grid") *value = c->grid_override;
    else if (k == "timing") *value = c->timing ? 1 : 0;
    else if (k == "last_host_mode") *value = c->last_host_mode;
    else return fail(FCAMD_ERR_BAD_ARG, "unknown option '%s'", name);
    return FCAMD_OK;
}

int fcamd_context_destroy(fcamd_context* c) {
    if (!c) return FCAMD_OK;
    (void)hipSetDevice(c->device);
    {
        std::lock_guard<std::recursive_mutex> lock(c->host_mu);
        release_registered_ranges(c);  // the page locks it shares: the last context to leave unlocks the pages
    }
    free_host_staging(c);
    fcamd::host_tangent_release(c);
    for (int i = 0; i < fcamd_context::kSlots; ++i)
        if (c->hstream[i]) (void)hipStreamDestroy(c->hstream[i]);
    for (hipStream_t s : c->peer_streams)
        if (s) (void)hipStreamDestroy(s);
    for (hipEvent_t e : c->peer_events)
        if (e) (void)hipEventDestroy(e);
    for (auto& kv : c->ipc_open) (void)hipIpcCloseMemHandle(kv.second.base);
    for (auto& sl : c->batch_slots) {
        if (sl.host) (void)hipHostFree(sl.host);
        if (sl.dev) (void)hipFree(sl.dev);
        if (sl.uploaded) (void)hipEventDestroy(sl.uploaded);
    }
    if (c->owns_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return FCAMD_OK;
}

int fcamd_context_set_stream(fcamd_context* c, void* stream) {
    if (!c) return fail(FCAMD_ERR_BAD_ARG, "context is NULL");
    if (c->owns_stream && c->stream) {
        (void)hipStreamDestroy(c->stream);
        c->owns_stream = false;
    }
    c->stream = static_cast<hipStream_t>(stream);
    return FCAMD_OK;
}

int fcamd_context_synchronize(fcamd_context* c) {
    if (!c) return fail(FCAMD_ERR_BAD_ARG, "context is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FCAMD_OK;
}

int fcamd_model_create(fcamd_context* c, int model_id, int constraint, const double* params,
                       int n_params, fcamd_model** out) {
    if (!out) return fail(FCAMD_ERR_BAD_ARG, "out is NULL");
    *out = nullptr;
    if (!c) return fail(FCAMD_ERR_BAD_ARG, "context is NULL");
    LawInfo li;
    if (!law_info(model_id, &li)) return fail(FCAMD_ERR_BAD_ARG, "unknown model id %d", model_id);
    if (constraint < FCAMD_UNIAXIAL_STRAIN || constraint > FCAMD_FULL)
        return fail(FCAMD_ERR_BAD_ARG, "unknown constraint %d", constraint);
    const bool all_constraints = model_id == FCAMD_LINEAR_ELASTICITY || model_id == FCAMD_SPRING_MAXWELL ||
                                 model_id == FCAMD_SPRING_KELVIN;
    if (constraint != FCAMD_FULL && !all_constraints)
        return fail(FCAMD_ERR_UNSUPPORTED, "model %d is implemented for StressStrainConstraint.FULL only (got %d)",
                    model_id, constraint);
    if (!params || n_params != li.n_params)
        return fail(FCAMD_ERR_BAD_ARG, "model %d expects %d parameters, got %d", model_id,
                    li.n_params, n_params);
    HIP_TRY(hipSetDevice(c->device));
    fcamd_model* m = new (std::nothrow) fcamd_model();
    if (!m) return fail(FCAMD_ERR_BAD_ARG, "out of host memory");
    m->ctx = c;
    m->law = model_id;
    m->constraint = constraint;
    m->dims = dims_of(constraint);
    if (all_constraints)
        for (int k = 0; k < li.n_hist; ++k) li.hist[k].dim = m->dims.sd;  // history_dim = stress_strain_dim
    m->info = li;
    for (int i = 0; i < n_params; ++i) m->params[i] = params[i];
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&m->d_counters), kCounterBytes);
    if (e == hipSuccess)
        e = hipHostMalloc(reinterpret_cast<void**>(&m->h_counters), kCounterBytes,
                          hipHostMallocDefault);
    if (e == hipSuccess) e = hipMemset(m->d_counters, 0, kCounterBytes);
    // the fill runs on the null stream and every launch on a non-blocking stream that does not wait for it: without
    // this wait it can land on top of the counts of the handle's first evaluate (found in round 3: 9157 instead of 9219
    // plastic points reported by a law's very first call)
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipEventCreate(&m->ev0);
    if (e == hipSuccess) e = hipEventCreate(&m->ev1);
    if (e != hipSuccess) {
        fcamd_model_destroy(m);
        return fail(FCAMD_ERR_HIP, "model allocation failed: %s", hipGetErrorString(e));
    }
    std::memset(m->h_counters, 0, kCounterBytes);
    *out = m;
    return FCAMD_OK;
}

int fcamd_model_destroy(fcamd_model* m) {
    if (!m) return FCAMD_OK;
    if (m->ctx) (void)hipSetDevice(m->ctx->device);
    if (m->d_counters) (void)hipFree(m->d_counters);
    if (m->h_counters) (void)hipHostFree(m->h_counters);
    if (m->ev0) (void)hipEventDestroy(m->ev0);
    if (m->ev1) (void)hipEventDestroy(m->ev1);
    delete m;
    return FCAMD_OK;
}

int fcamd_model_get_info(const fcamd_model* m, fcamd_model_info* info) {
    if (!m || !info) return fail(FCAMD_ERR_BAD_ARG, "NULL argument");
    std::memset(info, 0, sizeof(*info));
    info->model_id = m->law;
    info->constraint = m->constraint;
    info->stress_strain_dim = m->dims.sd;
    info->geometric_dim = m->dims.gdim;
    info->n_history = m->info.n_hist;
    for (int k = 0; k < m->info.n_hist && k < FCAMD_MAX_HISTORY; ++k) {
        info->history_name[k] = m->info.hist[k].name;
        info->history_dim[k] = m->info.hist[k].dim;
    }
    return FCAMD_OK;
}

// the checks of one device call (every form but the fused wrapper one); nothing is launched
static int check_device_ex(fcamd_model* m, double del_t, int64_t n, const fcamd_eval_args* x) {
    int st = validate_call(m, del_t, n, x->grad_del_u, x->stress_prev, x->stress,
                           reinterpret_cast<const void* const*>(x->history_prev),
                           reinterpret_cast<const void* const*>(x->history), x->n_hist, x->flags);
    if (st != FCAMD_OK) return st;
    if (x->flags & ~(FCAMD_EVAL_SPARSE_TANGENT | FCAMD_EVAL_SPLIT_HISTORY | FCAMD_EVAL_PACKED_HISTORY))
        return fail(FCAMD_ERR_UNSUPPORTED, "unknown FCAMD_EVAL_* flag in 0x%x (2, FCAMD_EVAL_DELTA_HISTORY of ABI 0.3, was removed in 0.4)", x->flags);
    if (x->parent_rows && m->constraint != FCAMD_FULL)
        return fail(FCAMD_ERR_UNSUPPORTED, "the indexed form exists for StressStrainConstraint.FULL only");
    if (x->history_mask && !has_sparse_history(m->law))
        return fail(FCAMD_ERR_UNSUPPORTED, "sparse trial history exists for the plasticity laws only");
    if ((x->flags & FCAMD_EVAL_SPARSE_TANGENT) && (!x->history_mask || !x->tangent))
        return fail(FCAMD_ERR_BAD_ARG, "FCAMD_EVAL_SPARSE_TANGENT needs history_mask and tangent");
    if ((x->flags & FCAMD_EVAL_PACKED_HISTORY) && n > 0) {
        const bool split = (x->flags & FCAMD_EVAL_SPLIT_HISTORY) != 0 && has_split_history(m->law);
        if (m->law != FCAMD_VON_MISES_3D && !split)
            return fail(FCAMD_ERR_UNSUPPORTED, "FCAMD_EVAL_PACKED_HISTORY exists for VonMises3D and, with FCAMD_EVAL_SPLIT_HISTORY, for the "
                                               "comfe-rs plasticity laws");
        if (!x->history_mask || !x->packed_mask_prev || !x->packed_mask)
            return fail(FCAMD_ERR_BAD_ARG, "FCAMD_EVAL_PACKED_HISTORY needs history_mask, packed_mask_prev and packed_mask");
        if (x->parent_rows && m->law != FCAMD_VON_MISES_3D)
            return fail(FCAMD_ERR_UNSUPPORTED, "FCAMD_EVAL_PACKED_HISTORY with parent_rows: VonMises3D only");
        const int kd = split ? 1 : 0;
        if (x->history[kd] == x->history_prev[kd] || x->packed_mask == x->packed_mask_prev)
            return fail(FCAMD_ERR_BAD_ARG, "FCAMD_EVAL_PACKED_HISTORY needs trial plastic-strain and mask arrays of their own");
    }
    if (!aligned16(x->grad_del_u) || !aligned16(x->stress) || !aligned16(x->stress_prev) || !aligned16(x->tangent) || !aligned16(x->stress2))
        return fail(FCAMD_ERR_ALIGN, "device arrays must be 16-byte aligned");
    for (int k = 0; k < x->n_hist; ++k)
        if (!aligned16(x->history[k]) || !aligned16(x->history_prev[k]))
            return fail(FCAMD_ERR_ALIGN, "device history arrays must be 16-byte aligned");
    if (x->flags & FCAMD_EVAL_SPLIT_HISTORY) {
        if (m->constraint != FCAMD_FULL) return fail(FCAMD_ERR_UNSUPPORTED, "FCAMD_EVAL_SPLIT_HISTORY: 3-D laws only");
        if ((x->history[0] == x->history_prev[0]) != (x->history[1] == x->history_prev[1]))
            return fail(FCAMD_ERR_BAD_ARG, "FCAMD_EVAL_SPLIT_HISTORY: both history arrays in place or both out of place");
    }
    return FCAMD_OK;
}

static int enqueue_ex(fcamd_model* m, double del_t, int64_t n, const fcamd_eval_args* x, hipStream_t stream, bool reset_counters) {
    return enqueue(m, del_t, n, x->grad_del_u, x->stress_prev, x->stress, x->tangent, x->history_prev, x->history,
                   stream, reset_counters, x->parent_rows, reinterpret_cast<unsigned long long*>(x->history_mask), x->flags,
                   x->stress2, reinterpret_cast<unsigned long long*>(x->counters),
                   (x->flags & FCAMD_EVAL_PACKED_HISTORY) ? reinterpret_cast<const unsigned long long*>(x->packed_mask_prev) : nullptr,
                   (x->flags & FCAMD_EVAL_PACKED_HISTORY) ? reinterpret_cast<unsigned long long*>(x->packed_mask) : nullptr);
}

int fcamd_evaluate_device_ex(fcamd_model* m, double t, double del_t, int64_t n, const fcamd_eval_args* x) {
    (void)t;
    if (!x) return fail(FCAMD_ERR_BAD_ARG, "args is NULL");
    if (x->wrapper_constraint != 0) {  // the fused 3D -> 1D/2D wrapper form
        if (x->parent_rows || x->history_mask || x->stress2 || x->counters || x->flags != 0)
            return fail(FCAMD_ERR_UNSUPPORTED, "the fused wrapper form takes no parent_rows / history_mask / stress2 / counters / flags");
        if (x->stress_prev != x->stress || (x->n_hist > 0 && x->history_prev != const_cast<const double* const*>(x->history)))
            return fail(FCAMD_ERR_BAD_ARG, "the fused wrapper form is in place: stress_prev == stress, history_prev == history");
        return evaluate_wrapped(m, x->wrapper_constraint, del_t, n, x->grad_del_u, x->stress, x->tangent, x->stress_3d, x->history, x->n_hist);
    }
    int st = check_device_ex(m, del_t, n, x);
    if (st != FCAMD_OK) return st;
    fcamd_context* c = m->ctx;
    HIP_TRY(hipSetDevice(c->device));
    if ((st = timing_begin(m)) != FCAMD_OK) return st;
    st = enqueue_ex(m, del_t, n, x, c->stream, !m->timed);
    if (st != FCAMD_OK) return st;
    return timing_end(m);
}

// The laws of one form() in one call (the reference calls them back to back: solver/_solver.py:143-144, one
// LawOnSubMesh.evaluate per material).  Every call is checked first -- nothing is launched if one of them is refused.  Laws
// that fill the device on their own keep their own launches (the kernels cut for their register budget); the others -- as
// separate launches three dispatches each: counters, main kernel, ragged tile, ~30 us of stream time per law at 1e4 points --
// leave as ONE launch of the batch kernel (fcamd_kernels.hip: evaluate_batch_kernel), which reads each law's arguments from a
// table in device memory.  The table is uploaded only when it CHANGES: between the Newton iterations of an increment nothing but
// the gradients' values does, so an iteration costs two dispatches (counters, batch kernel) whatever the number of laws.
// Results are bit for bit those of the same calls made one by one (same tile code, same arguments).
int fcamd_evaluate_batch(int count, fcamd_model* const* models, const int64_t* n, const fcamd_eval_args* args, double t, double del_t) {
    if (count < 0 || (count > 0 && (!models || !n || !args))) return fail(FCAMD_ERR_BAD_ARG, "NULL argument");
    if (count == 0) return FCAMD_OK;
    fcamd_context* c = models[0] ? models[0]->ctx : nullptr;
    for (int k = 0; k < count; ++k) {
        if (!models[k]) return fail(FCAMD_ERR_BAD_ARG, "models[%d] is NULL", k);
        if (models[k]->ctx != c) return fail(FCAMD_ERR_BAD_ARG, "fcamd_evaluate_batch: models[%d] belongs to another context", k);
        if (args[k].wrapper_constraint != 0) return fail(FCAMD_ERR_UNSUPPORTED, "fcamd_evaluate_batch: the fused wrapper form is not batched (models[%d])", k);
        const int st = check_device_ex(models[k], del_t, n[k], &args[k]);
        if (st != FCAMD_OK) return st;
    }
    if (c->timing) {  // context option "timing": every law bracketed by its own events, as fcamd_model_last_stats reports them
        for (int k = 0; k < count; ++k) {
            const int st = fcamd_evaluate_device_ex(models[k], t, del_t, n[k], &args[k]);
            if (st != FCAMD_OK) return st;
        }
        return FCAMD_OK;
    }
    HIP_TRY(hipSetDevice(c->device));
    // laws below this size do not fill the device (512 workgroups per CU at two tiles per wave: fcamd_kernels.hip)
    const int64_t small = (int64_t)c->num_cu * 64 * 64;
    auto batched = [&](int k) { return c->opt.batch_kernel && n[k] > 0 && n[k] < small && models[k]->constraint == FCAMD_FULL; };
    int n_small = 0;
    for (int k = 0; k < count; ++k) n_small += batched(k) ? 1 : 0;
    // the others first, each with its own launch
    for (int k = 0; k < count; ++k) {
        if (n_small && batched(k)) continue;
        const int st = enqueue_ex(models[k], del_t, n[k], &args[k], c->stream, true);
        if (st != FCAMD_OK) return st;
    }
    // tables of at most kBatchMax entries, the Drucker-Prager laws in tables of their own (their batch kernel is cut for 3 waves per
    // SIMD, the others' for 4: fcamd_kernels.hip)
    for (int pass = 0; pass < 2; ++pass)
    for (int k0 = 0; n_small && k0 < count;) {
        const bool dp = pass == 1;
        std::vector<fcamd::BatchEntry>& tab = c->batch_build;
        tab.clear();
        int blocks = 0;
        bool any_counts = false;
        int k = k0;
        for (; k < count && (int)tab.size() < fcamd_context::kBatchMax; ++k) {
            if (!batched(k) || fcamd::batch_law_is_dp(models[k]->law) != dp) continue;
            fcamd_model* m = models[k];
            const fcamd_eval_args* x = &args[k];
            fcamd::BatchEntry e;
            memset(&e, 0, sizeof(e));  // (padding bytes too: the table is compared byte for byte)
            const bool packed = (x->flags & FCAMD_EVAL_PACKED_HISTORY) != 0;
            fcamd::fill_args(m, del_t, n[k], x->grad_del_u, x->stress_prev, x->stress, x->tangent, x->history_prev, x->history, x->parent_rows,
                             reinterpret_cast<unsigned long long*>(x->history_mask), x->flags, x->stress2,
                             reinterpret_cast<unsigned long long*>(x->counters),
                             packed ? reinterpret_cast<const unsigned long long*>(x->packed_mask_prev) : nullptr,
                             packed ? reinterpret_cast<unsigned long long*>(x->packed_mask) : nullptr, e.args);
            e.variant = fcamd::batch_variant_of(m->law, e.args);
            e.first_block = blocks;
            e.main_blocks = n[k] >= 64 ? grid_for(m, n[k]) : 0;
            e.has_tail = (n[k] % 64) != 0 ? 1 : 0;
            e.counts = law_counts(m->law) ? 1 : 0;
            any_counts = any_counts || e.counts;
            blocks += e.main_blocks + e.has_tail;
            tab.push_back(e);
        }
        k0 = k;
        if (tab.empty()) break;
        const size_t bytes = tab.size() * sizeof(fcamd::BatchEntry);
        // a slot that holds this very table already (the Newton iterations of an increment): nothing to upload
        static_assert(sizeof(fcamd::BatchEntry) % 8 == 0, "the table is hashed word by word");
        uint64_t hash = 0x9E3779B97F4A7C15ull ^ bytes;
        const uint64_t* words = reinterpret_cast<const uint64_t*>(tab.data());
        for (size_t i = 0; i < bytes / 8; ++i) hash = (hash ^ words[i]) * 0x100000001B3ull;
        fcamd_context::BatchSlot* slot = nullptr;
        for (auto& sl : c->batch_slots)
            if (sl.host && sl.bytes == bytes && sl.hash == hash && memcmp(sl.host, tab.data(), bytes) == 0) slot = &sl;
        if (!slot) {
            slot = &c->batch_slots[c->batch_next++ % fcamd_context::kBatchSlots];
            if (!slot->host) {
                HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&slot->host), fcamd_context::kBatchMax * sizeof(fcamd::BatchEntry), hipHostMallocDefault));
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&slot->dev), fcamd_context::kBatchMax * sizeof(fcamd::BatchEntry)));
            } else if (slot->bytes) {
                HIP_TRY(hipDeviceSynchronize());  // a launch that still reads the slot's old table may be in flight (rare: all slots in use and a new table)
            }
            slot->bytes = 0;  // the slot advertises the new table only once its upload is queued (a failure below must not leave a match)
            memcpy(slot->host, tab.data(), bytes);
            HIP_TRY(hipMemcpyAsync(slot->dev, slot->host, bytes, hipMemcpyHostToDevice, c->stream));
            if (!slot->uploaded) HIP_TRY(hipEventCreateWithFlags(&slot->uploaded, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(slot->uploaded, c->stream));
            slot->stream = c->stream;
            slot->bytes = bytes;
            slot->hash = hash;
        } else if (slot->stream != c->stream) {
            // the context has been bound to another stream since the table went up: the launch waits for the upload
            HIP_TRY(hipStreamWaitEvent(c->stream, slot->uploaded, 0));
        }
        (void)hipGetLastError();
        HIP_TRY(fcamd::launch_evaluate_batch(slot->dev, (int)tab.size(), blocks, any_counts, dp, c->stream));
    }
    return FCAMD_OK;
}

}  // extern "C"

static int evaluate_wrapped(fcamd_model* m, int wrapper_constraint, double del_t, int64_t n, const double* grad_lo, double* stress_lo,
                            double* tangent_lo, double* stress_3d, double* const* hist, int n_hist) {
    int st = validate_call(m, del_t, n, grad_lo, stress_lo, stress_lo,
                           reinterpret_cast<const void* const*>(hist),
                           reinterpret_cast<const void* const*>(hist), n_hist);
    if (st != FCAMD_OK) return st;
    if (!has_sparse_history(m->law) && m->law != FCAMD_LINEAR_ELASTICITY)
        return fail(FCAMD_ERR_UNSUPPORTED, "the fused 3D wrapper kernel exists for LinearElasticityModel and the plasticity laws");
    if (m->constraint != FCAMD_FULL)
        return fail(FCAMD_ERR_UNSUPPORTED, "the wrapped model must be a FULL (3-D) one");
    const int wrap = wrapper_constraint == FCAMD_UNIAXIAL_STRAIN ? 1 : wrapper_constraint == FCAMD_PLANE_STRAIN ? 2 : 0;
    if (!wrap) return fail(FCAMD_ERR_BAD_ARG, "wrapper constraint must be UNIAXIAL_STRAIN or PLANE_STRAIN");
    if (n > 0 && !stress_3d) return fail(FCAMD_ERR_BAD_ARG, "stress_3d is NULL");
    if (!aligned16(stress_3d) || (wrap == 2 && (!aligned16(grad_lo) || !aligned16(stress_lo) || !aligned16(tangent_lo))))
        return fail(FCAMD_ERR_ALIGN, "device arrays must be 16-byte aligned");
    for (int k = 0; k < m->info.n_hist; ++k)
        if (!aligned16(hist[k])) return fail(FCAMD_ERR_ALIGN, "device history arrays must be 16-byte aligned");
    fcamd_context* c = m->ctx;
    HIP_TRY(hipSetDevice(c->device));
    EvalArgs a;
    a.grad = grad_lo;
    a.stress_in = stress_lo;
    a.stress_out = stress_lo;
    a.stress_out2 = nullptr;
    a.tangent = tangent_lo;
    a.h0_in = a.h0_out = m->info.n_hist > 0 ? hist[0] : nullptr;
    a.h1_in = a.h1_out = m->info.n_hist > 1 ? hist[1] : nullptr;
    a.rows = nullptr;
    a.hmask = nullptr;
    a.cache3d = stress_3d;
    a.n = n;
    a.counters = m->d_counters;
    a.masked_max = m->ctx->opt.masked_max >= 0 ? m->ctx->opt.masked_max : kMaskedRowMaxVonMises;  // eps_n rows of the fused VonMises3D wrapper
    a.flags = 0;
    constants_for(m, del_t);
    a.sc = m->sc;
    a.tb = m->tb;
    if ((st = timing_begin(m)) != FCAMD_OK) return st;
    if (!m->timed && law_counts(m->law)) HIP_TRY(hipMemsetAsync(m->d_counters, 0, kCounterBytes, c->stream));
    if (n > 0) {
        (void)hipGetLastError();  // as in enqueue()
        HIP_TRY(launch_evaluate_wrapped(m->law, wrap, a, grid_for(m, n), c->stream));
    }
    return timing_end(m);
}

extern "C" {

int fcamd_model_last_stats(fcamd_model* m, fcamd_stats* stats) {
    if (!m || !stats) return fail(FCAMD_ERR_BAD_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(m->ctx->device));
    const int st = read_stats(m, m->ctx->stream, stats);
    if (st != FCAMD_OK) return st;
    stats->kernel_ms = -1.0;
    if (m->host_ms >= 0.0f) {  // the last entry was a (synchronous) host entry: its wall-clock
        stats->kernel_ms = m->host_ms;
    } else if (m->timed) {
        float ms = 0.0f;
        HIP_TRY(hipEventSynchronize(m->ev1));
        HIP_TRY(hipEventElapsedTime(&ms, m->ev0, m->ev1));
        stats->kernel_ms = ms;
    }
    return FCAMD_OK;
}

int fcamd_strain_from_grad_u_device(fcamd_context* c, int64_t n, const double* grad_u,
                                    double* strain, int rust_factor) {
    if (!c) return fail(FCAMD_ERR_BAD_ARG, "context is NULL");
    if (n < 0) return fail(FCAMD_ERR_SIZE, "negative number of quadrature points");
    if (n == 0) return FCAMD_OK;
    if (!grad_u || !strain) return fail(FCAMD_ERR_BAD_ARG, "NULL array");
    if (!aligned16(grad_u) || !aligned16(strain))
        return fail(FCAMD_ERR_ALIGN, "device arrays must be 16-byte aligned");
    HIP_TRY(hipSetDevice(c->device));
    int grid = c->grid_override > 0 ? c->grid_override : c->num_cu * 8;
    const int64_t need = ((n + 63) / 64 + 3) / 4;
    if (need < grid) grid = (int)std::max<int64_t>(need, 1);
    (void)hipGetLastError();  // as in enqueue()
    HIP_TRY(launch_strain(grad_u, strain, n, rust_factor ? kFactorRs : kFactorPy, grid, c->stream));
    return FCAMD_OK;
}

int fcamd_convert_device(fcamd_context* c, int kind, int64_t n, const double* src, double* dst) {
    if (!c) return fail(FCAMD_ERR_BAD_ARG, "context is NULL");
    if (n < 0) return fail(FCAMD_ERR_SIZE, "negative number of quadrature points");
    if (n == 0) return FCAMD_OK;
    if (!src || !dst) return fail(FCAMD_ERR_BAD_ARG, "NULL array");
    CopyMap m{};
    switch (kind) {
        case FCAMD_GRAD_1D_TO_3D: m = {1, 1, 9, {0}, {0}}; break;
        case FCAMD_STRESS_1D_TO_3D: m = {1, 1, 6, {0}, {0}}; break;
        case FCAMD_STRESS_3D_TO_1D: m = {1, 6, 1, {0}, {0}}; break;
        case FCAMD_TANGENT_3D_TO_1D: m = {1, 36, 1, {0}, {0}}; break;
        case FCAMD_GRAD_2D_TO_3D: m = {4, 4, 9, {0, 1, 2, 3}, {0, 1, 3, 4}}; break;
        case FCAMD_STRESS_2D_TO_3D: m = {4, 4, 6, {0, 1, 2, 3}, {0, 1, 2, 3}}; break;
        case FCAMD_STRESS_3D_TO_2D: m = {4, 6, 4, {0, 1, 2, 3}, {0, 1, 2, 3}}; break;
        case FCAMD_TANGENT_3D_TO_2D:
            m.K = 16;
            m.in_stride = 36;
            m.out_stride = 16;
            for (int r = 0; r < 4; ++r)
                for (int cc = 0; cc < 4; ++cc) {
                    m.imap[4 * r + cc] = 6 * r + cc;
                    m.omap[4 * r + cc] = 4 * r + cc;
                }
            break;
        default: return fail(FCAMD_ERR_BAD_ARG, "unknown conversion kind %d", kind);
    }
    HIP_TRY(hipSetDevice(c->device));
    (void)hipGetLastError();  // as in enqueue()
    HIP_TRY(launch_strided_copy(src, dst, n, m, c->stream));
    return FCAMD_OK;
}

int fcamd_map_rows_device(fcamd_context* c, int64_t n_rows, int row_size, const double* src,
                          const int32_t* src_idx, double* dst, const int32_t* dst_idx) {
    if (!c) return fail(FCAMD_ERR_BAD_ARG, "context is NULL");
    if (n_rows < 0 || row_size <= 0) return fail(FCAMD_ERR_SIZE, "bad row count / row size");
    if (n_rows == 0) return FCAMD_OK;
    if (!src || !dst) return fail(FCAMD_ERR_BAD_ARG, "NULL array");
    HIP_TRY(hipSetDevice(c->device));
    (void)hipGetLastError();  // as in enqueue()
    HIP_TRY(launch_map_rows(src, src_idx, dst, dst_idx, n_rows, row_size, c->stream));
    return FCAMD_OK;
}

}  // extern "C"
