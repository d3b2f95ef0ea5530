// comfe-rs MisesPlasticity3D (comfe-rs/src/mises_plasticity.rs:58-126): point function, the 3-D tile, the fused wrapper tile.
// Part of the device code of libfcamd (translation unit: ../fcamd_kernels.hip, which holds the kernels and launchers).
#pragma once
#include "tile_io.h"
#include "tangent_writers.h"
#include "wrapped_io.h"
#include "history_rows.h"

namespace fcamd {

// comfe-rs MisesPlasticity3D (mises_plasticity.rs:58-126): the whole update of one point.  In: e, s (sigma_n),
// h = [alpha, eps_p(6)].  Out: s (total stress), h (updated if plastic), tangent parameters B, sc2 and the
// (non-unit) flow direction nv.  Returns whether the point is plastic.
__device__ __forceinline__ bool cm_point(ScalarsRef sc, bool live, const double (&e)[6], double (&s)[6], double (&h)[7],
                                         double& B, double& sc2, double (&nv)[6]) {
    const double kappa = sc.s[2], y_0 = sc.s[3], hh = sc.s[4], two_mu = sc.s[5], den = sc.s[6], s32 = sc.s[7],
                 three_mu = sc.s[8], hfac = sc.s[9];
    const double alpha = h[0];
    // (p_0, s_0) = vol_dev(sigma) ; (tr, dev) = trace_dev(d_eps)
    const double p_0 = ((s[0] + s[1]) + s[2]) / 3.0;
    const double eps_trace = (e[0] + e[1]) + e[2];
    const double eps_vol = eps_trace / 3.0;
    const double p_1 = p_0 + kappa * eps_trace;
    double s_tr[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double s0 = i < 3 ? s[i] + (-p_0) : s[i];
        const double ed = i < 3 ? e[i] + (-eps_vol) : e[i];
        s_tr[i] = s0 + two_mu * ed;
    }
    // mises_norm(): deviator once more, sqrt(3 * (0.5 * |dev|^2)), sequential sum
    const double v3 = ((s_tr[0] + s_tr[1]) + s_tr[2]) / 3.0;
    double n2 = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double d = i < 3 ? s_tr[i] + (-v3) : s_tr[i];
        n2 = i == 0 ? d * d : n2 + d * d;
    }
    const double q = sqrt(3.0 * (0.5 * n2));
    const double sigma_y = y_0 + hh * alpha;
    const bool plastic = live && !(q < sigma_y);  // strict "<" elastic test (:95)

    double theta = 1.0;
    sc2 = 0.0;  // 2 mu theta_bar
#pragma unroll
    for (int i = 0; i < 6; ++i) nv[i] = 0.0;
    if (plastic) {
        const double del_alpha = (q - sigma_y) / den;
        const double del_gamma = s32 * del_alpha;
        theta = 1.0 - (three_mu * del_alpha) / q;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            nv[i] = s_tr[i] / q;
            h[1 + i] = h[1 + i] + del_gamma * nv[i];
        }
        h[0] = alpha + del_alpha;
        const double theta_bar = hfac - (1.0 - theta);
        sc2 = two_mu * theta_bar;
    }
    // total (not incremental) stress:  p_1 1 + theta s_tr   (elastic: theta == 1 exactly)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double ts = theta * s_tr[i];
        s[i] = i < 3 ? p_1 + ts : ts;
    }
    B = plastic ? two_mu * theta : two_mu;
    return plastic;
}

// --- comfe-rs MisesPlasticity3D: linear hardening, closed-form radial return ---------------
// scalars: s[0]=strain factor (FRAC_1_SQRT_2), s[1]=mu, s[2]=kappa, s[3]=y_0, s[4]=h,
//          s[5]=2*mu, s[6]=3*mu+h, s[7]=sqrt(3/2), s[8]=3*mu, s[9]=1/(1+h/(3 mu))
// tables:  a = kappa*sym_id(x)sym_id, b = P_dev.   history field 0: [alpha, eps_p(6)] per point.
template <bool IDX, bool FULL, bool NT, int PM = 0>
__device__ __forceinline__ void tile_comfe_mises(ArgsRef a, const StressBases& sb, const Tables* T, double* region,
                                                 int* rows_lds, long long p0, int npts, int lane,
                                                 WaveStats& st) {
    const SparseWords w = sparse_words(a, p0);  // first: they arrive with the gradient
    Chunks<9> cg;
    StressRows<IDX, FULL, NT> sr;
    Chunks<7> ch;
    const bool split = (a.flags & kFlagSplitHistory) != 0;
    const bool live = FULL || lane < npts;
    tile_load<9, FULL, NT>(cg, a.grad + p0 * 9, npts * 9, lane);
    sr.load(a, sb, p0, npts, lane, rows_lds);
    double alpha_n = 0.0;
    if (split)
        alpha_n = live ? a.h0_in[p0 + lane] : 0.0;
    else
        tile_load<7, FULL, NT>(ch, a.h0_in + p0 * 7, npts * 7, lane);
    const bool hist_in_place = (a.h0_in == a.h0_out);

    double g[9], s[6], h[7], e[6];
    transpose_in<9>(cg, region, lane, g);
    sr.get(region, lane, s);
    if (split) {  // eps_p only accumulates: start the rows at zero, what comes back is the increment
        h[0] = alpha_n;
#pragma unroll
        for (int i = 1; i < 7; ++i) h[i] = 0.0;
    } else {
        transpose_in<7>(ch, region, lane, h);
    }
    mandel_strain(g, a.sc.s[0], e);

    double B, sc2, nv[6];
    const bool plastic = cm_point(a.sc, live, e, s, h, B, sc2, nv);
    const unsigned long long mask = __ballot(plastic);
    st.plastic += (lane == 0) ? (unsigned long long)__popcll(mask) : 0ull;
    const unsigned long long touched = sparse_touched(a, w, mask);

    // split layout: the committed eps_p rows are requested and taken into the lanes BEFORE the stress store is issued
    // (history_rows.h: SplitRows)
    SplitRows<FULL, NT> hr;
    double d6[6] = {h[1], h[2], h[3], h[4], h[5], h[6]};
    if (split) {
        hr.request(a, w, p0, npts, lane, touched, hist_in_place, region);
        hr.gather(region, lane, mask, d6);
    }
    sr.put(sb, region, lane, s, p0, npts, rows_lds);
    if (split) {
        hr.store(a, p0, npts, lane, mask, hist_in_place, region, h[0], d6);
    } else {
        history7_store<FULL, NT>(a, p0, npts, lane, touched, hist_in_place, region, h);
    }
    sparse_record(a, w, p0, mask, lane);

    const unsigned long long tneed = sparse_tangent_need<FULL>(a, touched);
    if (sb.tan && tneed != 0ull) {
        publish_tangent_params(region, lane, B, sc2, nv);
        wave_sync();
        if (tangent_params_mode<PM>(a))  // the host rebuilds the rows (fcamd_hosttangent.cpp)
            store_tangent_params<FULL, NT>(a, region, sb.tan, p0, npts, lane, mask);
        else
            tangent_mises<true, IDX, FULL, NT>(region, T->a, T->b, sb.tan, p0, rows_lds, npts, lane, tneed, (a.flags & kFlagExactTangentRows) != 0);
        wave_sync();
    }
}

template <int WRAP, bool FULL, bool NT>
__device__ __forceinline__ void tile_comfe_mises_wrapped(ArgsRef a, const Tables* T, double* region,
                                                         long long p0, int npts, int lane, WaveStats& st) {
    const bool live = FULL || lane < npts;
    Chunks<7> ch;
    tile_load<7, FULL, NT>(ch, a.h0_in + p0 * 7, npts * 7, lane);
    double g[9], s[6], h[7], e[6];
    wrapped_load<WRAP, FULL, NT>(a, region, p0, npts, lane, g, s);
    transpose_in<7>(ch, region, lane, h);
    mandel_strain(g, a.sc.s[0], e);

    double B, sc2, nv[6];
    const bool plastic = cm_point(a.sc, live, e, s, h, B, sc2, nv);
    const unsigned long long mask = __ballot(plastic);
    st.plastic += (lane == 0) ? (unsigned long long)__popcll(mask) : 0ull;
    wrapped_store_stress<WRAP, FULL, NT>(a, region, p0, npts, lane, s);
    if (mask != 0ull) transpose_out<7, FULL, NT>(h, region, lane, a.h0_out + p0 * 7, npts * 7);
    if (a.tangent) wrapped_tangent_mises<true, WRAP, FULL, NT>(a, T, region, p0, npts, lane, B, sc2, nv);
}

}  // namespace fcamd
