// comfe-rs general return mapping with the Drucker-Prager surfaces (plasticity/general.rs:105-266, drucker_prager_classic.rs,
// drucker_prager_hyperbolic.rs) in invariant coordinates.
// Part of the device code of libfcamd (translation unit: ../fcamd_kernels.hip, which holds the kernels and launchers).
#pragma once
#include "tile_io.h"
#include "tangent_writers.h"
#include "wrapped_io.h"
#include "history_rows.h"

namespace fcamd {

// --- comfe-rs general return mapping with the Drucker-Prager yield surfaces -------------------
// Reference: comfe-rs/src/plasticity/general.rs:105-266 (Newton on sigma(6), lambda, kappa; maxit
// 25, atol = rtol = 1e-8; consistent tangent = (last Jacobian)^-1 [0:6,0:6] . E),
// drucker_prager_classic.rs:62-116, drucker_prager_hyperbolic.rs:64-114.
//
// The same Newton iteration in invariant coordinates.  Both surfaces are isotropic: g and df/dsigma
// lie in span{1, s}, the Jacobian block I + 2 mu dl (c2 s s^T + c1 P_dev) maps that plane to itself,
// and the iteration starts at sigma_tr -- so every iterate is sigma = (I1/3) 1 + rho s_tr and the
// reference's 8 unknowns collapse, step for step, to (I1, rho, lambda, kappa):
//     vol :  dv + 3 kappa b_flow dlam                   = rv      (sigma residual = rv 1 + rd s_tr)
//     dev :  A_d dd + 2 mu c1 rho dlam                  = rd      A_d = 1 + w (c1 + c2 rho^2 |s_tr|^2)
//     f   :  3 b dv + c1 rho |s_tr|^2 dd                = f
//     kap :  dkap = res_k + dl (dk/dsigma . dsigma) + k dlam      (kappa column of rows 0..6 is zero)
// with w = 2 mu dl.  Convergence tests use the same norms (|a 1 + c s_tr|^2 = 3 a^2 + c^2 |s_tr|^2).
// The inverse of the bordered Jacobian is closed-form (Sherman-Morrison on the deviatoric block, Schur
// complement for the f row), which makes the tangent a five-term isotropic form
//     T = t11 1x1 + tP P_dev + tss s x s + t1s 1 x s + ts1 s x 1         (not symmetric if b != b_flow)
// that the tile writes exactly like the Mises tangents: 11 doubles per point through LDS, every
// lane rebuilding the two entries of the 16-byte chunk it stores.  The law is HBM-bound like the
// others.  Quirk kept as read: the kappa residual carries no del_lambda (general.rs:222).
// scalars: s[0]=strain factor, s[1]=mu, s[2]=kappa, s[3]=a, s[4]=b, s[5]=b_flow, s[6]=d*d,
//          s[7]=2*mu, s[8]=sqrt(2/3), s[9]=1/(4 mu), s[10]=1/(9 kappa)
// tables:  a = sym_id (x) sym_id, b = P_dev, c = E
constexpr int kDpStride = 14;  // doubles per point of the published tangent parameters (conflict-free b128)

struct DPInv {  // model state at (I1, rho)
    double f, c1, c2, root;
};

template <bool HYPER>
__device__ __forceinline__ DPInv dp_state(double I1, double rho, double n2, double a_, double b, double dsq,
                                          bool& tip) {
    DPInv m;
    const double j_2 = 0.5 * (rho * rho) * n2;
    if constexpr (HYPER) {
        m.root = sqrt(j_2 + dsq);
        m.c1 = 0.5 * (1.0 / m.root);
        m.c2 = -0.25 / ((j_2 + dsq) * m.root);
    } else {
        tip = tip || !(I1 < a_ / b);
        m.root = sqrt(j_2);
        m.c1 = 0.5 / m.root;
        m.c2 = -0.25 / (j_2 * m.root);
    }
    m.f = m.root + b * I1 - a_;
    return m;
}

// T[i][j..j+1] for chunk r (entries [i][j], [i][j + 1], j = 2 jj) of point p from the published parameters
__device__ __forceinline__ d2 tangent_dp_chunk(const double* tp, const double* t11tab, const double* pdtab, const double* etab,
                                               int p, int r, int i, int jj) {
    const double* t = tp + kDpStride * p;
    const d2 c0 = reinterpret_cast<const d2*>(t)[0];  // t11, tP
    const d2 c1 = reinterpret_cast<const d2*>(t)[1];  // tss, t1s
    const d2 c2 = reinterpret_cast<const d2*>(t)[2];  // ts1, plastic flag
    const double ts1 = c2.x;
    const double si = t[6 + i];
    const d2 sj = *reinterpret_cast<const d2*>(t + 6 + 2 * jj);
    const d2 o = *reinterpret_cast<const d2*>(t11tab + 2 * r);  // (1 x 1)[i][j]: 6 i + j = 2 r
    const d2 pd = *reinterpret_cast<const d2*>(pdtab + 2 * r);
    const double oi = i < 3 ? 1.0 : 0.0;
    d2 v;
    v.x = (c0.x * o.x + c0.y * pd.x) + ((c1.x * si) * sj.x + (c1.y * oi) * sj.x + (ts1 * si) * (jj < 2 ? 1.0 : 0.0));   // j < 3
    v.y = (c0.x * o.y + c0.y * pd.y) + ((c1.x * si) * sj.y + (c1.y * oi) * sj.y + (ts1 * si) * (jj < 1 ? 1.0 : 0.0));   // j + 1 < 3
    // elastic points of a mixed tile: the reference returns elastic_tangent() itself (general.rs:131-135),
    // i.e. the host-computed 2 mu P_dev + 3 kappa P_vol bit for bit, not kappa 1x1 + 2 mu P_dev
    const d2 el = *reinterpret_cast<const d2*>(etab + 2 * r);
    if (c2.y == 0.0) v = el;
    return v;
}

template <bool NT, bool MASKED, int K>
__device__ __forceinline__ void tangent_dp_pass(const double* tp, const double* t11tab, const double* pdtab, const double* etab,
                                                double* tile, int lane, ChunkLane& cl, unsigned long long tneed, bool exact_rows) {
    // the maps of pass k and k + 9 coincide up to a constant ((10 k) % 18 has period 9): left alone the compiler keeps them alive
    // across nine passes, and this kernel sits at its 168-VGPR cap (scratch: 9.0 -> 9.8 ms).  Fresh values per group of passes.
    if constexpr (K % kTangentGroup == 0) asm volatile("" : "+v"(cl.pl), "+v"(cl.r0));
    const ChunkMap m = chunk_map<K>(cl);
    bool wanted = true;
    if constexpr (MASKED) wanted = tangent_chunk_wanted(tneed, m.p, exact_rows);
    char* dst = reinterpret_cast<char*>(tile) + K * (kWave * 16) + (unsigned)lane * 16u;  // scalar base of the pass + the lane's byte offset
    if (wanted) store_tangent16<NT>(reinterpret_cast<double*>(dst), tangent_dp_chunk(tp, t11tab, pdtab, etab, m.p, m.r, m.i, m.jj));
    if constexpr (K % kTangentGroup == kTangentGroup - 1) __builtin_amdgcn_sched_barrier(0);
}

template <bool NT, bool MASKED, int... K>
__device__ __forceinline__ void tangent_dp_passes(const double* tp, const double* t11tab, const double* pdtab, const double* etab,
                                                  double* tile, int lane, unsigned long long tneed, bool exact_rows, std::integer_sequence<int, K...>) {
    ChunkLane cl = chunk_lane(lane);
    (tangent_dp_pass<NT, MASKED, K>(tp, t11tab, pdtab, etab, tile, lane, cl, tneed, exact_rows), ...);
}

template <bool IDX, bool FULL, bool NT>
__device__ __forceinline__ void tangent_dp(const double* tp, const double* t11tab, const double* pdtab,
                                           const double* etab, double* tangent, long long p0,
                                           const int* rows_lds, int npts, int lane, unsigned long long tneed, bool exact_rows = false) {
    if constexpr (FULL && !IDX) {  // the contiguous tile: incremental chunk maps (tangent_writers.h: chunk_map), need test only when sparse
        double* tile = tangent + p0 * 36;
        if (tneed == ~0ull)
            tangent_dp_passes<NT, false>(tp, t11tab, pdtab, etab, tile, lane, tneed, false, std::make_integer_sequence<int, 18>{});
        else
            tangent_dp_passes<NT, true>(tp, t11tab, pdtab, etab, tile, lane, tneed, exact_rows, std::make_integer_sequence<int, 18>{});
        return;
    }
    const int nchunks = npts * 18;
#pragma unroll
    for (int k = 0; k < 18; ++k) {
        const int q = k * kWave + lane;
        const int p = q / 18;
        const int r = q - 18 * p;
        const int i = r / 3;
        const d2 v = tangent_dp_chunk(tp, t11tab, pdtab, etab, p, r, i, r - 3 * i);
        if ((FULL || q < nchunks) && ((tneed >> p) & 1ull)) store_tangent16<NT>(tangent_chunk<IDX>(tangent, p0, q, rows_lds), v);
        if (k % kTangentGroup == kTangentGroup - 1) __builtin_amdgcn_sched_barrier(0);
    }
}

// trial state of one point: sigma_tr = E d_eps + sigma_0 = (I1_tr/3) 1 + s_tr   (E v = 2 mu dev v + kappa tr(v) 1)
struct DPTrial {
    double sig1[6], s_tr[6], I1_tr, n2;
    DPInv m;
    bool tip;
};

template <bool HYPER>
__device__ __forceinline__ void dp_trial(ScalarsRef sc, const double (&e)[6], const double (&sig0)[6], DPTrial& t) {
    const double kappa = sc.s[2], a_ = sc.s[3], b = sc.s[4], dsq = sc.s[6], two_mu = sc.s[7];
    {
        const double tr = (e[0] + e[1]) + e[2], vol = tr / 3.0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            t.sig1[i] = (i < 3 ? two_mu * (e[i] + (-vol)) + kappa * tr : two_mu * e[i]) + sig0[i];
    }
    t.I1_tr = (t.sig1[0] + t.sig1[1]) + t.sig1[2];
    t.n2 = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        t.s_tr[i] = i < 3 ? t.sig1[i] + (-(t.I1_tr / 3.0)) : t.sig1[i];
        t.n2 = i == 0 ? t.s_tr[0] * t.s_tr[0] : t.n2 + t.s_tr[i] * t.s_tr[i];
    }
    t.tip = false;
    t.m = dp_state<HYPER>(t.I1_tr, 1.0, t.n2, a_, b, dsq, t.tip);
}

// coefficients of the five-term tangent and the scale of the deviator (sigma = (I1/3) 1 + rho s_tr);
// the defaults are the elastic point: T = E = kappa 1x1 + 2 mu P_dev
struct DPTangent {
    double t11, tP, tss = 0.0, t1s = 0.0, ts1 = 0.0, rho = 1.0;
};

// return mapping of one plastic point: Newton in invariant coordinates, converged stress in t.sig1,
// history h = [alpha, plastic_strain(6)] updated, tangent coefficients in tg
template <bool HYPER>
__device__ __forceinline__ void dp_return(ScalarsRef sc, const double (&e)[6], const double (&sig0)[6], DPTrial& t,
                                          double (&h)[7], DPTangent& tg, WaveStats& st) {
    const double kappa = sc.s[2], a_ = sc.s[3], b = sc.s[4], bflow = sc.s[5], dsq = sc.s[6], two_mu = sc.s[7],
                 s23 = sc.s[8], inv4mu = sc.s[9], inv9k = sc.s[10];
    const double I1_tr = t.I1_tr, n2 = t.n2;
    DPInv m = t.m;
    double rho = 1.0;
    const double alpha_0 = h[0];
    double I1 = I1_tr, dl = 0.0, alpha_1 = alpha_0;
    double rv = 0.0, rd = 0.0, rf = m.f, rk = 0.0;
    int it = 0;
    bool failed = false;
    for (;;) {
        // Newton step with the Jacobian of the current state (m, rho, dl)
        const double w = two_mu * dl;
        const double Ad = 1.0 + w * (m.c1 + m.c2 * (rho * rho) * n2);
        const double gn2 = 3.0 * (bflow * bflow) + (m.c1 * m.c1) * (rho * rho) * n2;
        const double gnorm = sqrt(gn2), kk = s23 * gnorm;
        const double cr = m.c1 * rho;  // coefficient of s_tr in g and df/dsigma
        const double dlam = ((3.0 * b) * rv + (cr * n2) * (rd / Ad) - rf) /
                            ((9.0 * kappa) * (b * bflow) + two_mu * (cr * cr) * n2 / Ad);
        const double dv = rv - (3.0 * kappa * bflow) * dlam;
        const double dd = (rd - (two_mu * cr) * dlam) / Ad;
        const double dkds = (s23 / gnorm) * m.c1 * (m.c1 + m.c2 * (rho * rho) * n2) * rho * n2 * dd;
        const double dkap = rk + dl * dkds + kk * dlam;
        const double I1_prev = I1, rho_prev = rho, dl_prev = dl, al_prev = alpha_1;
        I1 = I1 - 3.0 * dv;
        rho = rho - dd;
        dl = dl - dlam;
        alpha_1 = alpha_1 - dkap;
        m = dp_state<HYPER>(I1, rho, n2, a_, b, dsq, t.tip);
        // residuals at the new state
        const double gn2n = 3.0 * (bflow * bflow) + (m.c1 * m.c1) * (rho * rho) * n2;
        rv = (I1 - I1_tr) / 3.0 + dl * (3.0 * kappa * bflow);
        rd = (rho - 1.0) + dl * (two_mu * m.c1) * rho;
        rf = m.f;
        rk = (alpha_1 - alpha_0) - s23 * sqrt(gn2n);
        const double atol = 1e-8, rtol = 1e-8;
        const double dI = (I1 - I1_prev) / 3.0, dr = rho - rho_prev;
        const bool conv_res = sqrt(3.0 * rv * rv + rd * rd * n2) < atol && fabs(rk) < atol && fabs(rf) < atol;
        const bool conv_inc = sqrt(3.0 * dI * dI + dr * dr * n2) < atol + rtol * sqrt(I1 * I1 / 3.0 + rho * rho * n2) &&
                              fabs(alpha_1 - al_prev) < atol + rtol * fabs(alpha_1) &&
                              fabs(dl - dl_prev) < atol + rtol * fabs(dl);
        if (conv_res || conv_inc) break;
        if (it > 25) {
            failed = true;
            break;
        }
        ++it;
    }
    st.iters += (unsigned long long)(it + 1);
    st.nonconv += failed ? 1ull : 0ull;
    // converged stress, history
#pragma unroll
    for (int i = 0; i < 6; ++i) t.sig1[i] = (i < 3 ? I1 / 3.0 : 0.0) + rho * t.s_tr[i];
    h[0] = alpha_1;
    {
        double ds[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) ds[i] = t.sig1[i] - sig0[i];
        const double tr = (ds[0] + ds[1]) + ds[2], vol = tr / 3.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            // plastic_strain += d_eps - E^-1 (sigma_1 - sigma_0),  E^-1 = isotropic_elastic_tangent(1/(4 mu), 1/(9 kappa))
            const double einv = i < 3 ? (2.0 * inv4mu) * (ds[i] + (-vol)) + (3.0 * inv9k) * vol : (2.0 * inv4mu) * ds[i];
            h[1 + i] = h[1 + i] + (e[i] - einv);
        }
    }
    // tangent from the inverse of the Jacobian at the final state (s = rho s_tr)
    {
        const double w = two_mu * dl, s2 = (rho * rho) * n2;
        const double Ad = 1.0 + w * (m.c1 + m.c2 * s2);
        const double alpha_d = 1.0 / (1.0 + w * m.c1);
        const double beta = alpha_d * w * m.c2 / Ad;
        const double uv = 3.0 * kappa * bflow, ud = two_mu * m.c1 / Ad;  // A^-1 E g       = uv 1 + ud s
        const double vv = b, vd = m.c1 / Ad;                              // df/dsigma A^-1 = vv 1^T + vd s^T
        const double D = 3.0 * vv * uv + vd * s2 * (two_mu * m.c1);
        tg.t11 = kappa - 3.0 * kappa * uv * vv / D;
        tg.tP = two_mu * alpha_d;
        tg.tss = -two_mu * beta - two_mu * ud * vd / D;
        tg.t1s = -two_mu * uv * vd / D;
        tg.ts1 = -3.0 * kappa * ud * vv / D;
        tg.rho = rho;
    }
}

// this lane's 11 tangent parameters (+ plastic flag) into the wave's LDS region, stride kDpStride
__device__ __forceinline__ void dp_publish(double* region, int lane, const DPTangent& tg, const double (&s_tr)[6],
                                           bool plastic) {
    double* t = region + kDpStride * lane;
    d2 v;
    v.x = tg.t11, v.y = tg.tP;
    reinterpret_cast<d2*>(t)[0] = v;
    v.x = tg.tss, v.y = tg.t1s;
    reinterpret_cast<d2*>(t)[1] = v;
    v.x = tg.ts1, v.y = plastic ? 1.0 : 0.0;
    reinterpret_cast<d2*>(t)[2] = v;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        v.x = tg.rho * s_tr[2 * i];
        v.y = tg.rho * s_tr[2 * i + 1];
        reinterpret_cast<d2*>(t)[3 + i] = v;
    }
}

// kFlagTangentParams (host tangent, fcamd_hosttangent.cpp) for the Drucker-Prager laws: the plastic points of a tile leave as the first 12
// doubles of their published record -- t11, tP, tss, t1s, ts1, the plastic flag, rho s_tr[6]: 96 bytes per point -- and the tile's
// plastic ballot as one word behind the launch's 12 * roundup(n, 64) parameter doubles; elastic points get E on the host.
constexpr int kDpParamDoubles = 12;
template <bool FULL, bool NT>
__device__ __forceinline__ void store_dp_params(ArgsRef a, const double* tp, double* params, long long p0, int npts, int lane,
                                                unsigned long long plastic, bool published) {
    double* tile = params + p0 * kDpParamDoubles;
    if (published) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int q = k * kWave + lane;  // chunk q of the tile's parameter image: point q / 6, 16-byte part q % 6
            const int pt = (int)(__umul24((unsigned)q, 10923u) >> 16);  // q / 6 for q < 384
            const d2 v = *reinterpret_cast<const d2*>(tp + kDpStride * pt + 2 * (q - 6 * pt));
            if ((FULL || q < 6 * npts) && ((plastic >> pt) & 1ull) != 0ull) store16<NT>(tile + 2 * q, v);
        }
    }
    if (lane == 0) reinterpret_cast<unsigned long long*>(params + kDpParamDoubles * ((a.n + 63) & ~63ll))[p0 >> 6] = plastic;
}

// PM: the tangent leaves as parameters (0 never, 1 always, 2 by kFlagTangentParams at run time; fcamd_kernels.hip: run_tile)
template <bool HYPER, bool IDX, bool FULL, bool NT, int PM = 0>
__device__ __forceinline__ void tile_comfe_dp(ArgsRef a, const StressBases& sb, const Tables* T,
                                              double* region, int* rows_lds, long long p0, int npts, int lane,
                                              int r0, WaveStats& st) {
    SparseWords w = sparse_words(a, p0);  // first: they arrive with the gradient
    Chunks<9> cg;
    StressRows<IDX, FULL, NT> sr;
    Chunks<7> ch;
    const bool split = (a.flags & kFlagSplitHistory) != 0;
    const bool live = FULL || lane < npts;
    tile_load<9, FULL, NT>(cg, a.grad + p0 * 9, npts * 9, lane);
    sr.load(a, sb, p0, npts, lane, rows_lds);
    double scalar_n = 0.0;
    if (split)
        scalar_n = live ? a.h0_in[p0 + lane] : 0.0;
    else
        tile_load<7, FULL, NT>(ch, a.h0_in + p0 * 7, npts * 7, lane);
    const bool hist_in_place = (a.h0_in == a.h0_out);

    double g9[9], sig0[6], h[7], e[6];
    transpose_in<9>(cg, region, lane, g9);
    sparse_words_uniform(w);  // arrived with the gradient: two SGPR pairs through the return mapping instead of VGPRs
    sr.get(region, lane, sig0);
    if (split) {  // the plastic strain only accumulates: start the rows at zero, what comes back is the increment
        h[0] = scalar_n;
#pragma unroll
        for (int i = 1; i < 7; ++i) h[i] = 0.0;
    } else {
        transpose_in<7>(ch, region, lane, h);
    }
    mandel_strain(g9, a.sc.s[0], e);

    DPTrial t;
    dp_trial<HYPER>(a.sc, e, sig0, t);
    const bool plastic = live && (t.m.f > 0.0);
    const unsigned long long mask = __ballot(plastic);

    const unsigned long long touched = sparse_touched(a, w, mask);
    SplitRows<FULL, NT> hr;
    if (mask == 0ull) {
        // fully elastic tile: stress = sigma_tr, tangent = E, history untouched (sparse protocol: stale rows restored)
        double d6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (split) {
            hr.request(a, w, p0, npts, lane, touched, hist_in_place, region);
            hr.gather(region, lane, 0ull, d6);
        }
        sr.put(sb, region, lane, t.sig1, p0, npts, rows_lds);
        if (split) {
            hr.store(a, p0, npts, lane, 0ull, hist_in_place, region, h[0], d6);
        } else {
            history7_store<FULL, NT>(a, p0, npts, lane, touched, hist_in_place, region, h);
        }
        sparse_record(a, w, p0, 0ull, lane);
        const unsigned long long tneed = sparse_tangent_need<FULL>(a, touched);
        if (sb.tan && tneed != 0ull) {
            if (tangent_params_mode<PM>(a)) {  // fully elastic tile: the ballot word alone (the host writes E into every row)
                store_dp_params<FULL, NT>(a, region, sb.tan, p0, npts, lane, 0ull, false);
                st.domain += (live && t.tip) ? 1ull : 0ull;
                return;
            }
            if constexpr (IDX) wave_sync();
            if (tneed == ~0ull)
                tangent_const<IDX, FULL, NT>(T->c, sb.tan, p0, rows_lds, npts, lane, r0);
            else
                tangent_const_masked<IDX, FULL, NT>(T->c, sb.tan, p0, rows_lds, npts, lane, tneed, (a.flags & kFlagExactTangentRows) != 0);
        }
        st.domain += (live && t.tip) ? 1ull : 0ull;
        return;
    }

    DPTangent tg;
    tg.t11 = a.sc.s[2], tg.tP = a.sc.s[7];
    if (plastic) dp_return<HYPER>(a.sc, e, sig0, t, h, tg, st);
    st.plastic += (lane == 0) ? (unsigned long long)__popcll(mask) : 0ull;
    st.domain += (live && t.tip) ? 1ull : 0ull;  // tip of the classic surface reached (reference: assert!)

    // the committed rows: requested after the return mapping (whose registers they would otherwise share: spills at the
    // 168-VGPR cap of the indexed kernels) and taken into the lanes before the first store of the tile
    double d6[6] = {h[1], h[2], h[3], h[4], h[5], h[6]};
    if (split) {
        hr.request(a, w, p0, npts, lane, touched, hist_in_place, region);
        hr.gather(region, lane, mask, d6);
    }
    sr.put(sb, region, lane, t.sig1, p0, npts, rows_lds);
    if (split) {
        hr.store(a, p0, npts, lane, mask, hist_in_place, region, h[0], d6);
    } else {
        history7_store<FULL, NT>(a, p0, npts, lane, touched, hist_in_place, region, h);
    }
    sparse_record(a, w, p0, mask, lane);

    const unsigned long long tneed = sparse_tangent_need<FULL>(a, touched);
    if (sb.tan && tneed != 0ull) {
        dp_publish(region, lane, tg, t.s_tr, plastic);
        wave_sync();
        if (tangent_params_mode<PM>(a))
            store_dp_params<FULL, NT>(a, region, sb.tan, p0, npts, lane, mask, true);
        else
            tangent_dp<IDX, FULL, NT>(region, T->a, T->b, T->c, sb.tan, p0, rows_lds, npts, lane, tneed, (a.flags & kFlagExactTangentRows) != 0);
        wave_sync();
    }
}

// fused 3D -> 1D/2D wrapper around the Drucker-Prager laws (see the Mises versions above)
template <bool HYPER, int WRAP, bool FULL, bool NT>
__device__ __forceinline__ void tile_comfe_dp_wrapped(ArgsRef a, const Tables* T, double* region,
                                                      long long p0, int npts, int lane, WaveStats& st) {
    const bool live = FULL || lane < npts;
    Chunks<7> ch;
    tile_load<7, FULL, NT>(ch, a.h0_in + p0 * 7, npts * 7, lane);
    double g[9], sig0[6], h[7], e[6];
    wrapped_load<WRAP, FULL, NT>(a, region, p0, npts, lane, g, sig0);
    transpose_in<7>(ch, region, lane, h);
    mandel_strain(g, a.sc.s[0], e);

    DPTrial t;
    dp_trial<HYPER>(a.sc, e, sig0, t);
    const bool plastic = live && (t.m.f > 0.0);
    const unsigned long long mask = __ballot(plastic);
    DPTangent tg;
    tg.t11 = a.sc.s[2], tg.tP = a.sc.s[7];
    if (plastic) dp_return<HYPER>(a.sc, e, sig0, t, h, tg, st);
    st.plastic += (lane == 0) ? (unsigned long long)__popcll(mask) : 0ull;
    st.domain += (live && t.tip) ? 1ull : 0ull;

    wrapped_store_stress<WRAP, FULL, NT>(a, region, p0, npts, lane, t.sig1);
    if (mask != 0ull) transpose_out<7, FULL, NT>(h, region, lane, a.h0_out + p0 * 7, npts * 7);
    if (a.tangent) {
        if constexpr (WRAP == 1) {
            // entry [0][0] exactly as tangent_dp forms it; elastic points carry E[0][0] itself
            const double s0 = tg.rho * t.s_tr[0];
            const double v = (tg.t11 * T->a[0] + tg.tP * T->b[0]) + ((tg.tss * s0) * s0 + (tg.t1s * 1.0) * s0 + (tg.ts1 * s0) * 1.0);
            if (live) a.tangent[p0 + lane] = plastic ? v : T->c[0];
        } else {
            dp_publish(region, lane, tg, t.s_tr, plastic);
            wave_sync();
            const int nchunks = npts * 8;  // block [0:4, 0:4]
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int q = k * kWave + lane;
                const int p = q >> 3, r = q & 7;
                const int i = r >> 1, j = 2 * (r & 1);
                const double* tp = region + kDpStride * p;
                const d2 c0 = reinterpret_cast<const d2*>(tp)[0];
                const d2 c1 = reinterpret_cast<const d2*>(tp)[1];
                const d2 c2 = reinterpret_cast<const d2*>(tp)[2];
                const double si = tp[6 + i];
                const d2 sj = *reinterpret_cast<const d2*>(tp + 6 + j);
                const d2 o = *reinterpret_cast<const d2*>(T->a + 6 * i + j);
                const d2 pd = *reinterpret_cast<const d2*>(T->b + 6 * i + j);
                const double oi = i < 3 ? 1.0 : 0.0;
                d2 v;
                v.x = (c0.x * o.x + c0.y * pd.x) + ((c1.x * si) * sj.x + (c1.y * oi) * sj.x + (c2.x * si) * (j < 3 ? 1.0 : 0.0));
                v.y = (c0.x * o.y + c0.y * pd.y) + ((c1.x * si) * sj.y + (c1.y * oi) * sj.y + (c2.x * si) * (j + 1 < 3 ? 1.0 : 0.0));
                const d2 el = *reinterpret_cast<const d2*>(T->c + 6 * i + j);
                if (c2.y == 0.0) v = el;
                if constexpr (FULL) {
                    store16<NT>(a.tangent + p0 * 16 + 2 * q, v);
                } else if (q < nchunks) {
                    a.tangent[p0 * 16 + 2 * q] = v.x;
                    a.tangent[p0 * 16 + 2 * q + 1] = v.y;
                }
            }
            wave_sync();
        }
    }
}

}  // namespace fcamd
