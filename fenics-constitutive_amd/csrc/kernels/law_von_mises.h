// VonMises3D (models/mises_plasticity_isotropic_hardening.py:57-175): point functions, the 3-D tile, the fused wrapper tile.
// Part of the device code of libfcamd (translation unit: ../fcamd_kernels.hip, which holds the kernels and launchers).
#pragma once
#include "tile_io.h"
#include "tangent_writers.h"
#include "wrapped_io.h"
#include "history_rows.h"

namespace fcamd {

// ---------------------------------------------------------------------------------------
// Point arithmetic of the two Mises laws, shared by the 3-D tiles and the fused 3D -> 1D/2D wrapper tiles
// (tile_von_mises / tile_von_mises_wrapped, tile_comfe_mises / tile_comfe_mises_wrapped): every statement
// of the reference exists once.  Everything is per lane and forced inline.
// ---------------------------------------------------------------------------------------

// VonMises3D (models/mises_plasticity_isotropic_hardening.py:75-94): trial state of one point
struct VMTrial {
    double dsig[6], sigtr[6];  // del_sigtr = 2 mu dev(d_eps), sigtr = dev(sigma_n) + del_sigtr
    double tr_eps, sigtrn, phitr;
};

__device__ __forceinline__ void vm_trial(ScalarsRef sc, const double (&e)[6], const double (&s)[6], double alpha_n,
                                         VMTrial& t) {
    const double two_mu = sc.s[2], s23 = sc.s[3], y0 = sc.s[4], dy = sc.s[5], mw = sc.s[6];
    t.tr_eps = (e[0] + e[1]) + e[2];
    const double tr_sig = (s[0] + s[1]) + s[2];
    const double tr_eps3 = t.tr_eps / 3.0, tr_sig3 = tr_sig / 3.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double ed = i < 3 ? e[i] - tr_eps3 : e[i];
        const double sd = i < 3 ? s[i] - tr_sig3 : s[i];
        t.dsig[i] = two_mu * ed;
        t.sigtr[i] = sd + t.dsig[i];
    }
    double nn = t.sigtr[0] * t.sigtr[0];
#pragma unroll
    for (int i = 1; i < 6; ++i) nn = __builtin_fma(t.sigtr[i], t.sigtr[i], nn);  // np.dot == fma chain
    t.sigtrn = sqrt(nn);
    t.phitr = t.sigtrn - s23 * (y0 + dy * (1.0 - exp(mw * alpha_n)));
}

// return mapping of one plastic point (:98-151): Newton on the plastic multiplier with the reference's
// stopping rule (it tests the residual of the PREVIOUS iterate, so one more update follows convergence)
struct VMReturn {
    double gamma = 0.0, xc1 = 0.0, xc2 = 0.0;
    double N[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
};

__device__ __forceinline__ void vm_return(ScalarsRef sc, const VMTrial& t, double alpha_n, VMReturn& r, WaveStats& st) {
    const double two_mu = sc.s[2], s23 = sc.s[3], y0 = sc.s[4], dy = sc.s[5], mw = sc.s[6], m2mu = sc.s[7], c23dyw = sc.s[8];
    double g0 = 1.0, g1 = 0.0, xr = 1.0, xg;
    int it = 0;
    bool failed = false;
    while (__builtin_fabs(xr) > 1e-12 && __builtin_fabs(g1 - g0) > 1e-8 * __builtin_fabs(g1)) {
        g0 = g1;
        ++it;
        const double ex = exp(mw * (alpha_n + s23 * g0));
        xr = (t.sigtrn - two_mu * g0) - s23 * (y0 + dy * (1.0 - ex));
        xg = m2mu - c23dyw * ex;
        g1 = g0 - xr / xg;
        if (it > 100) {
            failed = true;
            break;
        }
    }
    const double ex = exp(mw * (alpha_n + s23 * g1));
    xg = m2mu - c23dyw * ex;
    r.xc1 = -1.0 / xg;
    r.xc2 = g1 / t.sigtrn;
    r.gamma = g1;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.N[i] = t.sigtr[i] / t.sigtrn;
    st.iters += (unsigned long long)it;
    st.nonconv += failed ? 1ull : 0ull;
}

// stress (:165-167): sigma += (ka tr_eps) I2 + del_sigtr - (2 mu gamma) N;  tangent coefficients (:170-175)
__device__ __forceinline__ void vm_stress(ScalarsRef sc, const VMTrial& t, const VMReturn& r, double (&s)[6]) {
    const double kt = sc.s[1] * t.tr_eps, tmg = sc.s[2] * r.gamma;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double vol = i < 3 ? kt : kt * 0.0;
        s[i] = s[i] + ((vol + t.dsig[i]) - tmg * r.N[i]);
    }
}
__device__ __forceinline__ void vm_tangent_coefficients(ScalarsRef sc, const VMReturn& r, double& B, double& C) {
    const double two_mu = sc.s[2], four_mu2 = sc.s[9];
    B = two_mu * (1.0 - two_mu * r.xc2);
    C = four_mu2 * (r.xc2 - r.xc1);
}

// --- VonMises3D: J2 plasticity, saturation hardening, scalar Newton per point -----------------
// scalars: s[0]=strain factor, s[1]=ka, s[2]=2*mu, s[3]=sqrt(2/3), s[4]=y0, s[5]=y00-y0,
//          s[6]=-w, s[7]=(-2)*mu, s[8]=((2/3)*(y00-y0))*w, s[9]=(4*mu)*mu
// tables:  a = ka*xioi, b = xpp
// HIST: 0 = the caller's arrays as they are (in place or out of place), 1 = sparse trial history (a.hmask),
//       2 = sparse protocol on the packed plastic-strain layout (history_rows.h: PackedRows)
// PM: the tangent leaves as its 8 parameters per point (0 never, 1 always, 2 by kFlagTangentParams at run time; fcamd_kernels.hip: run_tile)
// TWIN: the synthetic twin (tangent_writers.h: kFlagTwin) -- ballots from a.cache3d, no constitutive arithmetic
template <bool IDX, int HIST, bool FULL, bool NT, int PM = 0, bool TWIN = false>
__device__ __forceinline__ void tile_von_mises(ArgsRef a, const StressBases& sb, const Tables* T, double* region,
                                               int* rows_lds, long long p0, int npts, int lane,
                                               WaveStats& st) {
    constexpr bool sparse = HIST != 0;
    constexpr bool packed = HIST == 2;
    // The per-tile words of the sparse protocol (previous plastic ballot; EVER mask of the committed packed run) are the
    // tile's FIRST loads: they arrive with the gradient instead of costing a dependent memory round trip after the ballot.
    unsigned long long m_old = 0ull;
    PackedRows<FULL, NT> pk;
    if constexpr (sparse) m_old = a.hmask[p0 >> 6];
    unsigned long long ever_trial = 0ull;
    if constexpr (packed) {
        pk.ever_in = a.emask_in[p0 >> 6];
        ever_trial = a.emask_out[p0 >> 6];
    }
    Chunks<9> cg;
    StressRows<IDX, FULL, NT> sr;
    tile_load<9, FULL, NT>(cg, a.grad + p0 * 9, npts * 9, lane);
    sr.load(a, sb, p0, npts, lane, rows_lds);
    const bool live = FULL || lane < npts;
    const double alpha_n = live ? a.h1_in[p0 + lane] : 0.0;
    const bool hist_in_place = (a.h0_in == a.h0_out) && (a.h1_in == a.h1_out);
    // packed layout: a tile that was plastic at the previous evaluate is touched whatever happens now (new values or stale
    // rows) -- its committed run is requested right away, long before the ballot
    // ... unless few rows of a long run were touched last time and the trial run has the committed layout: then few will be
    // touched now, and they are requested alone after the ballot (PackedRows::load_rows)
    const int run_rows = (int)__popcll(pk.ever_in);
    const bool same_layout = FULL && ever_trial == pk.ever_in && run_rows >= kPackedRowsMinRun;  // (short runs: the whole run is 1-6 lines)
    const bool early = packed && m_old != 0ull && !(same_layout && kPackedRowsDiv * (int)__popcll(m_old) <= run_rows);
    if constexpr (packed) {
        if (early) pk.load(a.h0_in, p0, lane);
    }
    double g[9], s[6], e[6];
    transpose_in<9>(cg, region, lane, g);
    sr.get(region, lane, s);
    mandel_strain(g, a.sc.s[0], e);

    VMTrial tr;
    bool plastic;
    if constexpr (TWIN) {
        const unsigned long long recorded = reinterpret_cast<const unsigned long long*>(a.cache3d)[p0 >> 6];
#pragma unroll
        for (int i = 0; i < 6; ++i) tr.dsig[i] = e[i], tr.sigtr[i] = s[i];
        tr.tr_eps = e[0], tr.sigtrn = 1.0, tr.phitr = 0.0;
        plastic = live && ((recorded >> lane) & 1ull) != 0ull;
    } else {
        vm_trial(a.sc, e, s, alpha_n, tr);
        plastic = live && (tr.phitr > 0.0);
    }
    const unsigned long long mask = __ballot(plastic);

    // plastic-strain history: needed only by tiles with a plastic point (in place), or always
    // when the trial history lives in a different array (out of place).
    //
    // Sparse trial history (a.hmask != nullptr; device-resident Newton loops): the trial arrays
    // are kept equal to the committed ones except at the points recorded in hmask (one 64-bit
    // word per tile = the plastic ballot of the previous evaluate).  Then only plastic points
    // (new trial value) and stale points (plastic last time, elastic now: restore the committed
    // value) need their 48-byte eps_n row touched; elastic points cost no history traffic at all,
    // which is exactly the algorithmic byte count (464 B/pt elastic, 568 B/pt plastic).  The plain
    // in-place call (the reference contract) is the same case without stale points.
    //
    // Row-masked tile access: a 48-byte row is exactly three 16-byte chunks of the tile's linear
    // image (chunk q belongs to row q / 3), so the tile keeps its three wave-wide, address-ordered
    // load and store instructions and every lane simply skips the chunks of untouched rows.  HBM
    // sees the touched rows only (reads at the 128-byte line granularity of the memory side), the
    // instruction count does not depend on how many rows are touched, and a fully plastic tile
    // degenerates to the plain coalesced tile access.
    Chunks<6> ce;
    const unsigned long long need_mask = mask | m_old;
    const unsigned long long eps_mask = need_mask;
    const bool masked = !packed && FULL && (sparse || hist_in_place) && ((int)__popcll(eps_mask) <= a.masked_max);
    const bool touch_eps = masked ? (eps_mask != 0ull)
                                  : (sparse ? (eps_mask != 0ull) : ((mask != 0ull) || !hist_in_place));
    const bool touch_alpha = touch_eps;  // stale points get their committed alpha back
    bool row_live[3] = {true, true, true};  // per chunk of this lane: its row is touched
    if (masked) {
#pragma unroll
        for (int k = 0; k < 3; ++k) row_live[k] = rows_granule_live(eps_mask, k * kWave + lane);
    }
    if constexpr (packed) {
        if (!early && touch_eps) {  // a tile that turns plastic now, or one with few touched rows in a long run: the late request
            if (same_layout && (mask & ~pk.ever_in) == 0ull && kPackedRowsDiv * (int)__popcll(need_mask) <= run_rows)
                pk.load_rows(a.h0_in, p0, lane, need_mask, region);
            else
                pk.load(a.h0_in, p0, lane);
        }
    } else if (touch_eps) {
        if (masked) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                d2 z;
                z.x = 0.0;
                z.y = 0.0;
                ce.v[k] = row_live[k] ? load16<NT>(a.h0_in + p0 * 6 + 2 * (k * kWave + lane)) : z;
            }
        } else {
            tile_load<6, FULL, NT>(ce, a.h0_in + p0 * 6, npts * 6, lane);
        }
    }

    VMReturn rm;
    if (mask != 0ull) {
        if constexpr (TWIN) {
            if (plastic) {
                rm.gamma = e[1];
#pragma unroll
                for (int i = 0; i < 6; ++i) rm.N[i] = s[i];
            }
        } else {
            if (plastic) vm_return(a.sc, tr, alpha_n, rm, st);
        }
        st.plastic += (lane == 0) ? (unsigned long long)__popcll(mask) : 0ull;
    }

    // The committed plastic-strain rows are taken out of their registers BEFORE the first store of this tile is issued: a
    // load consumed after younger stores makes the wave wait for those stores to complete (one vmcnt for both; measured in
    // round 4: 0.7 ms of 8.4 at 1e8 points with the rows consumed after the stress store).
    double ep[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const bool ep_in_lanes = !packed && touch_eps && mask != 0ull;
    if constexpr (packed) {
        if (touch_eps) pk.gather(region, lane, ep);
    } else {
        if (ep_in_lanes) transpose_in<6>(ce, region, lane, ep);
    }

    vm_stress(a.sc, tr, rm, s);
    sr.put(sb, region, lane, s, p0, npts, rows_lds);

    // history: eps_n += gamma N ; alpha += sqrt(2/3) gamma
    if constexpr (packed) {
        if (touch_eps) {
#pragma unroll
            for (int i = 0; i < 6; ++i) ep[i] = plastic ? ep[i] + rm.gamma * rm.N[i] : ep[i];  // the others keep their bits
            pk.scatter(a, a.h0_out, p0, lane, mask, region, ep);
        }
    } else if (touch_eps) {
        if (mask != 0ull) {
#pragma unroll
            for (int i = 0; i < 6; ++i) ep[i] = plastic ? ep[i] + rm.gamma * rm.N[i] : ep[i];  // the others keep their bits
            if (masked) {
                lds_put_point<6>(region, lane, ep);
                wave_sync();
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int q = k * kWave + lane;
                    if (row_live[k]) store16<NT>(a.h0_out + p0 * 6 + 2 * q, reinterpret_cast<const d2*>(region)[q]);
                }
                wave_sync();
            } else {
                transpose_out<6, FULL, NT>(ep, region, lane, a.h0_out + p0 * 6, npts * 6);
            }
        } else if (masked) {  // only stale rows: restore the committed values
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (row_live[k]) store16<NT>(a.h0_out + p0 * 6 + 2 * (k * kWave + lane), ce.v[k]);
        } else {
            tile_store<6, FULL, NT>(ce, a.h0_out + p0 * 6, npts * 6, lane);
        }
    }
    // alpha: one coalesced 512-byte store per touched tile (round 4: storing only the 32-byte sectors that hold a plastic or
    // stale point saves 3 B/pt and costs 1 % -- partial lines; 64- / 128-byte granules neither gain nor lose)
    if (touch_alpha && live) a.h1_out[p0 + lane] = alpha_n + a.sc.s[3] * rm.gamma;
    if constexpr (sparse) {
        if (lane == 0 && mask != m_old) a.hmask[p0 >> 6] = mask;
    }

    // tangent: ka xioi + 2 mu (1 - 2 mu xc2) xpp + 4 mu^2 (xc2 - xc1) N (x) N
    const unsigned long long tneed = sparse_tangent_need<FULL>(a, need_mask);
    if (sb.tan && tneed != 0ull) {
        double B, C;
        vm_tangent_coefficients(a.sc, rm, B, C);
        publish_tangent_params(region, lane, B, C, rm.N);
        wave_sync();
        if constexpr (TWIN && FULL && !IDX) {
            if (tneed == ~0ull) tangent_const<false, true, NT>(T->a, sb.tan, p0, rows_lds, npts, lane, lane % 18);
            else tangent_twin_passes<NT>(region, sb.tan + p0 * 36, lane, tneed, (a.flags & kFlagExactTangentRows) != 0, std::make_integer_sequence<int, 18>{});
        } else if (tangent_params_mode<PM>(a))  // the host rebuilds the rows (fcamd_hosttangent.cpp)
            store_tangent_params<FULL, NT>(a, region, sb.tan, p0, npts, lane, mask);
        else
            tangent_mises<false, IDX, FULL, NT>(region, T->a, T->b, sb.tan, p0, rows_lds, npts, lane, tneed, (a.flags & kFlagExactTangentRows) != 0);
        wave_sync();
    }
}

template <int WRAP, bool FULL, bool NT>
__device__ __forceinline__ void tile_von_mises_wrapped(ArgsRef a, const Tables* T, double* region,
                                                       long long p0, int npts, int lane, WaveStats& st) {
    const bool live = FULL || lane < npts;
    double g[9], s[6], e[6];
    const double alpha_n = live ? a.h1_in[p0 + lane] : 0.0;
    wrapped_load<WRAP, FULL, NT>(a, region, p0, npts, lane, g, s);
    mandel_strain(g, a.sc.s[0], e);

    VMTrial tr;
    vm_trial(a.sc, e, s, alpha_n, tr);
    const bool plastic = live && (tr.phitr > 0.0);
    const unsigned long long mask = __ballot(plastic);

    // eps_n of the plastic points, in place: requested right after the ballot (in flight with the Newton iteration), taken into
    // the lanes before the tile's first store, row-masked when few points are plastic (MaskedRows)
    MaskedRows<FULL, NT> er;
    if (mask != 0ull) er.request(a.h0_in, p0, npts, lane, mask, a.masked_max);
    VMReturn rm;
    if (mask != 0ull) {
        if (plastic) vm_return(a.sc, tr, alpha_n, rm, st);
        st.plastic += (lane == 0) ? (unsigned long long)__popcll(mask) : 0ull;
    }
    double ep[6];
    if (mask != 0ull) er.gather(region, lane, ep);
    vm_stress(a.sc, tr, rm, s);
    wrapped_store_stress<WRAP, FULL, NT>(a, region, p0, npts, lane, s);
    if (mask != 0ull) {
#pragma unroll
        for (int i = 0; i < 6; ++i) ep[i] = plastic ? ep[i] + rm.gamma * rm.N[i] : ep[i];  // the others keep their bits
        er.store(a.h0_out, p0, npts, lane, region, ep);
        // alpha, in place: the 32-byte sectors that hold a plastic point (the others keep their value; whole sectors so that no
        // partial sector is written: -2 % on identical buffers at 6 % plastic points against the whole tile's 512 bytes)
        if (live && ((mask >> (lane & ~3)) & 0xFull) != 0ull) a.h1_out[p0 + lane] = alpha_n + a.sc.s[3] * rm.gamma;
    }
    if (a.tangent) {
        double B, C;
        vm_tangent_coefficients(a.sc, rm, B, C);
        wrapped_tangent_mises<false, WRAP, FULL, NT>(a, T, region, p0, npts, lane, B, C, rm.N);
    }
}

}  // namespace fcamd
