// Low-dimensional constraints of LinearElasticityModel and the SLS laws (uniaxial strain / stress, plane strain / stress).
// Part of the device code of libfcamd (translation unit: ../fcamd_kernels.hip, which holds the kernels and launchers).
#pragma once
#include "tile_io.h"

namespace fcamd {

// ---------------------------------------------------------------------------------------
// Low-dimensional constraints (uniaxial strain/stress: DIMS = 1; plane strain/stress:
// DIMS = 2) of the three laws the reference implements "for all constraints": LE
// (linear_elasticity_model.py:26-45), Maxwell (spring_maxwell_model.py:40-88), Kelvin
// (spring_kelvin_model.py:43-88).  Same skeleton as the 3-D tiles; gradient GD2 = DIMS^2 and
// Mandel vectors SD = 1 / 4 doubles per point, tangent SD^2.  The strain/stress variants of
// one dimension differ only in host constants (tangent tables, identity vector).
// Tables are compact (row stride SD).  scalars as in tile_sls; s[8 + i] = I2[i].
// ---------------------------------------------------------------------------------------
template <int DIMS>
struct LowDim {
    static constexpr int GD2 = DIMS * DIMS;
    static constexpr int SD = DIMS == 2 ? 4 : 1;
};

// strain_from_grad_u, utils.py:153-186
template <int DIMS>
__device__ __forceinline__ void strain_lowdim(const double (&g)[LowDim<DIMS>::GD2], double f,
                                              double (&e)[LowDim<DIMS>::SD]) {
    if constexpr (DIMS == 1) {
        e[0] = g[0];
    } else {
        e[0] = g[0];
        e[1] = g[3];
        e[2] = 0.0;
        e[3] = f * (g[1] + g[2]);
    }
}

template <int SD, class TAB>
__device__ __forceinline__ void row_times_matrix_fma_n(const double (&x)[SD], TAB M,  // M: a table in LDS or among the kernel arguments
                                                       double (&y)[SD]) {
#pragma unroll
    for (int i = 0; i < SD; ++i) {
        double acc = x[0] * M[i];
#pragma unroll
        for (int k = 1; k < SD; ++k) acc = __builtin_fma(x[k], M[SD * k + i], acc);
        y[i] = acc;
    }
}

// tangent = tile(D.flatten()): SD = 4 -> 8 chunks per point (the chunk a lane stores never
// changes: 64 = 0 mod 8); SD = 1 -> half a chunk per point, lanes < 32 store (D, D).
template <int SD, bool FULL, bool NT>
__device__ __forceinline__ void tangent_const_n(const double* tab, double* dst, int npts, int lane) {
    if constexpr (SD == 4) {
        const d2 v = reinterpret_cast<const d2*>(tab)[lane & 7];
        const int nchunks = npts * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = k * kWave + lane;
            if (FULL || q < nchunks) store16<NT>(dst + 2 * q, v);
        }
    } else {
        const double D = tab[0];
        if constexpr (FULL) {
            d2 v;
            v.x = D;
            v.y = D;
            if (lane < 32) store16<NT>(dst + 2 * lane, v);
        } else {
            if (lane < npts) dst[lane] = D;
        }
    }
}

// One point: stress and the two history fields advanced by the strain increment e.  A, B = the law's tables (LDS for the
// tiled kernels, kernel arguments for the uniaxial stream).
template <int LAW, int DIMS, class TAB>
__device__ __forceinline__ void lowdim_point(ScalarsRef sc, TAB A, TAB B,
                                             const double (&e)[LowDim<DIMS>::SD], double (&s)[LowDim<DIMS>::SD],
                                             double (&ev)[LowDim<DIMS>::SD], double (&en)[LowDim<DIMS>::SD]) {
    constexpr int SD = LowDim<DIMS>::SD;
    double y[SD];
    if constexpr (LAW == LAW_LE) {
        row_times_matrix_fma_n<SD>(e, A, y);
#pragma unroll
        for (int i = 0; i < SD; ++i) s[i] = s[i] + y[i];
    } else {
        double dv[SD];
        const double inv_factor = sc.s[1], cA = sc.s[2], cB = sc.s[3], c2mu = sc.s[4];
        if constexpr (LAW == LAW_MAXWELL) {
            double x[SD];
#pragma unroll
            for (int i = 0; i < SD; ++i) x[i] = cA * (en[i] + e[i]);
            row_times_matrix_fma_n<SD>(x, A, y);
#pragma unroll
            for (int i = 0; i < SD; ++i) dv[i] = inv_factor * (y[i] - cB * ev[i]);
            row_times_matrix_fma_n<SD>(e, B, y);
        } else {
            const double cC = sc.s[5], cD = sc.s[6];
            double tr = e[0];  // np.sum(strain_increment[:, :gdim], axis=1), gdim = DIMS
            if constexpr (DIMS == 2) tr = e[0] + e[1];
            const double ctr = cD * tr;
#pragma unroll
            for (int i = 0; i < SD; ++i)
                dv[i] = inv_factor * (((cA * s[i] - cB * ev[i]) + cC * e[i]) + ctr * sc.s[8 + i]);
            row_times_matrix_fma_n<SD>(e, A, y);
        }
#pragma unroll
        for (int i = 0; i < SD; ++i) {
            s[i] = s[i] + (y[i] - c2mu * dv[i]);
            ev[i] = ev[i] + dv[i];
            en[i] = en[i] + e[i];
        }
    }
}

template <int LAW, int DIMS, bool FULL, bool NT>
__device__ __forceinline__ void tile_lowdim(ArgsRef a, const Tables* T, double* region,
                                            long long p0, int npts, int lane) {
    constexpr int GD2 = LowDim<DIMS>::GD2, SD = LowDim<DIMS>::SD;
    constexpr bool HIST = (LAW != LAW_LE);
    Chunks<GD2> cg;
    Chunks<SD> cs, cv, cn;
    tile_load<GD2, FULL, NT>(cg, a.grad + p0 * GD2, npts * GD2, lane);
    tile_load<SD, FULL, NT>(cs, a.stress_in + p0 * SD, npts * SD, lane);
    if constexpr (HIST) {
        tile_load<SD, FULL, NT>(cv, a.h0_in + p0 * SD, npts * SD, lane);
        tile_load<SD, FULL, NT>(cn, a.h1_in + p0 * SD, npts * SD, lane);
    }
    if (a.tangent) tangent_const_n<SD, FULL, NT>(T->c, a.tangent + p0 * SD * SD, npts, lane);
    double g[GD2], s[SD], e[SD], ev[SD], en[SD];
    transpose_in<GD2>(cg, region, lane, g);
    transpose_in<SD>(cs, region, lane, s);
    strain_lowdim<DIMS>(g, a.sc.s[0], e);
    if constexpr (HIST) {
        transpose_in<SD>(cv, region, lane, ev);
        transpose_in<SD>(cn, region, lane, en);
    }
    lowdim_point<LAW, DIMS>(a.sc, T->a, T->b, e, s, ev, en);
    transpose_out<SD, FULL, NT>(s, region, lane, a.stress_out + p0 * SD, npts * SD);
    if constexpr (HIST) {
        transpose_out<SD, FULL, NT>(ev, region, lane, a.h0_out + p0 * SD, npts * SD);
        transpose_out<SD, FULL, NT>(en, region, lane, a.h1_out + p0 * SD, npts * SD);
    }
}

// Uniaxial constraints are one double per point in every array: AoS = SoA, no transposition.  A lane takes PAIRS of
// consecutive points (16-byte accesses), two pairs per trip (2 KiB per wave and array in flight); the arithmetic is
// lowdim_point<LAW, 1>, the same instructions as the tiled form.
template <int LAW, bool NT>
__device__ __forceinline__ void pair_uniaxial(ArgsRef a, long long q, const d2 g, const d2 s0, const d2 v0, const d2 n0) {
    double e[1], s[1], ev[1], en[1];
    d2 so, vo, no;
    e[0] = g.x; s[0] = s0.x; ev[0] = v0.x; en[0] = n0.x;
    lowdim_point<LAW, 1>(a.sc, a.tb.a, a.tb.b, e, s, ev, en);
    so.x = s[0]; vo.x = ev[0]; no.x = en[0];
    e[0] = g.y; s[0] = s0.y; ev[0] = v0.y; en[0] = n0.y;
    lowdim_point<LAW, 1>(a.sc, a.tb.a, a.tb.b, e, s, ev, en);
    so.y = s[0]; vo.y = ev[0]; no.y = en[0];
    store16<NT>(a.stress_out + 2 * q, so);
    if constexpr (LAW != LAW_LE) {
        store16<NT>(a.h0_out + 2 * q, vo);
        store16<NT>(a.h1_out + 2 * q, no);
    }
    if (a.tangent) {
        d2 t;
        t.x = a.tb.c[0];
        t.y = a.tb.c[0];
        store16<NT>(a.tangent + 2 * q, t);
    }
}

template <int LAW, bool NT>
__device__ __forceinline__ void stream_uniaxial(ArgsRef a, long long npairs, long long first, long long stride) {
    constexpr bool HIST = (LAW != LAW_LE);
    const d2 z = {0.0, 0.0};
    long long q = first;
    for (; q + stride < npairs; q += 2 * stride) {
        const long long r = q + stride;
        const d2 g0 = load16<NT>(a.grad + 2 * q), g1 = load16<NT>(a.grad + 2 * r);
        const d2 s0 = load16<NT>(a.stress_in + 2 * q), s1 = load16<NT>(a.stress_in + 2 * r);
        d2 v0 = z, v1 = z, n0 = z, n1 = z;
        if constexpr (HIST) {
            v0 = load16<NT>(a.h0_in + 2 * q), v1 = load16<NT>(a.h0_in + 2 * r);
            n0 = load16<NT>(a.h1_in + 2 * q), n1 = load16<NT>(a.h1_in + 2 * r);
        }
        pair_uniaxial<LAW, NT>(a, q, g0, s0, v0, n0);
        pair_uniaxial<LAW, NT>(a, r, g1, s1, v1, n1);
    }
    if (q < npairs) {
        const d2 g0 = load16<NT>(a.grad + 2 * q), s0 = load16<NT>(a.stress_in + 2 * q);
        d2 v0 = z, n0 = z;
        if constexpr (HIST) v0 = load16<NT>(a.h0_in + 2 * q), n0 = load16<NT>(a.h1_in + 2 * q);
        pair_uniaxial<LAW, NT>(a, q, g0, s0, v0, n0);
    }
}

}  // namespace fcamd
