// LinearElasticityModel (models/linear_elasticity_model.py:26-45) and comfe-rs LinearElasticity3D (linear_elasticity.rs:42-75).
// Part of the device code of libfcamd (translation unit: ../fcamd_kernels.hip, which holds the kernels and launchers).
#pragma once
#include "tile_io.h"
#include "tangent_writers.h"
#include "wrapped_io.h"

namespace fcamd {

// ---------------------------------------------------------------------------------------
// tile bodies, one per law.  `region` is the wave's LDS scratch, `T` the staged tables.
// ---------------------------------------------------------------------------------------

// --- LinearElasticityModel: sigma += d_eps @ D ; tangent = tile(D) ----------------------
template <bool IDX, bool FULL, bool NT>
__device__ __forceinline__ void tile_linear_elasticity(ArgsRef a, const StressBases& sb, const Tables* T,
                                                       double* region, int* rows_lds, long long p0,
                                                       int npts, int lane, int r0) {
    Chunks<9> cg;
    StressRows<IDX, FULL, NT> sr;
    tile_load<9, FULL, NT>(cg, a.grad + p0 * 9, npts * 9, lane);
    sr.load(a, sb, p0, npts, lane, rows_lds);
    // the constant tangent does not depend on the loads: stream it while they are in flight
    if (sb.tan) {
        if constexpr (IDX) wave_sync();  // rows_lds visible to all lanes
        tangent_const<IDX, FULL, NT>(T->c, sb.tan, p0, rows_lds, npts, lane, r0);
    }
    double g[9], s[6], e[6], ds[6];
    transpose_in<9>(cg, region, lane, g);
    sr.get(region, lane, s);
    mandel_strain(g, a.sc.s[0], e);
    row_times_matrix_fma(e, T->a, ds);
#pragma unroll
    for (int i = 0; i < 6; ++i) s[i] = s[i] + ds[i];
    sr.put(sb, region, lane, s, p0, npts, rows_lds);
}

// --- comfe-rs LinearElasticity3D: sigma += C . d_eps (column axpy, no FMA) ---------------
template <bool IDX, bool FULL, bool NT>
__device__ __forceinline__ void tile_comfe_le(ArgsRef a, const StressBases& sb, const Tables* T, double* region,
                                              int* rows_lds, long long p0, int npts, int lane, int r0) {
    Chunks<9> cg;
    StressRows<IDX, FULL, NT> sr;
    tile_load<9, FULL, NT>(cg, a.grad + p0 * 9, npts * 9, lane);
    sr.load(a, sb, p0, npts, lane, rows_lds);
    if (sb.tan) {
        if constexpr (IDX) wave_sync();  // rows_lds visible to all lanes
        tangent_const<IDX, FULL, NT>(T->c, sb.tan, p0, rows_lds, npts, lane, r0);
    }
    double g[9], s[6], e[6];
    transpose_in<9>(cg, region, lane, g);
    sr.get(region, lane, s);
    mandel_strain(g, a.sc.s[0], e);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double acc = T->a[6 * i] * e[0];
#pragma unroll
        for (int j = 1; j < 6; ++j) acc = T->a[6 * i + j] * e[j] + acc;
        s[i] = s[i] + acc;
    }
    sr.put(sb, region, lane, s, p0, npts, rows_lds);
}

// LinearElasticityModel behind the wrappers (what the reference's own tests wrap, test_elasticity.py:206,278)
template <int WRAP, bool FULL, bool NT>
__device__ __forceinline__ void tile_linear_elasticity_wrapped(ArgsRef a, const Tables* T, double* region,
                                                               long long p0, int npts, int lane) {
    if (a.tangent) wrapped_tangent_const<WRAP, FULL, NT>(a, T->c, p0, npts, lane);
    double g[9], s[6], e[6], ds[6];
    wrapped_load<WRAP, FULL, NT>(a, region, p0, npts, lane, g, s);
    mandel_strain(g, a.sc.s[0], e);
    row_times_matrix_fma(e, T->a, ds);
#pragma unroll
    for (int i = 0; i < 6; ++i) s[i] = s[i] + ds[i];
    wrapped_store_stress<WRAP, FULL, NT>(a, region, p0, npts, lane, s);
}

}  // namespace fcamd
