// SpringMaxwellModel / SpringKelvinModel (models/spring_maxwell_model.py:40-88, models/spring_kelvin_model.py:43-88).
// Part of the device code of libfcamd (translation unit: ../fcamd_kernels.hip, which holds the kernels and launchers).
#pragma once
#include "tile_io.h"
#include "tangent_writers.h"

namespace fcamd {

// --- SLS Maxwell / Kelvin -----------------------------------------------------------------
// scalars: s[0]=strain factor, s[1]=1/factor, s[2]=1/(tau*2*mu1), s[3]=1/tau,
//          Maxwell: s[4]=2*mu1 ; Kelvin: s[4]=2*mu0, s[5]=mu0/(tau*mu1), s[6]=lam0/(tau*2*mu1)
// tables:  Maxwell a=D1, b=D0+D1, c=tangent ; Kelvin a=D0, c=tangent
template <bool KELVIN, bool IDX, bool FULL, bool NT>
__device__ __forceinline__ void tile_sls(ArgsRef a, const StressBases& sb, const Tables* T, double* region,
                                         int* rows_lds, long long p0, int npts, int lane, int r0) {
    Chunks<9> cg;
    StressRows<IDX, FULL, NT> sr;
    Chunks<6> cv, cn;
    tile_load<9, FULL, NT>(cg, a.grad + p0 * 9, npts * 9, lane);
    sr.load(a, sb, p0, npts, lane, rows_lds);
    tile_load<6, FULL, NT>(cv, a.h0_in + p0 * 6, npts * 6, lane);  // strain_visco
    tile_load<6, FULL, NT>(cn, a.h1_in + p0 * 6, npts * 6, lane);  // strain
    if (sb.tan) {
        if constexpr (IDX) wave_sync();  // rows_lds visible to all lanes
        tangent_const<IDX, FULL, NT>(T->c, sb.tan, p0, rows_lds, npts, lane, r0);
    }
    double g[9], s[6], ev[6], en[6], e[6], dv[6], y[6];
    transpose_in<9>(cg, region, lane, g);
    sr.get(region, lane, s);
    transpose_in<6>(cv, region, lane, ev);
    transpose_in<6>(cn, region, lane, en);
    mandel_strain(g, a.sc.s[0], e);
    const double inv_factor = a.sc.s[1], cA = a.sc.s[2], cB = a.sc.s[3], c2mu = a.sc.s[4];
    if constexpr (!KELVIN) {
        // deps_v = 1/factor * ((cA * (eps_n + d_eps)) @ D1 - 1/tau * eps_v)
        double x[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) x[i] = cA * (en[i] + e[i]);
        row_times_matrix_fma(x, T->a, y);
#pragma unroll
        for (int i = 0; i < 6; ++i) dv[i] = inv_factor * (y[i] - cB * ev[i]);
        // sigma += d_eps @ (D0 + D1) - 2 mu1 deps_v
        row_times_matrix_fma(e, T->b, y);
    } else {
        const double cC = a.sc.s[5], cD = a.sc.s[6];
        const double tr = (e[0] + e[1]) + e[2];
        const double ctr = cD * tr;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const double id = i < 3 ? ctr : ctr * 0.0;
            dv[i] = inv_factor * (((cA * s[i] - cB * ev[i]) + cC * e[i]) + id);
        }
        // sigma += d_eps @ D0 - 2 mu0 deps_v
        row_times_matrix_fma(e, T->a, y);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        s[i] = s[i] + (y[i] - c2mu * dv[i]);
        ev[i] = ev[i] + dv[i];
        en[i] = en[i] + e[i];
    }
    sr.put(sb, region, lane, s, p0, npts, rows_lds);
    transpose_out<6, FULL, NT>(ev, region, lane, a.h0_out + p0 * 6, npts * 6);
    transpose_out<6, FULL, NT>(en, region, lane, a.h1_out + p0 * 6, npts * 6);
}

}  // namespace fcamd
