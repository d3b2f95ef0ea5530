// Input / output side of the fused 3D -> 1D/2D wrapper tiles (UniaxialStrainFrom3D / PlaneStrainFrom3D, models/utils.py:211-412).
// Part of the device code of libfcamd (translation unit: ../fcamd_kernels.hip, which holds the kernels and launchers).
#pragma once
#include "tile_io.h"
#include "tangent_writers.h"

namespace fcamd {

// --- the reference's 3D -> 1D/2D wrappers, fused (VonMises3D, comfe-rs Mises) ---------------------
// UniaxialStrainFrom3D / PlaneStrainFrom3D (models/utils.py:211-412) copy the mapped components of the
// low-dimensional gradient and stress into cached 3-D arrays, call the 3-D law, and copy the mapped
// components of stress and tangent back.  The cached 3-D stress keeps its unmapped components from
// the previous call (utils.py:253-266: under uniaxial strain the lateral stresses live only there).
// Here one kernel does all of it: per tile it reads the low-dimensional gradient and stress plus the
// cached 3-D stress row, runs the law's update on the padded point, and writes the full row back to
// the cache and the mapped components to the caller's arrays.  No 3-D gradient or tangent array
// exists.  WRAP = 1: component 11 of everything; WRAP = 2: gradient (0,1,2,3) -> (0,1,3,4), Mandel
// components 0..3, tangent block [0:4, 0:4] (utils.py:282-297, 377-412).  In place only (the wrappers
// have no out-of-place form).  The per-point arithmetic is the 3-D tiles' own (vm_trial / vm_return / vm_stress,
// cm_point, dp_trial / dp_return); tests/test_gpu_wrappers.py holds wrapper and 3-D law to bit equality.

// inputs of a wrapped tile: padded gradient g[9] and the 3-D stress row s[6] (cache + mapped components)
template <int WRAP, bool FULL, bool NT>
__device__ __forceinline__ void wrapped_load(ArgsRef a, double* region, long long p0, int npts, int lane,
                                             double (&g)[9], double (&s)[6]) {
    constexpr int LD = WRAP == 1 ? 1 : 4;  // doubles per point of the low-dimensional gradient and stress
    const bool live = FULL || lane < npts;
    Chunks<6> cc;
    tile_load<6, FULL, NT>(cc, a.cache3d + p0 * 6, npts * 6, lane);
#pragma unroll
    for (int i = 0; i < 9; ++i) g[i] = 0.0;
    double s_lo[LD];
    if constexpr (WRAP == 1) {
        g[0] = live ? a.grad[p0 + lane] : 0.0;
        s_lo[0] = live ? a.stress_in[p0 + lane] : 0.0;
    } else {
        Chunks<LD> cg, cs;
        tile_load<LD, FULL, NT>(cg, a.grad + p0 * LD, npts * LD, lane);
        tile_load<LD, FULL, NT>(cs, a.stress_in + p0 * LD, npts * LD, lane);
        double g_lo[LD];
        transpose_in<LD>(cg, region, lane, g_lo);
        transpose_in<LD>(cs, region, lane, s_lo);
        g[0] = g_lo[0], g[1] = g_lo[1], g[3] = g_lo[2], g[4] = g_lo[3];
    }
    transpose_in<6>(cc, region, lane, s);
#pragma unroll
    for (int i = 0; i < LD; ++i) s[i] = s_lo[i];  // mapped components come from the caller, the others persist
}

// the full 3-D row goes back to the wrapper's cache, the mapped components to the caller
template <int WRAP, bool FULL, bool NT>
__device__ __forceinline__ void wrapped_store_stress(ArgsRef a, double* region, long long p0, int npts,
                                                     int lane, const double (&s)[6]) {
    transpose_out<6, FULL, NT>(s, region, lane, a.cache3d + p0 * 6, npts * 6);
    if constexpr (WRAP == 1) {
        if (FULL || lane < npts) a.stress_out[p0 + lane] = s[0];
    } else {
        double s_lo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) s_lo[i] = s[i];
        transpose_out<4, FULL, NT>(s_lo, region, lane, a.stress_out + p0 * 4, npts * 4);
    }
}

// mapped block of the Mises tangents, entries formed exactly as tangent_mises does
template <bool COMFE, int WRAP, bool FULL, bool NT>
__device__ __forceinline__ void wrapped_tangent_mises(ArgsRef a, const Tables* T, double* region,
                                                      long long p0, int npts, int lane, double B, double C,
                                                      const double (&N)[6]) {
    if constexpr (WRAP == 1) {
        if (FULL || lane < npts)
            a.tangent[p0 + lane] = COMFE ? (T->a[0] + B * T->b[0]) + (C * N[0]) * N[0]
                                         : (T->a[0] + B * T->b[0]) + C * (N[0] * N[0]);
    } else {
        publish_tangent_params(region, lane, B, C, N);
        wave_sync();
        const int nchunks = npts * 8;  // block [0:4, 0:4]: 16 doubles = 8 chunks per point
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = k * kWave + lane;
            const int p = q >> 3, r = q & 7;
            const int i = r >> 1, j = 2 * (r & 1);
            const double* t = region + 10 * p;
            const d2 bc = reinterpret_cast<const d2*>(t)[0];
            const double ni = t[2 + i];
            const d2 nj = *reinterpret_cast<const d2*>(t + 2 + j);
            const d2 ta = *reinterpret_cast<const d2*>(T->a + 6 * i + j);
            const d2 tb = *reinterpret_cast<const d2*>(T->b + 6 * i + j);
            d2 v;
            if constexpr (COMFE) {
                v.x = (ta.x + bc.x * tb.x) + (bc.y * nj.x) * ni;
                v.y = (ta.y + bc.x * tb.y) + (bc.y * nj.y) * ni;
            } else {
                v.x = (ta.x + bc.x * tb.x) + bc.y * (ni * nj.x);
                v.y = (ta.y + bc.x * tb.y) + bc.y * (ni * nj.y);
            }
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

// mapped block of a point-independent tangent table (LE): [0][0] or the block [0:4, 0:4]
template <int WRAP, bool FULL, bool NT>
__device__ __forceinline__ void wrapped_tangent_const(ArgsRef a, const double* tab, long long p0, int npts,
                                                      int lane) {
    if constexpr (WRAP == 1) {
        if (FULL || lane < npts) a.tangent[p0 + lane] = tab[0];
    } else {
        const int nchunks = npts * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = k * kWave + lane;
            const int r = q & 7;
            const d2 v = *reinterpret_cast<const d2*>(tab + 6 * (r >> 1) + 2 * (r & 1));
            if constexpr (FULL) {
                store16<NT>(a.tangent + p0 * 16 + 2 * q, v);
            } else if (q < nchunks) {
                a.tangent[p0 * 16 + 2 * q] = v.x;
                a.tangent[p0 * 16 + 2 * q + 1] = v.y;
            }
        }
    }
}

}  // namespace fcamd
