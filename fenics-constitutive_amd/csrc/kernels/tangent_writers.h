// Tangent writers (288 of the 456-648 bytes per point): constant tangents streamed from an LDS table, the Mises tangents rebuilt
// per 16-byte chunk from 8 doubles per point; the protocol flags of EvalArgs::flags.
// Part of the device code of libfcamd (translation unit: ../fcamd_kernels.hip, which holds the kernels and launchers).
#pragma once
#include "tile_io.h"

namespace fcamd {

// ---------------------------------------------------------------------------------------
// tangent writers
// ---------------------------------------------------------------------------------------

// Constant tangent (LE, SLS, comfe LE): every point gets the same 36 doubles = 18 chunks,
// read from the LDS table `tab` (np.tile(D.flatten(), n) in the reference).
template <bool IDX, bool FULL, bool NT>
__device__ __forceinline__ void tangent_const(const double* tab, double* tangent, long long p0,
                                              const int* rows_lds, int npts, int lane,
                                              int r0 /* lane % 18 */) {
    const int nchunks = npts * 18;
#pragma unroll
    for (int k = 0; k < 18; ++k) {
        int r = r0 + (10 * k) % 18;
        r = r >= 18 ? r - 18 : r;
        const int q = k * kWave + lane;
        d2 v = reinterpret_cast<const d2*>(tab)[r];
        if (FULL || q < nchunks) store_tangent16<NT>(tangent_chunk<IDX>(tangent, p0, q, rows_lds), v);
    }
}

// The same for the points in `tneed` only (elastic tiles of the Drucker-Prager laws under the
// sparse-tangent protocol: rows of formerly plastic points get the elastic tangent back).
template <bool IDX, bool FULL, bool NT>
__device__ __forceinline__ void tangent_const_masked(const double* tab, double* tangent, long long p0,
                                                     const int* rows_lds, int npts, int lane,
                                                     unsigned long long tneed) {
    const int nchunks = npts * 18;
#pragma unroll
    for (int k = 0; k < 18; ++k) {
        const int q = k * kWave + lane;
        const int p = q / 18;
        d2 v = reinterpret_cast<const d2*>(tab)[q - 18 * p];
        const bool wanted = (IDX || !FULL) ? ((tneed >> p) & 1ull) != 0ull : tangent_granule_live(tneed, q);
        if ((FULL || q < nchunks) && wanted) store_tangent16<NT>(tangent_chunk<IDX>(tangent, p0, q, rows_lds), v);
    }
}

// Sparse-tangent protocol (EvalArgs::flags & kFlagSparseTangent, with the sparse trial-history protocol):
// the caller owns the tangent array across evaluates and it holds the tangent of the PREVIOUS evaluate
// of this state.  The tangent of an elastic point is one constant for all points and calls, so a row has
// to be written only if its point is plastic now (new tangent) or was plastic at the previous evaluate
// (back to the elastic tangent) -- the same `mask | m_old` set as the history rows.  Rows of points that
// stay elastic, 288 of their 464 bytes, are not touched.  Ragged last tiles are written in full.
constexpr int kFlagSparseTangent = 1;
// Split history of the laws whose reference layout is one [alpha, eps_p(6)] row per point (comfe-rs Mises and
// Drucker-Prager): h0 = the scalar (n doubles), h1 = the plastic-strain rows (6 n).  eps_p is write-only for the
// stress update (it only accumulates), the scalar is needed by every point (Mises: it enters the yield function) or
// by the plastic ones (Drucker-Prager); in the 7-double rows every point pays 56 bytes of history reads for it.
// A layout of device-resident states only (ResidentState), never of the interface arrays.
constexpr int kFlagSplitHistory = 4;
// Packed plastic-strain history (kernels/history_rows.h: PackedRows): committed and trial plastic-strain arrays hold, per tile,
// the rows of the ever-plastic points as one contiguous run; EvalArgs::emask_in / emask_out are the tiles' EVER masks.
constexpr int kFlagPackedHistory = 8;
template <bool FULL>
__device__ __forceinline__ unsigned long long sparse_tangent_need(ArgsRef a, unsigned long long need) {
    return (FULL && (a.flags & kFlagSparseTangent) != 0 && a.hmask != nullptr) ? need : ~0ull;
}

// The tangent chunks of a tile are computed and stored in groups: the scheduler may interleave the LDS
// reads, the arithmetic and the stores of one group, not across groups (bounds the register pressure).
constexpr int kTangentGroup = 3;

// Point-dependent tangent of the two Mises laws.  Lane p has published
//   tp[10p + 0] = B, tp[10p + 1] = C, tp[10p + 2 .. 7] = N   (stride 10: conflict-free b128)
// and the tile's tangent is   T[p][i][j] = (ta[i][j] + B * tb[i][j]) + third(i, j)  with
//   VonMises3D:   third = C * (N_i * N_j)      (aah, mises_plasticity_isotropic_hardening.py:170-175)
//   comfe Mises:  third = (C * N_j) * N_i      (column-major .data.0 of ((2 mu theta_bar) n) n^T,
//                                               mises_plasticity.rs:118-123)
// `tneed`: the points of the tile whose tangent rows are written (all ones unless the caller runs the
// sparse-tangent protocol, see sparse_tangent_need()).
template <bool COMFE, bool IDX, bool FULL, bool NT>
__device__ __forceinline__ void tangent_mises(const double* tp, const double* ta, const double* tb,
                                              double* tangent, long long p0, const int* rows_lds,
                                              int npts, int lane, unsigned long long tneed) {
    const int nchunks = npts * 18;
#pragma unroll
    for (int k = 0; k < 18; ++k) {
        const int q = k * kWave + lane;
        const int p = q / 18;
        const int r = q - 18 * p;
        const int i = r / 3;
        const int j = 2 * (r - 3 * i);
        const double* t = tp + 10 * p;
        const d2 bc = reinterpret_cast<const d2*>(t)[0];
        const double ni = t[2 + i];
        const d2 nj = *reinterpret_cast<const d2*>(t + 2 + j);
        const d2 a = *reinterpret_cast<const d2*>(ta + 6 * i + j);
        const d2 b = *reinterpret_cast<const d2*>(tb + 6 * i + j);
        d2 v;
        if constexpr (COMFE) {
            v.x = (a.x + bc.x * b.x) + (bc.y * nj.x) * ni;
            v.y = (a.y + bc.x * b.y) + (bc.y * nj.y) * ni;
        } else {
            v.x = (a.x + bc.x * b.x) + bc.y * (ni * nj.x);
            v.y = (a.y + bc.x * b.y) + bc.y * (ni * nj.y);
        }
        const bool wanted = (IDX || !FULL) ? ((tneed >> p) & 1ull) != 0ull : tangent_granule_live(tneed, q);
        if ((FULL || q < nchunks) && wanted) store_tangent16<NT>(tangent_chunk<IDX>(tangent, p0, q, rows_lds), v);
        // bound the register pressure: let the scheduler interleave at most 3 chunks
        if (k % kTangentGroup == kTangentGroup - 1) __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ void publish_tangent_params(double* region, int lane, double B, double C,
                                                       const double (&N)[6]) {
    double* t = region + 10 * lane;
    d2 v;
    v.x = B;
    v.y = C;
    reinterpret_cast<d2*>(t)[0] = v;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        v.x = N[2 * i];
        v.y = N[2 * i + 1];
        reinterpret_cast<d2*>(t)[1 + i] = v;
    }
}

}  // namespace fcamd
