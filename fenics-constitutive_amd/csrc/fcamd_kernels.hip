// HIP kernels (gfx950 / CDNA4) for the quadrature-point constitutive update.
//
// Execution model
//   * one quadrature point per lane, one 64-point tile per wavefront iteration,
//     grid-stride (tiles are dealt round-robin to all wavefronts of the grid);
//   * the caller's arrays stay in the reference's point-major AoS layout in HBM.  Every
//     global access is a wave-contiguous 16-byte-per-lane stream (1 KiB per instruction);
//     the AoS<->lane transposition happens in a small wave-private LDS region, so no
//     workgroup barrier is needed inside the tile loop;
//   * the 36-double tangent (288 of the 456-648 bytes per point) is never staged as a
//     matrix: laws with a constant tangent stream it from an LDS-resident table; the
//     plasticity laws publish 8 doubles per point (two scalars + the flow direction) to LDS
//     and every lane rebuilds the two tangent entries of the 16-byte chunk it stores;
//   * material constants are pre-computed on the host in the reference's expression order
//     and reach the kernel as kernel arguments; the 6x6 tables are staged into LDS once per
//     workgroup;
//   * the elastic/plastic branch is taken per wavefront from a 64-bit ballot: fully elastic
//     tiles skip the Newton iteration and (in place) the plastic-strain traffic.
//
// Arithmetic follows the reference operation by operation (compiled with
// -ffp-contract=off; fma() only where NumPy/OpenBLAS use one: the n x 6 . 6 x 6 products and
// the 6-term dot product, both verified to be ascending-k FMA chains).
//
// Reference algorithms restated here (never copied):
//   strain:   src/fenics_constitutive/models/utils.py:187-208, comfe-rs/src/mandel.rs:143-171
//   LE:       models/linear_elasticity_model.py:26-45
//   VonMises: models/mises_plasticity_isotropic_hardening.py:57-175
//   Maxwell:  models/spring_maxwell_model.py:40-88      Kelvin: models/spring_kelvin_model.py:43-88
//   comfe LE: comfe-rs/src/linear_elasticity.rs:49-74   comfe Mises: comfe-rs/src/mises_plasticity.rs:58-126
#include "fcamd_internal.h"

#include <cstdlib>

namespace fcamd {

typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kBlock = kWave * kWavesPerBlock;
// wave-private LDS region: 64 points x 14 doubles (largest user: Drucker-Prager tangent parameters)
constexpr int kRegionDoubles = 64 * 14;

// ---------------------------------------------------------------------------------------
// wave-level helpers
// ---------------------------------------------------------------------------------------

// Order LDS traffic between the lanes of one wavefront.  The hardware executes a wave's DS
// instructions in order; this only pins the compiler.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool NT>
__device__ __forceinline__ d2 load16(const double* p) {
    if constexpr (NT)
        return __builtin_nontemporal_load(reinterpret_cast<const d2*>(p));
    else
        return *reinterpret_cast<const d2*>(p);
}
template <bool NT>
__device__ __forceinline__ void store16(double* p, d2 v) {
    if constexpr (NT)
        __builtin_nontemporal_store(v, reinterpret_cast<d2*>(p));
    else
        *reinterpret_cast<d2*>(p) = v;
}

// Store policy of the tangent stream (288 of the 456-648 bytes per point), a build-time experiment knob
// (tools/ab_lib.py A/Bs two builds in one process): 0 = non-temporal (ships: +13 % over plain stores in round 1; `sc1` /
// `sc0 sc1` write-through stores, which drop the line from the L2 at once, measured no better in round 2).
#ifndef FCAMD_TANGENT_STORE
#define FCAMD_TANGENT_STORE 0
#endif
template <bool NT>
__device__ __forceinline__ void store_tangent16(double* p, d2 v) {
#if FCAMD_TANGENT_STORE == 1
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#elif FCAMD_TANGENT_STORE == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
#elif FCAMD_TANGENT_STORE == 3
    *reinterpret_cast<d2*>(p) = v;
#elif FCAMD_TANGENT_STORE == 4
    asm volatile("global_store_dwordx4 %0, %1, off nt sc1" ::"v"(p), "v"(v) : "memory");
#else
    store16<NT>(p, v);
#endif
}

// A 64-point tile of an AoS array with NC doubles per point is 32*NC contiguous 16-byte
// chunks; lane l owns chunks l, l+64, ...  (K = ceil(NC/2) per lane, the last one only on
// lanes < 32 when NC is odd).
template <int NC>
struct Chunks {
    static constexpr int K = (NC + 1) / 2;
    d2 v[K];
};

// chunk k of lane `lane` exists: always, except the odd last half-instruction of odd NC
template <int NC>
__device__ __forceinline__ bool chunk_live(int k, int lane) {
    return (2 * (k + 1) <= NC) || lane < 32;
}

// global -> registers.  FULL: whole tile, 16-byte loads.  Otherwise (last, ragged tile):
// guarded 8-byte loads of the `nelem` valid doubles.
template <int NC, bool FULL, bool NT>
__device__ __forceinline__ void tile_load(Chunks<NC>& c, const double* src, int nelem, int lane) {
#pragma unroll
    for (int k = 0; k < Chunks<NC>::K; ++k) {
        const int q = k * kWave + lane;
        if constexpr (FULL) {
            if (chunk_live<NC>(k, lane)) c.v[k] = load16<NT>(src + 2 * q);
        } else {
            const int e = 2 * q;
            c.v[k].x = e < nelem ? src[e] : 0.0;
            c.v[k].y = e + 1 < nelem ? src[e + 1] : 0.0;
        }
    }
}

// registers -> LDS (linear image of the tile)
template <int NC>
__device__ __forceinline__ void tile_to_lds(const Chunks<NC>& c, double* lds, int lane) {
#pragma unroll
    for (int k = 0; k < Chunks<NC>::K; ++k) {
        const int q = k * kWave + lane;
        if (chunk_live<NC>(k, lane)) reinterpret_cast<d2*>(lds)[q] = c.v[k];
    }
}

// registers -> global
template <int NC, bool FULL, bool NT>
__device__ __forceinline__ void tile_store(const Chunks<NC>& c, double* dst, int nelem, int lane) {
#pragma unroll
    for (int k = 0; k < Chunks<NC>::K; ++k) {
        const int q = k * kWave + lane;
        if constexpr (FULL) {
            if (chunk_live<NC>(k, lane)) store16<NT>(dst + 2 * q, c.v[k]);
        } else {
            const int e = 2 * q;
            if (e < nelem) dst[e] = c.v[k].x;
            if (e + 1 < nelem) dst[e + 1] = c.v[k].y;
        }
    }
}

// LDS (linear image) -> global
template <int NC, bool FULL, bool NT>
__device__ __forceinline__ void lds_to_global(const double* lds, double* dst, int nelem, int lane) {
    Chunks<NC> c;
#pragma unroll
    for (int k = 0; k < Chunks<NC>::K; ++k) {
        const int q = k * kWave + lane;
        if (chunk_live<NC>(k, lane)) c.v[k] = reinterpret_cast<const d2*>(lds)[q];
    }
    tile_store<NC, FULL, NT>(c, dst, nelem, lane);
}

// per-lane view of the LDS image: the NC doubles of this lane's point
template <int NC>
__device__ __forceinline__ void lds_get_point(const double* lds, int lane, double (&x)[NC]) {
    const double* p = lds + lane * NC;
    if constexpr (NC % 2 == 0) {
#pragma unroll
        for (int i = 0; i < NC / 2; ++i) {
            d2 v = reinterpret_cast<const d2*>(p)[i];
            x[2 * i] = v.x;
            x[2 * i + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NC; ++i) x[i] = p[i];
    }
}
template <int NC>
__device__ __forceinline__ void lds_put_point(double* lds, int lane, const double (&x)[NC]) {
    double* p = lds + lane * NC;
    if constexpr (NC % 2 == 0) {
#pragma unroll
        for (int i = 0; i < NC / 2; ++i) {
            d2 v;
            v.x = x[2 * i];
            v.y = x[2 * i + 1];
            reinterpret_cast<d2*>(p)[i] = v;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NC; ++i) p[i] = x[i];
    }
}

// AoS tile -> per-lane point values, through the wave's LDS region.
template <int NC>
__device__ __forceinline__ void transpose_in(const Chunks<NC>& c, double* region, int lane,
                                             double (&x)[NC]) {
    tile_to_lds<NC>(c, region, lane);
    wave_sync();
    lds_get_point<NC>(region, lane, x);
    wave_sync();
}
// per-lane point values -> AoS tile in global memory, through the wave's LDS region.
template <int NC, bool FULL, bool NT>
__device__ __forceinline__ void transpose_out(const double (&x)[NC], double* region, int lane,
                                              double* dst, int nelem) {
    lds_put_point<NC>(region, lane, x);
    wave_sync();
    lds_to_global<NC, FULL, NT>(region, dst, nelem, lane);
    wave_sync();
}

// ---------------------------------------------------------------------------------------
// Where the stress / tangent rows of a tile live.
//   IDX = false: the law's own arrays, point p0 + lane at row p0 + lane (contiguous tile).
//   IDX = true : rows of PARENT arrays, point p0 + lane at row a.rows[p0 + lane] -- the submesh
//                gather/scatter of the reference (solver/maps.py:82-123, "parent_array[parent] =
//                sub_array[sub]") folded into the kernel's addressing: every lane loads and stores
//                its own 48-byte stress row, and the tangent writers look the row of each chunk's
//                point up in a per-wave LDS table.
// ---------------------------------------------------------------------------------------
// Base pointers of the stress / tangent arrays as seen by one tile.  Normally the kernel arguments;
// for a tile of the indexed kernel whose 64 parent rows are consecutive they are shifted by
// (row0 - p0) rows, so that the coalesced (non-indexed) tile body addresses the parent rows directly.
struct StressBases {
    const double* sin;
    double* sout;
    double* tan;
    double* sout2 = nullptr;  // second copy of the stress rows (EvalArgs::stress_out2; contiguous tiles only)
};

template <bool IDX, bool FULL, bool NT>
struct StressRows {
    Chunks<6> c;
    long long row = 0;

    __device__ __forceinline__ void load(const EvalArgs& a, const StressBases& sb, long long p0, int npts, int lane,
                                         int* rows_lds) {
        if constexpr (IDX) {
            const bool live = FULL || lane < npts;
            row = live ? (long long)a.rows[p0 + lane] : 0ll;
            rows_lds[lane] = (int)row;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                d2 z;
                z.x = 0.0;
                z.y = 0.0;
                c.v[k] = live ? load16<NT>(sb.sin + row * 6 + 2 * k) : z;
            }
        } else {
            tile_load<6, FULL, NT>(c, sb.sin + p0 * 6, npts * 6, lane);
        }
    }
    __device__ __forceinline__ void get(double* region, int lane, double (&s)[6]) {
        if constexpr (IDX) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                s[2 * k] = c.v[k].x;
                s[2 * k + 1] = c.v[k].y;
            }
        } else {
            transpose_in<6>(c, region, lane, s);
        }
    }
    __device__ __forceinline__ void put(const StressBases& sb, double* region, int lane, const double (&s)[6],
                                        long long p0, int npts) {
        if constexpr (IDX) {
            if (FULL || lane < npts) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    d2 v;
                    v.x = s[2 * k];
                    v.y = s[2 * k + 1];
                    store16<NT>(sb.sout + row * 6 + 2 * k, v);
                    if (sb.sout2 != nullptr) store16<NT>(sb.sout2 + row * 6 + 2 * k, v);
                }
            }
        } else if (sb.sout2 == nullptr) {
            transpose_out<6, FULL, NT>(s, region, lane, sb.sout + p0 * 6, npts * 6);
        } else {  // resident state on the device + the host assembler's array: one LDS image, two streams
            lds_put_point<6>(region, lane, s);
            wave_sync();
            lds_to_global<6, FULL, NT>(region, sb.sout + p0 * 6, npts * 6, lane);
            lds_to_global<6, FULL, NT>(region, sb.sout2 + p0 * 6, npts * 6, lane);
            wave_sync();
        }
    }
};

// destination of tangent chunk q (= 16 bytes) of the tile
template <bool IDX>
__device__ __forceinline__ double* tangent_chunk(double* tangent, long long p0, int q, const int* rows_lds) {
    if constexpr (IDX) {
        const int p = q / 18;
        return tangent + (long long)rows_lds[p] * 36 + 2 * (q - 18 * p);
    } else {
        return tangent + p0 * 36 + 2 * q;
    }
}

// Mandel strain increment from the row-major 3x3 displacement-gradient increment.
__device__ __forceinline__ void mandel_strain(const double (&g)[9], double f, double (&e)[6]) {
    e[0] = g[0];
    e[1] = g[4];
    e[2] = g[8];
    e[3] = f * (g[1] + g[3]);
    e[4] = f * (g[2] + g[6]);
    e[5] = f * (g[5] + g[7]);
}

// y_i = sum_k x_k * M[k][i] as an ascending-k FMA chain (what OpenBLAS dgemm does for the
// reference's "strain.reshape(-1, 6) @ D"); M is an LDS-resident row-major 6x6 table.
__device__ __forceinline__ void row_times_matrix_fma(const double (&x)[6], const double* M,
                                                     double (&y)[6]) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double acc = x[0] * M[i];
#pragma unroll
        for (int k = 1; k < 6; ++k) acc = __builtin_fma(x[k], M[6 * k + i], acc);
        y[i] = acc;
    }
}

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
        if ((FULL || q < nchunks) && ((tneed >> p) & 1ull)) store_tangent16<NT>(tangent_chunk<IDX>(tangent, p0, q, rows_lds), v);
    }
}

// Sparse-tangent protocol (EvalArgs::flags & kFlagSparseTangent, with the sparse trial-history protocol):
// the caller owns the tangent array across evaluates and it holds the tangent of the PREVIOUS evaluate
// of this state.  The tangent of an elastic point is one constant for all points and calls, so a row has
// to be written only if its point is plastic now (new tangent) or was plastic at the previous evaluate
// (back to the elastic tangent) -- the same `mask | m_old` set as the history rows.  Rows of points that
// stay elastic, 288 of their 464 bytes, are not touched.  Ragged last tiles are written in full.
constexpr int kFlagSparseTangent = 1;
// Delta trial history (FCAMD_EVAL_DELTA_HISTORY; VonMises3D under the sparse protocol): eps_n is write-only with
// respect to the stress update, so the trial array need not hold eps_n + gamma N -- it receives the INCREMENT
// gamma N at the plastic points (and is not defined elsewhere), the committed rows are not read at all
// (28 of the 156 bytes read per point on the 22 % mixture: -3.5 % kernel time), and the commit adds the increments
// of the plastic points to the committed array (commit_delta_kernel, once per increment instead of once per
// Newton iteration).  alpha is not affected (it enters the yield function and is read for every point anyway).
constexpr int kFlagDeltaHistory = 2;
// Split history of the laws whose reference layout is one [alpha, eps_p(6)] row per point (comfe-rs Mises and
// Drucker-Prager): h0 = the scalar (n doubles), h1 = the plastic-strain rows (6 n).  eps_p is write-only for the
// stress update (it only accumulates), the scalar is needed by every point (Mises: it enters the yield function) or
// by the plastic ones (Drucker-Prager); in the 7-double rows every point pays 56 bytes of history reads for it.
// A layout of device-resident states only (ResidentState), never of the interface arrays.
constexpr int kFlagSplitHistory = 4;
template <bool FULL>
__device__ __forceinline__ unsigned long long sparse_tangent_need(const EvalArgs& a, unsigned long long need) {
    return (FULL && (a.flags & kFlagSparseTangent) != 0 && a.hmask != nullptr) ? need : ~0ull;
}

// The tangent chunks of a tile are computed and stored in groups: the scheduler may interleave the LDS
// reads, the arithmetic and the stores of one group, not across groups (bounds the register pressure).
#ifndef FCAMD_TANGENT_GROUP
#define FCAMD_TANGENT_GROUP 3
#endif
constexpr int kTangentGroup = FCAMD_TANGENT_GROUP;

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
        if ((FULL || q < nchunks) && ((tneed >> p) & 1ull)) store_tangent16<NT>(tangent_chunk<IDX>(tangent, p0, q, rows_lds), v);
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

// ---------------------------------------------------------------------------------------
// per-wave statistics
// ---------------------------------------------------------------------------------------
struct WaveStats {
    unsigned long long nonconv = 0, plastic = 0, iters = 0, domain = 0;
};

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

// ---------------------------------------------------------------------------------------
// tile bodies, one per law.  `region` is the wave's LDS scratch, `T` the staged tables.
// ---------------------------------------------------------------------------------------

// --- LinearElasticityModel: sigma += d_eps @ D ; tangent = tile(D) ----------------------
template <bool IDX, bool FULL, bool NT>
__device__ __forceinline__ void tile_linear_elasticity(const EvalArgs& a, const StressBases& sb, const Tables* T,
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
    sr.put(sb, region, lane, s, p0, npts);
}

// --- comfe-rs LinearElasticity3D: sigma += C . d_eps (column axpy, no FMA) ---------------
template <bool IDX, bool FULL, bool NT>
__device__ __forceinline__ void tile_comfe_le(const EvalArgs& a, const StressBases& sb, const Tables* T, double* region,
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
    sr.put(sb, region, lane, s, p0, npts);
}

// --- SLS Maxwell / Kelvin -----------------------------------------------------------------
// scalars: s[0]=strain factor, s[1]=1/factor, s[2]=1/(tau*2*mu1), s[3]=1/tau,
//          Maxwell: s[4]=2*mu1 ; Kelvin: s[4]=2*mu0, s[5]=mu0/(tau*mu1), s[6]=lam0/(tau*2*mu1)
// tables:  Maxwell a=D1, b=D0+D1, c=tangent ; Kelvin a=D0, c=tangent
template <bool KELVIN, bool IDX, bool FULL, bool NT>
__device__ __forceinline__ void tile_sls(const EvalArgs& a, const StressBases& sb, const Tables* T, double* region,
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
    sr.put(sb, region, lane, s, p0, npts);
    transpose_out<6, FULL, NT>(ev, region, lane, a.h0_out + p0 * 6, npts * 6);
    transpose_out<6, FULL, NT>(en, region, lane, a.h1_out + p0 * 6, npts * 6);
}

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

__device__ __forceinline__ void vm_trial(const Scalars& sc, const double (&e)[6], const double (&s)[6], double alpha_n,
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

__device__ __forceinline__ void vm_return(const Scalars& sc, const VMTrial& t, double alpha_n, VMReturn& r, WaveStats& st) {
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
__device__ __forceinline__ void vm_stress(const Scalars& sc, const VMTrial& t, const VMReturn& r, double (&s)[6]) {
    const double kt = sc.s[1] * t.tr_eps, tmg = sc.s[2] * r.gamma;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double vol = i < 3 ? kt : kt * 0.0;
        s[i] = s[i] + ((vol + t.dsig[i]) - tmg * r.N[i]);
    }
}
__device__ __forceinline__ void vm_tangent_coefficients(const Scalars& sc, const VMReturn& r, double& B, double& C) {
    const double two_mu = sc.s[2], four_mu2 = sc.s[9];
    B = two_mu * (1.0 - two_mu * r.xc2);
    C = four_mu2 * (r.xc2 - r.xc1);
}

// comfe-rs MisesPlasticity3D (mises_plasticity.rs:58-126): the whole update of one point.  In: e, s (sigma_n),
// h = [alpha, eps_p(6)].  Out: s (total stress), h (updated if plastic), tangent parameters B, sc2 and the
// (non-unit) flow direction nv.  Returns whether the point is plastic.
__device__ __forceinline__ bool cm_point(const Scalars& sc, bool live, const double (&e)[6], double (&s)[6], double (&h)[7],
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

// --- VonMises3D: J2 plasticity, saturation hardening, scalar Newton per point -----------------
// scalars: s[0]=strain factor, s[1]=ka, s[2]=2*mu, s[3]=sqrt(2/3), s[4]=y0, s[5]=y00-y0,
//          s[6]=-w, s[7]=(-2)*mu, s[8]=((2/3)*(y00-y0))*w, s[9]=(4*mu)*mu
// tables:  a = ka*xioi, b = xpp
template <bool IDX, bool SPARSE, bool FULL, bool NT>
__device__ __forceinline__ void tile_von_mises(const EvalArgs& a, const StressBases& sb, const Tables* T, double* region,
                                               int* rows_lds, long long p0, int npts, int lane,
                                               WaveStats& st) {
    Chunks<9> cg;
    StressRows<IDX, FULL, NT> sr;
    tile_load<9, FULL, NT>(cg, a.grad + p0 * 9, npts * 9, lane);
    sr.load(a, sb, p0, npts, lane, rows_lds);
    const bool live = FULL || lane < npts;
    const double alpha_n = live ? a.h1_in[p0 + lane] : 0.0;
    const bool hist_in_place = (a.h0_in == a.h0_out) && (a.h1_in == a.h1_out);

    double g[9], s[6], e[6];
    transpose_in<9>(cg, region, lane, g);
    sr.get(region, lane, s);
    mandel_strain(g, a.sc.s[0], e);

    VMTrial tr;
    vm_trial(a.sc, e, s, alpha_n, tr);
    const bool plastic = live && (tr.phitr > 0.0);
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
    constexpr bool sparse = SPARSE;
    unsigned long long m_old = 0ull;
    if constexpr (sparse) m_old = a.hmask[p0 >> 6];
    const unsigned long long need_mask = mask | m_old;
    // delta trial history: only the rows of points that are plastic NOW are written (their increment), nothing is read
    const bool delta = sparse && (a.flags & kFlagDeltaHistory) != 0;
    const unsigned long long eps_mask = delta ? mask : need_mask;
    const bool masked = FULL && (sparse || hist_in_place) && ((int)__popcll(eps_mask) <= a.masked_max);
    const bool touch_eps = masked ? (eps_mask != 0ull)
                                  : (sparse ? (eps_mask != 0ull) : ((mask != 0ull) || !hist_in_place));
    const bool touch_alpha = delta ? (need_mask != 0ull) : touch_eps;  // stale points get their committed alpha back
    bool row_live[3] = {true, true, true};  // per chunk of this lane: its row is touched
    if (masked) {
#pragma unroll
        for (int k = 0; k < 3; ++k) row_live[k] = ((eps_mask >> ((k * kWave + lane) / 3)) & 1ull) != 0ull;
    }
    if (delta) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            ce.v[k].x = 0.0;
            ce.v[k].y = 0.0;
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
        if (plastic) vm_return(a.sc, tr, alpha_n, rm, st);
        st.plastic += (lane == 0) ? (unsigned long long)__popcll(mask) : 0ull;
    }

    vm_stress(a.sc, tr, rm, s);
    sr.put(sb, region, lane, s, p0, npts);

    // history: eps_n += gamma N ; alpha += sqrt(2/3) gamma
    if (touch_eps) {
        if (mask != 0ull) {
            double ep[6];
            transpose_in<6>(ce, region, lane, ep);
#pragma unroll
            for (int i = 0; i < 6; ++i) ep[i] = ep[i] + rm.gamma * rm.N[i];
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
    // alpha: one coalesced 512-byte store per touched tile
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
        tangent_mises<false, IDX, FULL, NT>(region, T->a, T->b, sb.tan, p0, rows_lds, npts, lane, tneed);
        wave_sync();
    }
}

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
__device__ __forceinline__ void wrapped_load(const EvalArgs& a, double* region, long long p0, int npts, int lane,
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
__device__ __forceinline__ void wrapped_store_stress(const EvalArgs& a, double* region, long long p0, int npts,
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
__device__ __forceinline__ void wrapped_tangent_mises(const EvalArgs& a, const Tables* T, double* region,
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
__device__ __forceinline__ void wrapped_tangent_const(const EvalArgs& a, const double* tab, long long p0, int npts,
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

// LinearElasticityModel behind the wrappers (what the reference's own tests wrap, test_elasticity.py:206,278)
template <int WRAP, bool FULL, bool NT>
__device__ __forceinline__ void tile_linear_elasticity_wrapped(const EvalArgs& a, const Tables* T, double* region,
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

template <int WRAP, bool FULL, bool NT>
__device__ __forceinline__ void tile_von_mises_wrapped(const EvalArgs& a, const Tables* T, double* region,
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

    Chunks<6> ce;
    if (mask != 0ull) tile_load<6, FULL, NT>(ce, a.h0_in + p0 * 6, npts * 6, lane);
    VMReturn rm;
    if (mask != 0ull) {
        if (plastic) vm_return(a.sc, tr, alpha_n, rm, st);
        st.plastic += (lane == 0) ? (unsigned long long)__popcll(mask) : 0ull;
    }
    vm_stress(a.sc, tr, rm, s);
    wrapped_store_stress<WRAP, FULL, NT>(a, region, p0, npts, lane, s);
    if (mask != 0ull) {
        double ep[6];
        transpose_in<6>(ce, region, lane, ep);
#pragma unroll
        for (int i = 0; i < 6; ++i) ep[i] = ep[i] + rm.gamma * rm.N[i];
        transpose_out<6, FULL, NT>(ep, region, lane, a.h0_out + p0 * 6, npts * 6);
        if (live) a.h1_out[p0 + lane] = alpha_n + a.sc.s[3] * rm.gamma;
    }
    if (a.tangent) {
        double B, C;
        vm_tangent_coefficients(a.sc, rm, B, C);
        wrapped_tangent_mises<false, WRAP, FULL, NT>(a, T, region, p0, npts, lane, B, C, rm.N);
    }
}

template <int WRAP, bool FULL, bool NT>
__device__ __forceinline__ void tile_comfe_mises_wrapped(const EvalArgs& a, const Tables* T, double* region,
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

// History write policy of the laws with one [alpha, eps_p(6)] row per point (comfe-rs Mises and
// Drucker-Prager; the row is always READ: alpha enters the yield function).  Which rows change:
//   in place                      : the plastic points of this evaluate (ballot `mask`);
//   out of place                  : every row is copied;
//   out of place, sparse protocol : (a.hmask != nullptr, see tile_von_mises) the trial array equals the
//                                   committed one wherever the tile's mask word is clear, so the rows
//                                   of the points that are plastic now (new values) or were plastic at
//                                   the previous evaluate (stale: restore the committed values).
// Row-masked tile store: the tile keeps its four wave-wide, address-ordered store instructions and a
// lane skips the 16-byte chunks that lie entirely in untouched rows (chunk q holds doubles 2q and
// 2q + 1 of the tile image, i.e. parts of rows 2q / 7 and (2q + 1) / 7; a chunk straddling a touched
// and an untouched row rewrites 8 bytes of the latter with the value it already has).
// plastic | formerly plastic points of the tile under the sparse protocol; records the new ballot
__device__ __forceinline__ unsigned long long sparse_need(const EvalArgs& a, long long p0, unsigned long long mask,
                                                          int lane) {
    if (a.hmask == nullptr) return mask;
    const unsigned long long m_old = a.hmask[p0 >> 6];
    if (lane == 0 && mask != m_old) a.hmask[p0 >> 6] = mask;
    return mask | m_old;
}

// `touched`: sparse_need() of the tile
template <bool FULL, bool NT>
__device__ __forceinline__ void history7_store(const EvalArgs& a, long long p0, int npts, int lane,
                                               unsigned long long touched, bool hist_in_place, double* region,
                                               const double (&h)[7]) {
    const unsigned long long need = (a.hmask != nullptr || hist_in_place) ? touched : ~0ull;
    if (need == 0ull) return;
    if (!FULL || need == ~0ull || (int)__popcll(need) > a.masked_max) {
        transpose_out<7, FULL, NT>(h, region, lane, a.h0_out + p0 * 7, npts * 7);
        return;
    }
    lds_put_point<7>(region, lane, h);
    wave_sync();
#pragma unroll
    for (int k = 0; k < Chunks<7>::K; ++k) {
        const int q = k * kWave + lane;
        const bool hit = (((need >> ((2 * q) / 7)) | (need >> ((2 * q + 1) / 7))) & 1ull) != 0ull;
        if (chunk_live<7>(k, lane) && hit)
            store16<NT>(a.h0_out + p0 * 7 + 2 * q, reinterpret_cast<const d2*>(region)[q]);
    }
    wave_sync();
}

// Split history (kFlagSplitHistory): the scalar of every point of a touched tile and the rows `rows` of the
// plastic-strain array, rows_out = rows_in + delta (delta = 0 at points that are not plastic: they get their committed
// values back).  Which rows, by protocol, as in history7_store.  Row-masked access as in tile_von_mises (a 48-byte row
// is three 16-byte chunks of the tile image).
template <bool FULL, bool NT>
__device__ __forceinline__ void split_history_store(const EvalArgs& a, long long p0, int npts, int lane, unsigned long long mask,
                                                    unsigned long long touched, bool hist_in_place, double* region,
                                                    double scalar, const double (&delta)[6]) {
    const bool live = FULL || lane < npts;
    const unsigned long long rows = (a.hmask != nullptr || hist_in_place) ? touched : ~0ull;
    if (rows == 0ull) return;
    if (live) a.h0_out[p0 + lane] = scalar;  // one coalesced 512-byte store per touched tile
    const bool masked = FULL && rows != ~0ull && (int)__popcll(rows) <= a.masked_max;
    Chunks<6> ce;
    bool row_live[3] = {true, true, true};
    if (masked) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            row_live[k] = ((rows >> ((k * kWave + lane) / 3)) & 1ull) != 0ull;
            d2 z;
            z.x = 0.0;
            z.y = 0.0;
            ce.v[k] = row_live[k] ? load16<NT>(a.h1_in + p0 * 6 + 2 * (k * kWave + lane)) : z;
        }
    } else {
        tile_load<6, FULL, NT>(ce, a.h1_in + p0 * 6, npts * 6, lane);
    }
    if (mask != 0ull) {
        double ep[6];
        transpose_in<6>(ce, region, lane, ep);
#pragma unroll
        for (int i = 0; i < 6; ++i) ep[i] = ep[i] + delta[i];
        if (masked) {
            lds_put_point<6>(region, lane, ep);
            wave_sync();
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int q = k * kWave + lane;
                if (row_live[k]) store16<NT>(a.h1_out + p0 * 6 + 2 * q, reinterpret_cast<const d2*>(region)[q]);
            }
            wave_sync();
        } else {
            transpose_out<6, FULL, NT>(ep, region, lane, a.h1_out + p0 * 6, npts * 6);
        }
    } else if (masked) {  // only stale rows: restore the committed values
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (row_live[k]) store16<NT>(a.h1_out + p0 * 6 + 2 * (k * kWave + lane), ce.v[k]);
    } else if (!hist_in_place) {
        tile_store<6, FULL, NT>(ce, a.h1_out + p0 * 6, npts * 6, lane);
    }
}

// --- comfe-rs MisesPlasticity3D: linear hardening, closed-form radial return ---------------
// scalars: s[0]=strain factor (FRAC_1_SQRT_2), s[1]=mu, s[2]=kappa, s[3]=y_0, s[4]=h,
//          s[5]=2*mu, s[6]=3*mu+h, s[7]=sqrt(3/2), s[8]=3*mu, s[9]=1/(1+h/(3 mu))
// tables:  a = kappa*sym_id(x)sym_id, b = P_dev.   history field 0: [alpha, eps_p(6)] per point.
template <bool IDX, bool FULL, bool NT>
__device__ __forceinline__ void tile_comfe_mises(const EvalArgs& a, const StressBases& sb, const Tables* T, double* region,
                                                 int* rows_lds, long long p0, int npts, int lane,
                                                 WaveStats& st) {
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

    sr.put(sb, region, lane, s, p0, npts);
    const unsigned long long touched = sparse_need(a, p0, mask, lane);
    if (split) {
        const double d6[6] = {h[1], h[2], h[3], h[4], h[5], h[6]};
        split_history_store<FULL, NT>(a, p0, npts, lane, mask, touched, hist_in_place, region, h[0], d6);
    } else {
        history7_store<FULL, NT>(a, p0, npts, lane, touched, hist_in_place, region, h);
    }

    const unsigned long long tneed = sparse_tangent_need<FULL>(a, touched);
    if (sb.tan && tneed != 0ull) {
        publish_tangent_params(region, lane, B, sc2, nv);
        wave_sync();
        tangent_mises<true, IDX, FULL, NT>(region, T->a, T->b, sb.tan, p0, rows_lds, npts, lane, tneed);
        wave_sync();
    }
}



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

// T[i][j..j+1] for chunk q of the tile from the published parameters
template <bool IDX, bool FULL, bool NT>
__device__ __forceinline__ void tangent_dp(const double* tp, const double* t11tab, const double* pdtab,
                                           const double* etab, double* tangent, long long p0,
                                           const int* rows_lds, int npts, int lane, unsigned long long tneed) {
    const int nchunks = npts * 18;
#pragma unroll
    for (int k = 0; k < 18; ++k) {
        const int q = k * kWave + lane;
        const int p = q / 18;
        const int r = q - 18 * p;
        const int i = r / 3;
        const int j = 2 * (r - 3 * i);
        const double* t = tp + kDpStride * p;
        const d2 c0 = reinterpret_cast<const d2*>(t)[0];  // t11, tP
        const d2 c1 = reinterpret_cast<const d2*>(t)[1];  // tss, t1s
        const d2 c2 = reinterpret_cast<const d2*>(t)[2];  // ts1, plastic flag
        const double ts1 = c2.x;
        const double si = t[6 + i];
        const d2 sj = *reinterpret_cast<const d2*>(t + 6 + j);
        const d2 o = *reinterpret_cast<const d2*>(t11tab + 6 * i + j);  // (1 x 1)[i][j]
        const d2 pd = *reinterpret_cast<const d2*>(pdtab + 6 * i + j);
        const double oi = i < 3 ? 1.0 : 0.0;
        d2 v;
        v.x = (c0.x * o.x + c0.y * pd.x) + ((c1.x * si) * sj.x + (c1.y * oi) * sj.x + (ts1 * si) * (j < 3 ? 1.0 : 0.0));
        v.y = (c0.x * o.y + c0.y * pd.y) + ((c1.x * si) * sj.y + (c1.y * oi) * sj.y + (ts1 * si) * (j + 1 < 3 ? 1.0 : 0.0));
        // elastic points of a mixed tile: the reference returns elastic_tangent() itself (general.rs:131-135),
        // i.e. the host-computed 2 mu P_dev + 3 kappa P_vol bit for bit, not kappa 1x1 + 2 mu P_dev
        const d2 el = *reinterpret_cast<const d2*>(etab + 6 * i + j);
        if (c2.y == 0.0) v = el;
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
__device__ __forceinline__ void dp_trial(const Scalars& sc, const double (&e)[6], const double (&sig0)[6], DPTrial& t) {
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
__device__ __forceinline__ void dp_return(const Scalars& sc, const double (&e)[6], const double (&sig0)[6], DPTrial& t,
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

template <bool HYPER, bool IDX, bool FULL, bool NT>
__device__ __forceinline__ void tile_comfe_dp(const EvalArgs& a, const StressBases& sb, const Tables* T,
                                              double* region, int* rows_lds, long long p0, int npts, int lane,
                                              int r0, WaveStats& st) {
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

    if (mask == 0ull) {
        // fully elastic tile: stress = sigma_tr, tangent = E, history untouched
        sr.put(sb, region, lane, t.sig1, p0, npts);
        const unsigned long long touched = sparse_need(a, p0, 0ull, lane);
        if (split) {
            const double d6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            split_history_store<FULL, NT>(a, p0, npts, lane, 0ull, touched, hist_in_place, region, h[0], d6);
        } else {
            history7_store<FULL, NT>(a, p0, npts, lane, touched, hist_in_place, region, h);
        }
        const unsigned long long tneed = sparse_tangent_need<FULL>(a, touched);
        if (sb.tan && tneed != 0ull) {
            if constexpr (IDX) wave_sync();
            if (tneed == ~0ull)
                tangent_const<IDX, FULL, NT>(T->c, sb.tan, p0, rows_lds, npts, lane, r0);
            else
                tangent_const_masked<IDX, FULL, NT>(T->c, sb.tan, p0, rows_lds, npts, lane, tneed);
        }
        st.domain += (live && t.tip) ? 1ull : 0ull;
        return;
    }

    DPTangent tg;
    tg.t11 = a.sc.s[2], tg.tP = a.sc.s[7];
    if (plastic) dp_return<HYPER>(a.sc, e, sig0, t, h, tg, st);
    st.plastic += (lane == 0) ? (unsigned long long)__popcll(mask) : 0ull;
    st.domain += (live && t.tip) ? 1ull : 0ull;  // tip of the classic surface reached (reference: assert!)

    sr.put(sb, region, lane, t.sig1, p0, npts);
    const unsigned long long touched = sparse_need(a, p0, mask, lane);
    if (split) {
        const double d6[6] = {h[1], h[2], h[3], h[4], h[5], h[6]};
        split_history_store<FULL, NT>(a, p0, npts, lane, mask, touched, hist_in_place, region, h[0], d6);
    } else {
        history7_store<FULL, NT>(a, p0, npts, lane, touched, hist_in_place, region, h);
    }

    const unsigned long long tneed = sparse_tangent_need<FULL>(a, touched);
    if (sb.tan && tneed != 0ull) {
        dp_publish(region, lane, tg, t.s_tr, plastic);
        wave_sync();
        tangent_dp<IDX, FULL, NT>(region, T->a, T->b, T->c, sb.tan, p0, rows_lds, npts, lane, tneed);
        wave_sync();
    }
}

// fused 3D -> 1D/2D wrapper around the Drucker-Prager laws (see the Mises versions above)
template <bool HYPER, int WRAP, bool FULL, bool NT>
__device__ __forceinline__ void tile_comfe_dp_wrapped(const EvalArgs& a, const Tables* T, double* region,
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

template <int SD>
__device__ __forceinline__ void row_times_matrix_fma_n(const double (&x)[SD], const double* M,
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

template <int LAW, int DIMS, bool FULL, bool NT>
__device__ __forceinline__ void tile_lowdim(const EvalArgs& a, const Tables* T, double* region,
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
    double g[GD2], s[SD], e[SD], y[SD];
    transpose_in<GD2>(cg, region, lane, g);
    transpose_in<SD>(cs, region, lane, s);
    strain_lowdim<DIMS>(g, a.sc.s[0], e);
    if constexpr (LAW == LAW_LE) {
        row_times_matrix_fma_n<SD>(e, T->a, y);
#pragma unroll
        for (int i = 0; i < SD; ++i) s[i] = s[i] + y[i];
        transpose_out<SD, FULL, NT>(s, region, lane, a.stress_out + p0 * SD, npts * SD);
    } else {
        double ev[SD], en[SD], dv[SD];
        transpose_in<SD>(cv, region, lane, ev);
        transpose_in<SD>(cn, region, lane, en);
        const double inv_factor = a.sc.s[1], cA = a.sc.s[2], cB = a.sc.s[3], c2mu = a.sc.s[4];
        if constexpr (LAW == LAW_MAXWELL) {
            double x[SD];
#pragma unroll
            for (int i = 0; i < SD; ++i) x[i] = cA * (en[i] + e[i]);
            row_times_matrix_fma_n<SD>(x, T->a, y);
#pragma unroll
            for (int i = 0; i < SD; ++i) dv[i] = inv_factor * (y[i] - cB * ev[i]);
            row_times_matrix_fma_n<SD>(e, T->b, y);
        } else {
            const double cC = a.sc.s[5], cD = a.sc.s[6];
            double tr = e[0];  // np.sum(strain_increment[:, :gdim], axis=1), gdim = DIMS
            if constexpr (DIMS == 2) tr = e[0] + e[1];
            const double ctr = cD * tr;
#pragma unroll
            for (int i = 0; i < SD; ++i)
                dv[i] = inv_factor * (((cA * s[i] - cB * ev[i]) + cC * e[i]) + ctr * a.sc.s[8 + i]);
            row_times_matrix_fma_n<SD>(e, T->a, y);
        }
#pragma unroll
        for (int i = 0; i < SD; ++i) {
            s[i] = s[i] + (y[i] - c2mu * dv[i]);
            ev[i] = ev[i] + dv[i];
            en[i] = en[i] + e[i];
        }
        transpose_out<SD, FULL, NT>(s, region, lane, a.stress_out + p0 * SD, npts * SD);
        transpose_out<SD, FULL, NT>(ev, region, lane, a.h0_out + p0 * SD, npts * SD);
        transpose_out<SD, FULL, NT>(en, region, lane, a.h1_out + p0 * SD, npts * SD);
    }
}

// ---------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------
template <int LAW, bool IDX, bool FULL, bool NT, bool SPARSE = false>
__device__ __forceinline__ void run_tile(const EvalArgs& a, const StressBases& sb, const Tables* T, double* region,
                                         int* rows_lds, long long p0, int npts, int lane, int r0,
                                         WaveStats& st) {
    // Everything derived from the lane id (chunk -> point/row/column maps, LDS and global
    // offsets) is tile-invariant; left alone, LICM hoists ~100 such values out of the
    // persistent loop and spills them.  Laundering the lane id per tile keeps them as
    // cheap per-tile integer VALU work instead.
    asm volatile("" : "+v"(lane));
    asm volatile("" : "+v"(r0));
    lane &= kWave - 1;  // range known again: per-lane offsets are provably small and non-negative
    if constexpr (LAW == LAW_LE)
        tile_linear_elasticity<IDX, FULL, NT>(a, sb, T, region, rows_lds, p0, npts, lane, r0);
    else if constexpr (LAW == LAW_COMFE_LE)
        tile_comfe_le<IDX, FULL, NT>(a, sb, T, region, rows_lds, p0, npts, lane, r0);
    else if constexpr (LAW == LAW_MAXWELL)
        tile_sls<false, IDX, FULL, NT>(a, sb, T, region, rows_lds, p0, npts, lane, r0);
    else if constexpr (LAW == LAW_KELVIN)
        tile_sls<true, IDX, FULL, NT>(a, sb, T, region, rows_lds, p0, npts, lane, r0);
    else if constexpr (LAW == LAW_VM3D)
        tile_von_mises<IDX, SPARSE, FULL, NT>(a, sb, T, region, rows_lds, p0, npts, lane, st);
    else if constexpr (LAW == LAW_COMFE_DP)
        tile_comfe_dp<false, IDX, FULL, NT>(a, sb, T, region, rows_lds, p0, npts, lane, r0, st);
    else if constexpr (LAW == LAW_COMFE_DP_HYPER)
        tile_comfe_dp<true, IDX, FULL, NT>(a, sb, T, region, rows_lds, p0, npts, lane, r0, st);
    else
        tile_comfe_mises<IDX, FULL, NT>(a, sb, T, region, rows_lds, p0, npts, lane, st);
}

// One full tile of the main kernel.  Indexed kernel: when the 64 parent rows of the tile are
// consecutive (cells of a material are mostly numbered in runs) the coalesced tile body runs on
// shifted base pointers; only tiles with scattered rows pay the per-lane row accesses.
template <int LAW, bool IDX, bool NT, bool SPARSE>
__device__ __forceinline__ void run_full_tile(const EvalArgs& a, const Tables* T, double* region,
                                              int* rows_lds, long long p0, int lane, int r0, WaveStats& st) {
    const StressBases sb{a.stress_in, a.stress_out, a.tangent, a.stress_out2};
    if constexpr (IDX) {
        const int row = a.rows[p0 + lane];
        const int row0 = __builtin_amdgcn_readfirstlane(row);
        if (__all(row == row0 + lane)) {
            const long long shift = (long long)row0 - p0;
            const StressBases sc{a.stress_in + shift * 6, a.stress_out + shift * 6,
                                 a.tangent ? a.tangent + shift * 36 : nullptr,
                                 a.stress_out2 ? a.stress_out2 + shift * 6 : nullptr};
            run_tile<LAW, false, true, NT, SPARSE>(a, sc, T, region, rows_lds, p0, kWave, lane, r0, st);
            return;
        }
    }
    run_tile<LAW, IDX, true, NT, SPARSE>(a, sb, T, region, rows_lds, p0, kWave, lane, r0, st);
}

__device__ __forceinline__ void stage_tables(const EvalArgs& a, Tables* T) {
    const double* src = reinterpret_cast<const double*>(&a.tb);
    double* dst = reinterpret_cast<double*>(T);
    for (int i = threadIdx.x; i < 108; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}

// Per-wave statistics go to one of kCounterSlots copies of the counters (slot = workgroup % slots):
// with tens of thousands of waves, atomics on ONE address serialise at ~10 ns each (measured:
// +0.8 ms at 65k waves); spread over 64 addresses they vanish.  The host sums the slots.
template <int LAW>
__device__ __forceinline__ void flush_stats(const EvalArgs& a, const WaveStats& st, int lane) {
    if constexpr (LAW == LAW_VM3D || LAW == LAW_COMFE_MISES || LAW == LAW_COMFE_DP || LAW == LAW_COMFE_DP_HYPER) {
        const unsigned long long nc = wave_sum(st.nonconv);
        const unsigned long long np = wave_sum(st.plastic);
        const unsigned long long ni = wave_sum(st.iters);
        const unsigned long long nd = wave_sum(st.domain);
        if (lane == 0) {
            unsigned long long* c = a.counters + 4 * (blockIdx.x & (kCounterSlots - 1));
            if (nc) atomicAdd(c + 0, nc);
            if (np) atomicAdd(c + 1, np);
            if (ni) atomicAdd(c + 2, ni);
            if (nd) atomicAdd(c + 3, nd);
        }
    }
}

// Main kernel: all full 64-point tiles.  Persistent: wave w of the grid takes tiles
// w, w + W, w + 2W, ...
template <int LAW, bool NT, bool IDX, bool SPARSE = false>
__global__ void __launch_bounds__(kBlock, (LAW >= LAW_COMFE_DP ? 3 : 4)) evaluate_kernel(const EvalArgs a) {
    __shared__ __attribute__((aligned(16))) Tables T;
    __shared__ __attribute__((aligned(16))) double scratch[kWavesPerBlock][kRegionDoubles];
    __shared__ int rows_all[IDX ? kWavesPerBlock : 1][kWave];
    stage_tables(a, &T);

    const int lane = threadIdx.x & (kWave - 1);
    // wave index as a scalar: tile index, p0 and every array's tile base pointer then live in SGPRs and
    // the per-lane part of an address is a small 32-bit offset (saddr addressing)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    double* region = scratch[wave];
    int* rows_lds = rows_all[IDX ? wave : 0];
    const int r0 = lane % 18;
    const long long nfull = a.n / kWave;
    WaveStats st;
    if (a.tile_map == 0) {
        const long long wstride = (long long)gridDim.x * kWavesPerBlock;
        for (long long tile = (long long)blockIdx.x * kWavesPerBlock + wave; tile < nfull; tile += wstride)
            run_full_tile<LAW, IDX, NT, SPARSE>(a, &T, region, rows_lds, tile * kWave, lane, r0, st);
    } else {
        // XCD-aware variant (experiment): workgroups b and b+8 share an XCD (round-robin dispatch);
        // give every XCD one contiguous eighth of the tiles.  There is no data reuse to keep in an
        // L2, so this only changes DRAM/TLB locality.
        const int xcd = blockIdx.x & 7;
        const long long per = (nfull + 7) / 8;
        const long long lo = xcd * per, hi = (lo + per < nfull) ? lo + per : nfull;
        const long long wstride = (long long)((gridDim.x + 7 - xcd) / 8) * kWavesPerBlock;
        for (long long tile = lo + (long long)(blockIdx.x >> 3) * kWavesPerBlock + wave; tile < hi; tile += wstride)
            run_full_tile<LAW, IDX, NT, SPARSE>(a, &T, region, rows_lds, tile * kWave, lane, r0, st);
    }
    flush_stats<LAW>(a, st, lane);
}

// Low-dimensional constraints: same persistent structure, DIMS = 1 or 2.
template <int LAW, int DIMS, bool NT>
__global__ void __launch_bounds__(kBlock, 4) evaluate_lowdim_kernel(const EvalArgs a) {
    __shared__ __attribute__((aligned(16))) Tables T;
    __shared__ __attribute__((aligned(16))) double scratch[kWavesPerBlock][kRegionDoubles];
    stage_tables(a, &T);
    int lane = threadIdx.x & (kWave - 1);
    // wave index as a scalar: tile index, p0 and every array's tile base pointer then live in SGPRs and
    // the per-lane part of an address is a small 32-bit offset (saddr addressing)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const long long nfull = a.n / kWave;
    const long long wstride = (long long)gridDim.x * kWavesPerBlock;
    for (long long tile = (long long)blockIdx.x * kWavesPerBlock + wave; tile < nfull; tile += wstride) {
        asm volatile("" : "+v"(lane));
        tile_lowdim<LAW, DIMS, true, NT>(a, &T, scratch[wave], tile * kWave, kWave, lane);
    }
}

template <int LAW, int DIMS>
__global__ void __launch_bounds__(kWave) evaluate_lowdim_tail_kernel(const EvalArgs a) {
    __shared__ __attribute__((aligned(16))) Tables T;
    __shared__ __attribute__((aligned(16))) double region[kRegionDoubles];
    stage_tables(a, &T);
    const long long p0 = (a.n / kWave) * kWave;
    tile_lowdim<LAW, DIMS, false, false>(a, &T, region, p0, (int)(a.n - p0), (int)threadIdx.x);
}

// Fused wrapper kernels (VonMises3D under UniaxialStrainFrom3D / PlaneStrainFrom3D).
template <int LAW, int WRAP, bool FULL, bool NT>
__device__ __forceinline__ void run_wrapped_tile(const EvalArgs& a, const Tables* T, double* region, long long p0,
                                                 int npts, int lane, WaveStats& st) {
    if constexpr (LAW == LAW_VM3D)
        tile_von_mises_wrapped<WRAP, FULL, NT>(a, T, region, p0, npts, lane, st);
    else if constexpr (LAW == LAW_LE)
        tile_linear_elasticity_wrapped<WRAP, FULL, NT>(a, T, region, p0, npts, lane);
    else if constexpr (LAW == LAW_COMFE_DP)
        tile_comfe_dp_wrapped<false, WRAP, FULL, NT>(a, T, region, p0, npts, lane, st);
    else if constexpr (LAW == LAW_COMFE_DP_HYPER)
        tile_comfe_dp_wrapped<true, WRAP, FULL, NT>(a, T, region, p0, npts, lane, st);
    else
        tile_comfe_mises_wrapped<WRAP, FULL, NT>(a, T, region, p0, npts, lane, st);
}

template <int LAW, int WRAP, bool NT>
__global__ void __launch_bounds__(kBlock, (LAW >= LAW_COMFE_DP ? 3 : 4)) evaluate_wrapped_kernel(const EvalArgs a) {
    __shared__ __attribute__((aligned(16))) Tables T;
    __shared__ __attribute__((aligned(16))) double scratch[kWavesPerBlock][kRegionDoubles];
    stage_tables(a, &T);
    int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const long long nfull = a.n / kWave;
    const long long wstride = (long long)gridDim.x * kWavesPerBlock;
    WaveStats st;
    for (long long tile = (long long)blockIdx.x * kWavesPerBlock + wave; tile < nfull; tile += wstride) {
        asm volatile("" : "+v"(lane));
        lane &= kWave - 1;
        run_wrapped_tile<LAW, WRAP, true, NT>(a, &T, scratch[wave], tile * kWave, kWave, lane, st);
    }
    flush_stats<LAW>(a, st, lane);
}

template <int LAW, int WRAP>
__global__ void __launch_bounds__(kWave) evaluate_wrapped_tail_kernel(const EvalArgs a) {
    __shared__ __attribute__((aligned(16))) Tables T;
    __shared__ __attribute__((aligned(16))) double region[kRegionDoubles];
    stage_tables(a, &T);
    const long long p0 = (a.n / kWave) * kWave;
    WaveStats st;
    run_wrapped_tile<LAW, WRAP, false, false>(a, &T, region, p0, (int)(a.n - p0), (int)threadIdx.x, st);
    flush_stats<LAW>(a, st, (int)threadIdx.x);
}

// Tail kernel: the last, ragged tile (n % 64 points), one wavefront, guarded 8-byte accesses.
template <int LAW, bool IDX, bool SPARSE = false>
__global__ void __launch_bounds__(kWave) evaluate_tail_kernel(const EvalArgs a) {
    __shared__ __attribute__((aligned(16))) Tables T;
    __shared__ __attribute__((aligned(16))) double region[kRegionDoubles];
    __shared__ int rows_lds[kWave];
    stage_tables(a, &T);
    const int lane = threadIdx.x;
    const long long p0 = (a.n / kWave) * kWave;
    WaveStats st;
    const StressBases sb{a.stress_in, a.stress_out, a.tangent, a.stress_out2};
    run_tile<LAW, IDX, false, false, SPARSE>(a, sb, &T, region, rows_lds, p0, (int)(a.n - p0), lane, lane % 18, st);
    flush_stats<LAW>(a, st, lane);
}

// Commit of a delta trial history (kFlagDeltaHistory): committed[row] += delta[row] for the rows whose bit is set in
// the tile's mask word (the plastic set of the last evaluate).  One wave per 64-point tile, the same row-masked
// 16-byte-chunk access as the evaluate kernel; the ragged last tile with guarded 8-byte accesses.
__global__ void __launch_bounds__(kBlock)
    commit_delta_kernel(double* committed, const double* delta, const unsigned long long* hmask, long long n) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const long long ntiles = (n + kWave - 1) / kWave;
    const long long nfull = n / kWave;
    const long long wstride = (long long)gridDim.x * kWavesPerBlock;
    for (long long tile = (long long)blockIdx.x * kWavesPerBlock + wave; tile < ntiles; tile += wstride) {
        const unsigned long long m = hmask[tile];
        if (m == 0ull) continue;
        const long long base = tile * kWave * 6;
        if (tile < nfull) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int q = k * kWave + lane;
                if ((m >> (q / 3)) & 1ull) {
                    const d2 c = load16<true>(committed + base + 2 * q);
                    const d2 d = load16<true>(delta + base + 2 * q);
                    d2 r;
                    r.x = c.x + d.x;
                    r.y = c.y + d.y;
                    store16<true>(committed + base + 2 * q, r);
                }
            }
        } else {
            const int npts = (int)(n - tile * kWave);
            if (lane < npts && ((m >> lane) & 1ull)) {
#pragma unroll
                for (int i = 0; i < 6; ++i) committed[base + 6 * lane + i] = committed[base + 6 * lane + i] + delta[base + 6 * lane + i];
            }
        }
    }
}

hipError_t launch_commit_delta(double* committed, const double* delta, const unsigned long long* hmask, long long n, int grid,
                               hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(commit_delta_kernel, dim3(grid), dim3(kBlock), 0, stream, committed, delta, hmask, n);
    return hipGetLastError();
}

// strain_from_grad_u (FULL): [9n] -> [6n]
template <bool NT>
__global__ void __launch_bounds__(kBlock)
    strain_kernel(const double* grad, double* strain, long long n, double factor) {
    __shared__ __attribute__((aligned(16))) double scratch[kWavesPerBlock][kRegionDoubles];
    const int lane = threadIdx.x & (kWave - 1);
    // wave index as a scalar: tile index, p0 and every array's tile base pointer then live in SGPRs and
    // the per-lane part of an address is a small 32-bit offset (saddr addressing)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    double* region = scratch[wave];
    const long long ntiles = (n + kWave - 1) / kWave;
    const long long nfull = n / kWave;
    const long long wstride = (long long)gridDim.x * kWavesPerBlock;
    for (long long tile = (long long)blockIdx.x * kWavesPerBlock + wave; tile < ntiles;
         tile += wstride) {
        const long long p0 = tile * kWave;
        Chunks<9> cg;
        double g[9], e[6];
        if (tile < nfull) {
            tile_load<9, true, NT>(cg, grad + p0 * 9, kWave * 9, lane);
            transpose_in<9>(cg, region, lane, g);
            mandel_strain(g, factor, e);
            transpose_out<6, true, NT>(e, region, lane, strain + p0 * 6, kWave * 6);
        } else {
            const int npts = (int)(n - p0);
            tile_load<9, false, NT>(cg, grad + p0 * 9, npts * 9, lane);
            transpose_in<9>(cg, region, lane, g);
            mandel_strain(g, factor, e);
            transpose_out<6, false, NT>(e, region, lane, strain + p0 * 6, npts * 6);
        }
    }
}

// ---------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------
constexpr bool kNT = true;

template <int LAW>
static hipError_t launch_law(const EvalArgs& args, int grid, hipStream_t stream) {
    if constexpr (LAW == LAW_VM3D) {
        if (args.hmask && args.rows) {  // sparse trial history on a submesh (history is local, stress/tangent indexed)
            if (args.n >= kWave)
                hipLaunchKernelGGL((evaluate_kernel<LAW, true, true, true>), dim3(grid), dim3(kBlock), 0, stream, args);
            if (args.n % kWave != 0)
                hipLaunchKernelGGL((evaluate_tail_kernel<LAW, true, true>), dim3(1), dim3(kWave), 0, stream, args);
            return hipGetLastError();
        }
        if (args.hmask) {  // sparse trial history
            if (args.n >= kWave)
                hipLaunchKernelGGL((evaluate_kernel<LAW, true, false, true>), dim3(grid), dim3(kBlock), 0, stream, args);
            if (args.n % kWave != 0)
                hipLaunchKernelGGL((evaluate_tail_kernel<LAW, false, true>), dim3(1), dim3(kWave), 0, stream, args);
            return hipGetLastError();
        }
    }
    if (args.rows) {  // stress / tangent rows addressed through a parent-row index
        if (args.n >= kWave)
            hipLaunchKernelGGL((evaluate_kernel<LAW, true, true>), dim3(grid), dim3(kBlock), 0, stream, args);
        if (args.n % kWave != 0)
            hipLaunchKernelGGL((evaluate_tail_kernel<LAW, true>), dim3(1), dim3(kWave), 0, stream, args);
        return hipGetLastError();
    }
    if (args.n >= kWave) {
        if (args.nontemporal)  // experiments: Options::nontemporal = 0 selects plain (temporal) accesses
            hipLaunchKernelGGL((evaluate_kernel<LAW, true, false>), dim3(grid), dim3(kBlock), 0, stream, args);
        else
            hipLaunchKernelGGL((evaluate_kernel<LAW, false, false>), dim3(grid), dim3(kBlock), 0, stream, args);
    }
    if (args.n % kWave != 0)
        hipLaunchKernelGGL((evaluate_tail_kernel<LAW, false>), dim3(1), dim3(kWave), 0, stream, args);
    return hipGetLastError();
}

template <int LAW, int DIMS>
static hipError_t launch_lowdim(const EvalArgs& args, int grid, hipStream_t stream) {
    if (args.n >= kWave)
        hipLaunchKernelGGL((evaluate_lowdim_kernel<LAW, DIMS, true>), dim3(grid), dim3(kBlock), 0, stream, args);
    if (args.n % kWave != 0)
        hipLaunchKernelGGL((evaluate_lowdim_tail_kernel<LAW, DIMS>), dim3(1), dim3(kWave), 0, stream, args);
    return hipGetLastError();
}

hipError_t launch_evaluate(int law, int dims, const EvalArgs& args, int grid, hipStream_t stream) {
    if (dims == 1 || dims == 2) {
        switch (law * 10 + dims) {
            case LAW_LE * 10 + 1: return launch_lowdim<LAW_LE, 1>(args, grid, stream);
            case LAW_LE * 10 + 2: return launch_lowdim<LAW_LE, 2>(args, grid, stream);
            case LAW_MAXWELL * 10 + 1: return launch_lowdim<LAW_MAXWELL, 1>(args, grid, stream);
            case LAW_MAXWELL * 10 + 2: return launch_lowdim<LAW_MAXWELL, 2>(args, grid, stream);
            case LAW_KELVIN * 10 + 1: return launch_lowdim<LAW_KELVIN, 1>(args, grid, stream);
            case LAW_KELVIN * 10 + 2: return launch_lowdim<LAW_KELVIN, 2>(args, grid, stream);
            default: return hipErrorInvalidValue;
        }
    }
    switch (law) {
        case LAW_LE: return launch_law<LAW_LE>(args, grid, stream);
        case LAW_VM3D: return launch_law<LAW_VM3D>(args, grid, stream);
        case LAW_MAXWELL: return launch_law<LAW_MAXWELL>(args, grid, stream);
        case LAW_KELVIN: return launch_law<LAW_KELVIN>(args, grid, stream);
        case LAW_COMFE_LE: return launch_law<LAW_COMFE_LE>(args, grid, stream);
        case LAW_COMFE_MISES: return launch_law<LAW_COMFE_MISES>(args, grid, stream);
        case LAW_COMFE_DP: return launch_law<LAW_COMFE_DP>(args, grid, stream);
        case LAW_COMFE_DP_HYPER: return launch_law<LAW_COMFE_DP_HYPER>(args, grid, stream);
        default: return hipErrorInvalidValue;
    }
}

template <int LAW, int WRAP>
static hipError_t launch_wrapped(const EvalArgs& args, int grid, hipStream_t stream) {
    if (args.n >= kWave)
        hipLaunchKernelGGL((evaluate_wrapped_kernel<LAW, WRAP, true>), dim3(grid), dim3(kBlock), 0, stream, args);
    if (args.n % kWave != 0)
        hipLaunchKernelGGL((evaluate_wrapped_tail_kernel<LAW, WRAP>), dim3(1), dim3(kWave), 0, stream, args);
    return hipGetLastError();
}

hipError_t launch_evaluate_wrapped(int law, int wrap, const EvalArgs& args, int grid, hipStream_t stream) {
    if (args.n <= 0) return hipSuccess;
    if (wrap != 1 && wrap != 2) return hipErrorInvalidValue;
    switch (law) {
        case LAW_VM3D: return wrap == 1 ? launch_wrapped<LAW_VM3D, 1>(args, grid, stream) : launch_wrapped<LAW_VM3D, 2>(args, grid, stream);
        case LAW_LE: return wrap == 1 ? launch_wrapped<LAW_LE, 1>(args, grid, stream) : launch_wrapped<LAW_LE, 2>(args, grid, stream);
        case LAW_COMFE_MISES:
            return wrap == 1 ? launch_wrapped<LAW_COMFE_MISES, 1>(args, grid, stream)
                             : launch_wrapped<LAW_COMFE_MISES, 2>(args, grid, stream);
        case LAW_COMFE_DP:
            return wrap == 1 ? launch_wrapped<LAW_COMFE_DP, 1>(args, grid, stream) : launch_wrapped<LAW_COMFE_DP, 2>(args, grid, stream);
        case LAW_COMFE_DP_HYPER:
            return wrap == 1 ? launch_wrapped<LAW_COMFE_DP_HYPER, 1>(args, grid, stream)
                             : launch_wrapped<LAW_COMFE_DP_HYPER, 2>(args, grid, stream);
        default: return hipErrorInvalidValue;
    }
}

int default_grid(int law, int num_cu) {
    // Measured on MI355X (VonMises3D, 1e8 points, tools/variance_probe2.py): the more workgroups
    // the better, monotonically -- 512: 9.5-10.3 ms, 1024 (= resident set): 9.3-10.0, 4096:
    // 9.0-9.6, 16384: 9.0-9.4 -- short queues of workgroups rebalance CUs and HBM channels.
    (void)law;
    return 64 * num_cu;
}

// Component maps of the 3D<->1D/2D wrappers: a pure strided copy, one thread per moved double.
__global__ void __launch_bounds__(kBlock)
    strided_copy_kernel(const double* in, double* out, long long n, const CopyMap m) {
    const long long total = n * m.K;
    for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total;
         e += (long long)gridDim.x * kBlock) {
        const long long i = e / m.K;
        const int k = (int)(e - i * m.K);
        out[i * m.out_stride + m.omap[k]] = in[i * m.in_stride + m.imap[k]];
    }
}

hipError_t launch_strided_copy(const double* in, double* out, long long n, const CopyMap& m,
                               hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const long long total = n * m.K;
    long long blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(strided_copy_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, stream, in, out, n, m);
    return hipGetLastError();
}

// Row gather/scatter (submesh <-> parent maps).  One thread per 16-byte chunk when rows are an
// even number of doubles (6, 36, 16, 4: consecutive lanes walk along a row, so each row is moved
// by full-width contiguous accesses), one per double otherwise.
template <int W>  // doubles per thread: 2 or 1
__global__ void __launch_bounds__(kBlock)
    map_rows_kernel(const double* src, const int* src_idx, double* dst, const int* dst_idx,
                    long long n_rows, int row_size) {
    const int per_row = row_size / W;
    const long long total = n_rows * per_row;
    for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total;
         e += (long long)gridDim.x * kBlock) {
        const long long r = e / per_row;
        const int c = (int)(e - r * per_row);
        const long long sr = src_idx ? (long long)src_idx[r] : r;
        const long long dr = dst_idx ? (long long)dst_idx[r] : r;
        if constexpr (W == 2) {
            *reinterpret_cast<d2*>(dst + dr * row_size + 2 * c) =
                *reinterpret_cast<const d2*>(src + sr * row_size + 2 * c);
        } else {
            dst[dr * row_size + c] = src[sr * row_size + c];
        }
    }
}

hipError_t launch_map_rows(const double* src, const int* src_idx, double* dst, const int* dst_idx,
                           long long n_rows, int row_size, hipStream_t stream) {
    if (n_rows <= 0 || row_size <= 0) return hipSuccess;
    const bool wide = (row_size % 2 == 0) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) &&
                      ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0);
    const long long total = n_rows * (wide ? row_size / 2 : row_size);
    long long blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (wide)
        hipLaunchKernelGGL(map_rows_kernel<2>, dim3((unsigned)blocks), dim3(kBlock), 0, stream, src, src_idx, dst, dst_idx, n_rows, row_size);
    else
        hipLaunchKernelGGL(map_rows_kernel<1>, dim3((unsigned)blocks), dim3(kBlock), 0, stream, src, src_idx, dst, dst_idx, n_rows, row_size);
    return hipGetLastError();
}

hipError_t launch_strain(const double* grad, double* strain, long long n, double factor, int grid,
                         hipStream_t stream) {
    hipLaunchKernelGGL((strain_kernel<kNT>), dim3(grid), dim3(kBlock), 0, stream, grad, strain, n,
                       factor);
    return hipGetLastError();
}

}  // namespace fcamd
