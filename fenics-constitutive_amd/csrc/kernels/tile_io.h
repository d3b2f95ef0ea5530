// Wave-level building blocks of the evaluate kernels: 16-byte-per-lane global streams over AoS tiles, the AoS <-> lane
// transposition through wave-private LDS, stress rows (contiguous or parent-indexed), Mandel strain, the FMA-chain product.
// Part of the device code of libfcamd (translation unit: ../fcamd_kernels.hip, which holds the kernels and launchers).
#pragma once
#include "../fcamd_internal.h"

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

// The tangent stream (288 of the 456-648 bytes per point) leaves non-temporal like every other stream (+13 % over plain stores
// in round 1; `sc1` / `sc0 sc1` write-through stores measured no better in round 2).
template <bool NT>
__device__ __forceinline__ void store_tangent16(double* p, d2 v) {
    store16<NT>(p, v);
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
//                sub_array[sub]") folded into the kernel's addressing: stress and tangent chunks keep
//                their chunk-major lane assignment and look the row of their point up in a per-wave
//                LDS table of the tile's 64 parent rows.
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

// address of stress chunk q (= 16 bytes) of an indexed tile: chunk q belongs to point q / 3
__device__ __forceinline__ long long stress_chunk_offset(int q, const int* rows_lds) {
    const int p = q / 3;
    return (long long)rows_lds[p] * 6 + 2 * (q - 3 * p);
}

template <bool IDX, bool FULL, bool NT>
struct StressRows {
    Chunks<6> c;

    __device__ __forceinline__ void load(ArgsRef a, const StressBases& sb, long long p0, int npts, int lane,
                                         int* rows_lds) {
        if constexpr (IDX) {
            // chunk-major like the contiguous tile: lane l moves chunks l, l + 64, l + 128 of the tile's VIRTUAL image, three
            // neighbouring lanes one 48-byte row, the rows of a cell (consecutive parent rows, maps.py:159-161) one contiguous
            // piece -- every parent line is requested by one instruction (round 4; before: every lane its own row, three
            // instructions over the same lines: ascending cells of 4 rows 5.06 ms at 5e7 points)
            rows_lds[lane] = (FULL || lane < npts) ? a.rows[p0 + lane] : 0;
            wave_sync();
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int q = k * kWave + lane;
                d2 z;
                z.x = 0.0;
                z.y = 0.0;
                c.v[k] = (FULL || q < 3 * npts) ? load16<NT>(sb.sin + stress_chunk_offset(q, rows_lds)) : z;
            }
        } else {
            tile_load<6, FULL, NT>(c, sb.sin + p0 * 6, npts * 6, lane);
        }
    }
    __device__ __forceinline__ void get(double* region, int lane, double (&s)[6]) { transpose_in<6>(c, region, lane, s); }
    __device__ __forceinline__ void put(const StressBases& sb, double* region, int lane, const double (&s)[6],
                                        long long p0, int npts, const int* rows_lds) {
        if constexpr (IDX) {
            lds_put_point<6>(region, lane, s);
            wave_sync();
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                int q = k * kWave + lane;
                asm volatile("" : "+v"(q));  // recomputed here, not kept alive since load(): registers
                if (FULL || q < 3 * npts) {
                    const d2 v = reinterpret_cast<const d2*>(region)[q];
                    const long long o = stress_chunk_offset(q, rows_lds);
                    store16<NT>(sb.sout + o, v);
                    if (sb.sout2 != nullptr) store16<NT>(sb.sout2 + o, v);
                }
            }
            wave_sync();
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

// Row-masked access moves aligned 64-byte granules (4 chunks of 16 bytes), not bare rows.  A 48-byte plastic-strain row or a
// 288-byte tangent row that is touched alone would leave as partial 64-byte units of the memory side; every chunk of a granule
// that holds a piece of a touched row moves instead, so that whole granules leave.  The neighbours' values are the ones their rows
// hold already (history: trial == committed at untouched points; tangent: the elastic constant the same formula wrote at the
// previous evaluate).  Measured in one process on identical buffers, 1e8 points, 22 % plastic (tools/ab_lib.py, round 5):
// sparse protocol on the reference layout 16 / 32 / 64 / 128-byte units: 9.36 / 9.27 / 9.06 / 9.09 ms; sparse tangent
// 288-byte rows / 64 / 128: 4.67 / 4.51 / 4.60 ms.  The price is bytes: +48 B per isolated plastic-strain row on average,
// +32 B per isolated tangent row.  (The tangent writers test "any chunk of my granule in a needed row" as an OR over the quad of
// lanes that holds the granule: tangent_writers.h, quad_any.)
constexpr int kRowGranule = 4;  // chunks of 16 bytes
// chunk q of a tile image with 3 chunks per row is moved: its granule holds a piece of a row in `rows`
__device__ __forceinline__ bool rows_granule_live(unsigned long long rows, int q) {
    constexpr int G = kRowGranule;
    const int g0 = q & ~(G - 1);
    const int lo = g0 / 3, hi = (g0 + G - 1) / 3;  // <= 63: q <= 191
    return ((rows >> lo) & ((2ull << (hi - lo)) - 1ull)) != 0ull;
}
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

}  // namespace fcamd
