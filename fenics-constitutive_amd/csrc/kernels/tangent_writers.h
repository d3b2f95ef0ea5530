// Tangent writers (288 of the 456-648 bytes per point): constant tangents streamed from an LDS table, the Mises tangents rebuilt
// per 16-byte chunk from 8 doubles per point; the protocol flags of EvalArgs::flags.
// Part of the device code of libfcamd (translation unit: ../fcamd_kernels.hip, which holds the kernels and launchers).
#pragma once
#include <utility>

#include "tile_io.h"

namespace fcamd {

// ---------------------------------------------------------------------------------------
// tangent writers
// ---------------------------------------------------------------------------------------

// Sparse-tangent protocol: is chunk q of this lane written?  Whole 64-byte granules move (tile_io.h: kRowGranule = 4 chunks), and
// the four chunks of a granule are the chunks of four neighbouring lanes of the SAME pass (q = 64 k + lane): a lane's chunk is
// written if the row of any chunk of its quad is needed -- an OR over the quad (two DPP operations) of each lane's own row bit.
__device__ __forceinline__ bool quad_any(bool mine) {
    int v = mine ? 1 : 0;
    v |= __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);  // quad_perm [1, 0, 3, 2]
    v |= __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);  // quad_perm [2, 3, 0, 1]
    return v != 0;
}

// chunk of point p under the sparse-tangent protocol: its row is needed, or (unless the rows go out bare) a row of its granule is
__device__ __forceinline__ bool tangent_chunk_wanted(unsigned long long tneed, int p, bool exact_rows) {
    const bool mine = ((tneed >> p) & 1ull) != 0ull;
    const bool quad = quad_any(mine);  // (all lanes take part)
    return exact_rows ? mine : quad;
}

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
                                                     unsigned long long tneed, bool exact_rows = false) {
    const int nchunks = npts * 18;
#pragma unroll
    for (int k = 0; k < 18; ++k) {
        const int q = k * kWave + lane;
        const int p = q / 18;
        d2 v = reinterpret_cast<const d2*>(tab)[q - 18 * p];
        bool wanted = ((tneed >> p) & 1ull) != 0ull;
        if constexpr (FULL && !IDX) wanted = tangent_chunk_wanted(tneed, p, exact_rows);  // whole 64-byte granules unless the rows go out bare
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
// (library-internal, set by fill_args when the tangent array is page-locked HOST memory the kernel writes over PCIe:) the sparse tangent
// moves the needed rows alone, not whole 64-byte granules -- over the link every byte counts and there is no partial-unit penalty
// to avoid (host assembler, resident state + sparse tangent, 1e7 points: 503 Mpts/s with bare rows, 488 with granules)
constexpr int kFlagExactTangentRows = 16;
// (library-internal, set by the host entries that rebuild the tangent on the CPU, fcamd_hosttangent.cpp:) EvalArgs::tangent is an array
// of 8 doubles per point followed by one 64-bit word per tile, and the Mises laws store what they publish -- B, C, N[6] -- for the
// PLASTIC points of the tile and the tile's plastic ballot, instead of the 36 entries built from it: 64 bytes per plastic point and 8
// per tile cross PCIe instead of 288 per point; the host expands them with tangent_mises_chunk's own expression and gives every
// elastic point the row that expression yields for an elastic point's parameters (one constant per law).
constexpr int kFlagTangentParams = 32;
// (library-internal, context option "twin_masks":) the launch is the SYNTHETIC TWIN of the packed sparse-protocol VonMises3D kernel: the
// same loads and stores at the same addresses -- the plastic ballot of every tile is read from EvalArgs::cache3d (one word per tile,
// recorded from a real evaluate of the same step) instead of computed -- and no constitutive arithmetic: what the memory system alone
// takes for the step's request stream (bench.py: mem_floor_ms of the row resident_sparse_tangent).  The values it writes mean nothing.
constexpr int kFlagTwin = 64;
template <bool FULL>
__device__ __forceinline__ unsigned long long sparse_tangent_need(ArgsRef a, unsigned long long need) {
    return (FULL && (a.flags & kFlagSparseTangent) != 0 && a.hmask != nullptr) ? need : ~0ull;
}

// The tangent chunks of a tile are computed and stored in groups: the scheduler may interleave the LDS
// reads, the arithmetic and the stores of one group, not across groups (bounds the register pressure).
constexpr int kTangentGroup = 3;

// Which point and which entries chunk q = 64 k + lane of a tile's tangent image holds (18 chunks of 16 bytes per point:
// point p = q / 18, chunk r = q % 18 of its row-major 6 x 6 matrix, i.e. entries [i][j], [i][j + 1] with i = r / 3, j = 2 (r % 3)).
// The evaluate kernels are VALU co-limited wherever they move few bytes per tile (sparse tangent: VALUBusy 88 %; Drucker-Prager:
// 79 %; profiles/r05_row_counters.md), and these integer maps used to be half of the tangent writer's instructions (a division
// by 18 and one by 3 per pass).  With lane = 18 pl + r0 split ONCE per tile, pass k needs a compare and three adds:
//     64 k = 18 (3 k + (10 k) / 18) + (10 k) % 18   ->   r = r0 + (10 k) % 18 (- 18 on carry),  p = pl + 3 k + (10 k) / 18 + carry
// and the table offset of the chunk is 6 i + j = 2 r.
struct ChunkLane {
    int pl, r0;
};
__device__ __forceinline__ ChunkLane chunk_lane(int lane) {
    ChunkLane c;
    c.pl = (int)(__umul24((unsigned)lane, 57u) >> 10);  // lane / 18 for lane < 64
    c.r0 = lane - __mul24(18, c.pl);
    return c;
}
struct ChunkMap {
    int p, r, i, jj;  // point of the tile, chunk of its matrix, matrix row, column pair (j = 2 jj)
};
template <int K>
__device__ __forceinline__ ChunkMap chunk_map(const ChunkLane& c) {
    constexpr int m = (10 * K) % 18, base = 3 * K + (10 * K) / 18;
    const int carry = (c.r0 + m >= 18) ? 1 : 0;
    ChunkMap x;
    x.p = c.pl + base + carry;
    x.r = c.r0 + m - 18 * carry;
    x.i = (int)(__umul24((unsigned)x.r, 11u) >> 5);  // r / 3 for r < 18 (24-bit multiply: full rate; the 32-bit one is a quarter)
    x.jj = x.r - __mul24(3, x.i);
    return x;
}

// Point-dependent tangent of the two Mises laws.  Lane p has published
//   tp[10p + 0] = B, tp[10p + 1] = C, tp[10p + 2 .. 7] = N   (stride 10: conflict-free b128)
// and the tile's tangent is   T[p][i][j] = (ta[i][j] + B * tb[i][j]) + third(i, j)  with
//   VonMises3D:   third = C * (N_i * N_j)      (aah, mises_plasticity_isotropic_hardening.py:170-175)
//   comfe Mises:  third = (C * N_j) * N_i      (column-major .data.0 of ((2 mu theta_bar) n) n^T,
//                                               mises_plasticity.rs:118-123)
// `tneed`: the points of the tile whose tangent rows are written (all ones unless the caller runs the
// sparse-tangent protocol, see sparse_tangent_need()).
template <bool COMFE>
__device__ __forceinline__ d2 tangent_mises_chunk(const double* tp, const double* ta, const double* tb, int p, int r, int i, int jj) {
    const double* t = tp + 10 * p;
    const d2 bc = reinterpret_cast<const d2*>(t)[0];
    const double ni = t[2 + i];
    const d2 nj = *reinterpret_cast<const d2*>(t + 2 + 2 * jj);
    const d2 a = *reinterpret_cast<const d2*>(ta + 2 * r);  // 6 i + j = 2 r
    const d2 b = *reinterpret_cast<const d2*>(tb + 2 * r);
    d2 v;
    if constexpr (COMFE) {
        v.x = (a.x + bc.x * b.x) + (bc.y * nj.x) * ni;
        v.y = (a.y + bc.x * b.y) + (bc.y * nj.y) * ni;
    } else {
        v.x = (a.x + bc.x * b.x) + bc.y * (ni * nj.x);
        v.y = (a.y + bc.x * b.y) + bc.y * (ni * nj.y);
    }
    return v;
}

template <bool COMFE, bool NT, bool MASKED, int K>
__device__ __forceinline__ void tangent_mises_pass(const double* tp, const double* ta, const double* tb, double* tile, int lane,
                                                   const ChunkLane& cl, unsigned long long tneed, bool exact_rows) {
    const ChunkMap m = chunk_map<K>(cl);
    bool wanted = true;
    if constexpr (MASKED) wanted = tangent_chunk_wanted(tneed, m.p, exact_rows);
    // destination: the pass's base (scalar) + this lane's byte offset (32 bits, the same in every pass)
    char* dst = reinterpret_cast<char*>(tile) + K * (kWave * 16) + (unsigned)lane * 16u;
    if (wanted) store_tangent16<NT>(reinterpret_cast<double*>(dst), tangent_mises_chunk<COMFE>(tp, ta, tb, m.p, m.r, m.i, m.jj));
    // bound the register pressure: let the scheduler interleave at most 3 chunks
    if constexpr (K % kTangentGroup == kTangentGroup - 1) __builtin_amdgcn_sched_barrier(0);
}

template <bool COMFE, bool NT, bool MASKED, int... K>
__device__ __forceinline__ void tangent_mises_passes(const double* tp, const double* ta, const double* tb, double* tile, int lane,
                                                     unsigned long long tneed, bool exact_rows, std::integer_sequence<int, K...>) {
    const ChunkLane cl = chunk_lane(lane);
    (tangent_mises_pass<COMFE, NT, MASKED, K>(tp, ta, tb, tile, lane, cl, tneed, exact_rows), ...);
}

// The request stream of the masked passes without their arithmetic (the synthetic twin of the sparse-tangent iteration, law_von_mises.h:
// TWIN): the same chunks leave, each holding the two doubles at the head of its point's published parameters.
template <bool NT, int K>
__device__ __forceinline__ void tangent_twin_pass(const double* tp, double* tile, int lane, const ChunkLane& cl, unsigned long long tneed,
                                                  bool exact_rows) {
    const ChunkMap m = chunk_map<K>(cl);
    const bool wanted = tangent_chunk_wanted(tneed, m.p, exact_rows);
    char* dst = reinterpret_cast<char*>(tile) + K * (kWave * 16) + (unsigned)lane * 16u;
    if (wanted) store_tangent16<NT>(reinterpret_cast<double*>(dst), *reinterpret_cast<const d2*>(tp + 10 * m.p));
}
template <bool NT, int... K>
__device__ __forceinline__ void tangent_twin_passes(const double* tp, double* tile, int lane, unsigned long long tneed, bool exact_rows,
                                                    std::integer_sequence<int, K...>) {
    const ChunkLane cl = chunk_lane(lane);
    (tangent_twin_pass<NT, K>(tp, tile, lane, cl, tneed, exact_rows), ...);
}

template <bool COMFE, bool IDX, bool FULL, bool NT>
__device__ __forceinline__ void tangent_mises(const double* tp, const double* ta, const double* tb,
                                              double* tangent, long long p0, const int* rows_lds,
                                              int npts, int lane, unsigned long long tneed, bool exact_rows = false) {
    if constexpr (FULL && !IDX) {  // the contiguous tile: incremental chunk maps; the need test only under the sparse-tangent protocol
        double* tile = tangent + p0 * 36;
        if (tneed == ~0ull)
            tangent_mises_passes<COMFE, NT, false>(tp, ta, tb, tile, lane, tneed, false, std::make_integer_sequence<int, 18>{});
        else
            tangent_mises_passes<COMFE, NT, true>(tp, ta, tb, tile, lane, tneed, exact_rows, std::make_integer_sequence<int, 18>{});
        return;
    }
    const int nchunks = npts * 18;
#pragma unroll
    for (int k = 0; k < 18; ++k) {
        const int q = k * kWave + lane;
        const int p = q / 18;
        const int r = q - 18 * p;
        const int i = r / 3;
        const d2 v = tangent_mises_chunk<COMFE>(tp, ta, tb, p, r, i, r - 3 * i);
        if ((FULL || q < nchunks) && ((tneed >> p) & 1ull)) store_tangent16<NT>(tangent_chunk<IDX>(tangent, p0, q, rows_lds), v);
        // bound the register pressure: let the scheduler interleave at most 3 chunks
        if (k % kTangentGroup == kTangentGroup - 1) __builtin_amdgcn_sched_barrier(0);
    }
}

template <int PM>
__device__ __forceinline__ bool tangent_params_mode(ArgsRef a) {
    if constexpr (PM == 0) return false;
    if constexpr (PM == 1) return true;
    return (a.flags & kFlagTangentParams) != 0;
}

// kFlagTangentParams: the published parameters (LDS, stride 10) of the tile's plastic points leave as 8 doubles = one aligned 64-byte
// unit per point (four neighbouring lanes), and the plastic ballot as the tile's word behind the parameters of the launch's n points
template <bool FULL, bool NT>
__device__ __forceinline__ void store_tangent_params(ArgsRef a, const double* tp, double* params, long long p0, int npts, int lane,
                                                     unsigned long long plastic) {
    double* tile = params + p0 * 8;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int q = k * kWave + lane;
        const d2 v = *reinterpret_cast<const d2*>(tp + 10 * (q >> 2) + 2 * (q & 3));
        if ((FULL || q < 4 * npts) && ((plastic >> (q >> 2)) & 1ull) != 0ull) store16<NT>(tile + 2 * q, v);
    }
    if (lane == 0) reinterpret_cast<unsigned long long*>(params + 8 * ((a.n + 63) & ~63ll))[p0 >> 6] = plastic;
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
