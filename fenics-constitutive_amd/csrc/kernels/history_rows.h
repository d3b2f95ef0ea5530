// History write policies of the laws with one [scalar, eps_p(6)] row per point (comfe-rs Mises, Drucker-Prager): 7-double rows
// (the reference layout) and the split layout of device-resident states.
// Part of the device code of libfcamd (translation unit: ../fcamd_kernels.hip, which holds the kernels and launchers).
#pragma once
#include "tile_io.h"
#include "tangent_writers.h"

namespace fcamd {

// Packed plastic-strain history (kFlagPackedHistory; a layout of device-resident states, committed AND trial array).
// The plastic-strain array of a law only accumulates (mises_plasticity_isotropic_hardening.py:161, mises_plasticity.rs:112,
// general.rs:243) and is +0.0 at every point that has never been plastic.  A state therefore keeps, per 64-point tile, the
// rows of the points whose row is not all +0.0 -- the tile's EVER mask, one word per tile next to the array -- packed at the
// head of the tile's slot: the k-th set bit of the mask (ascending point order) owns row k of the slot, rows
// [popcount(ever), 64) are undefined.  One evaluate of a touched tile (a point plastic now, or plastic at the previous
// evaluate: the sparse protocol's `need`) reads the committed run and writes the trial run, ever_trial = ever_committed |
// plastic-now, each as ONE contiguous stream of full lines -- where the unpacked layout moves isolated 48-byte rows (reads
// arrive in 128-byte lines, writes leave in partial 32-byte sectors: profiles/r03_von_mises_mixed_rocprof.md).  Untouched
// tiles keep trial == committed (run and mask word), so the commit stays a pointer swap of arrays and mask arrays.
// Values are bit for bit those of the plain protocol: a row outside ever_committed is the +0.0 row the plain array holds.
//
// load(): requests the committed run (as early as the tile knows that it is touched, so that the request is in flight with the
// law's arithmetic); gather(): the committed row of every lane's point, BEFORE the tile issues its first store; scatter():
// rows_out run <- rows + d at the plastic points, records ever_trial.
//
// Rows inside the run (load_rows()): where the run is long and few of its rows are touched (every point has yielded once, a
// scattered few yield now) moving the whole run costs more than the isolated rows of the plain layout (measured: 1.13 x).  If
// the trial run has the committed run's layout (same EVER word: nothing grew since the last commit) and no new row appears,
// the untouched rows of the trial run already hold the committed values -- the sparse protocol's invariant, row by row -- and
// only the touched rows move, chunk-masked like MaskedRows but through the rank of a row in the run.
// rows inside a packed run (load_rows) when the run has at least kPackedRowsMinRun rows and at most 1 / kPackedRowsDiv of them are touched
constexpr int kPackedRowsDiv = 3, kPackedRowsMinRun = 32;

template <bool FULL, bool NT>
struct PackedRows {
    Chunks<6> c;
    unsigned long long ever_in = 0ull;
    bool partial = false;              // only the touched rows of the run move (load_rows)
    bool row_live[3] = {true, true, true};  // ... chunk k of this lane lies in such a row

    // 16-byte chunks of a run of `rows` rows, rounded up to whole 128-byte lines (a slot is 24 lines; the ragged last tile,
    // whose slot ends with the array: the exact run)
    static __device__ __forceinline__ int run_chunks(int rows) { return FULL ? ((3 * rows + 7) & ~7) : 3 * rows; }  // <= 192

    // requests the committed run (ever_in must be set: the tile's EVER word, a.emask_in[tile])
    __device__ __forceinline__ void load(const double* rows_in, long long p0, int lane) {
        const int nq = run_chunks((int)__popcll(ever_in));
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int q = k * kWave + lane;
            d2 v;
            v.x = 0.0;
            v.y = 0.0;
            if (q < nq) v = load16<NT>(rows_in + p0 * 6 + 2 * q);
            c.v[k] = v;
        }
    }
    // requests only the rows of the run that belong to the points in `touched` (a subset of ever_in); uses the wave's LDS region
    __device__ __forceinline__ void load_rows(const double* rows_in, long long p0, int lane, unsigned long long touched, double* region) {
        int* flag = reinterpret_cast<int*>(region);  // flag[r]: row r of the run belongs to a touched point
        const int rank_in = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ever_in >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ever_in, 0u));
        flag[lane] = 0;  // rows beyond the run (chunk q of lane l belongs to row q / 3 <= 63)
        wave_sync();
        if (((ever_in >> lane) & 1ull) != 0ull) flag[rank_in] = (int)((touched >> lane) & 1ull);
        wave_sync();
        partial = true;
#pragma unroll
        for (int k = 0; k < 3; ++k) {  // whole 64-byte granules, as in the plain layout (rows_granule_live): a granule of 4 chunks meets two rows
            const int r = ((k * kWave + lane) & ~(kRowGranule - 1)) / 3;
            row_live[k] = (flag[r] | flag[r + 1 < kWave ? r + 1 : r]) != 0;
        }
        wave_sync();
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            d2 v;
            v.x = 0.0;
            v.y = 0.0;
            if (row_live[k]) v = load16<NT>(rows_in + p0 * 6 + 2 * (k * kWave + lane));
            c.v[k] = v;
        }
    }
    // ep <- the committed row of this lane's point (+0.0 row for a point outside ever_in), through the wave's LDS region
    __device__ __forceinline__ void gather(double* region, int lane, double (&ep)[6]) {
#pragma unroll
        for (int i = 0; i < 6; ++i) ep[i] = 0.0;
        if (ever_in == 0ull) return;
        const int nq_in = 3 * (int)__popcll(ever_in);
        const int rank_in = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ever_in >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ever_in, 0u));  // set bits below this lane
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int q = k * kWave + lane;
            if (q < nq_in) reinterpret_cast<d2*>(region)[q] = c.v[k];
        }
        wave_sync();
        if (((ever_in >> lane) & 1ull) != 0ull) lds_get_point<6>(region, rank_in, ep);
        wave_sync();
    }
    // trial run <- `row` (the final row of this lane's point: committed row + increment where it is plastic now, the committed
    // row's bits elsewhere, as the plain protocol leaves them); records ever_trial = ever_in | mask
    __device__ __forceinline__ void scatter(ArgsRef a, double* rows_out, long long p0, int lane, unsigned long long mask,
                                            double* region, const double (&row)[6]) {
        const unsigned long long ever_out = ever_in | mask;
        // the run that is written ends on a 128-byte line (the rest of the slot is undefined by contract): no partial line leaves
        const int nq_out = run_chunks((int)__popcll(ever_out));
        const int rank_out = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ever_out >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ever_out, 0u));
        if (((ever_out >> lane) & 1ull) != 0ull) lds_put_point<6>(region, rank_out, row);
        wave_sync();
        if (!partial) {  // the whole run, and its EVER word
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int q = k * kWave + lane;
                if (q < nq_out) store16<NT>(rows_out + p0 * 6 + 2 * q, reinterpret_cast<const d2*>(region)[q]);
            }
            if (lane == 0) a.emask_out[p0 >> 6] = ever_out;
        } else {  // rows inside the run: the trial word is ever_in already
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int q = k * kWave + lane;
                if (row_live[k]) store16<NT>(rows_out + p0 * 6 + 2 * q, reinterpret_cast<const d2*>(region)[q]);
            }
        }
        wave_sync();
    }
};

// In-place plastic-strain rows of a tile whose plastic points are `rows` (the fused wrapper tiles; the plain in-place call of
// tile_von_mises has the same logic inline): few rows -- the three wave-wide chunk instructions of the tile with the chunks of
// untouched rows skipped (a 48-byte row is three 16-byte chunks); many rows -- the whole tile.  request() after the ballot,
// gather() before the tile's first store, store() after the stress store.
template <bool FULL, bool NT>
struct MaskedRows {
    Chunks<6> c;
    bool masked = false;
    bool row_live[3] = {true, true, true};

    __device__ __forceinline__ void request(const double* rows_in, long long p0, int npts, int lane, unsigned long long rows, int masked_max) {
        masked = FULL && (int)__popcll(rows) <= masked_max;
        if (masked) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                row_live[k] = rows_granule_live(rows, k * kWave + lane);
                d2 v;
                v.x = 0.0;
                v.y = 0.0;
                if (row_live[k]) v = load16<NT>(rows_in + p0 * 6 + 2 * (k * kWave + lane));
                c.v[k] = v;
            }
        } else {
            tile_load<6, FULL, NT>(c, rows_in + p0 * 6, npts * 6, lane);
        }
    }
    __device__ __forceinline__ void gather(double* region, int lane, double (&ep)[6]) { transpose_in<6>(c, region, lane, ep); }
    __device__ __forceinline__ void store(double* rows_out, long long p0, int npts, int lane, double* region, const double (&ep)[6]) {
        if (masked) {
            lds_put_point<6>(region, lane, ep);
            wave_sync();
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int q = k * kWave + lane;
                if (row_live[k]) store16<NT>(rows_out + p0 * 6 + 2 * q, reinterpret_cast<const d2*>(region)[q]);
            }
            wave_sync();
        } else {
            transpose_out<6, FULL, NT>(ep, region, lane, rows_out + p0 * 6, npts * 6);
        }
    }
};

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
// The per-tile words of the sparse protocol -- the previous plastic ballot and, for a packed state, the EVER mask of the committed
// run.  A tile loads them FIRST, with its gradient: they arrive with it instead of costing a dependent memory round trip after the
// ballot.  The new ballot is recorded at the END of the tile (sparse_record): a store issued earlier would sit, in the wave's one
// vmcnt, in front of the row loads that follow it.
struct SparseWords {
    unsigned long long m_old = 0ull, ever = 0ull;
    bool same_layout = false;  // packed: the trial run has the committed run's layout (same EVER word: nothing grew since the last commit)
};
__device__ __forceinline__ SparseWords sparse_words(ArgsRef a, long long p0) {
    SparseWords w;
    if (a.hmask != nullptr) {
        w.m_old = a.hmask[p0 >> 6];
        if ((a.flags & kFlagPackedHistory) != 0) {
            w.ever = a.emask_in[p0 >> 6];
            w.same_layout = a.emask_out[p0 >> 6] == w.ever;  // the trial word as the last evaluate of this tile left it
        }
    }
    return w;
}
// the words as scalars (they are wave-uniform): call once the tile's first loads have been waited for
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ void sparse_words_uniform(SparseWords& w) {
    w.m_old = uniform64(w.m_old);
    w.ever = uniform64(w.ever);
    w.same_layout = __builtin_amdgcn_readfirstlane((int)w.same_layout) != 0;
}
// plastic | formerly plastic points of the tile under the sparse protocol
__device__ __forceinline__ unsigned long long sparse_touched(ArgsRef a, const SparseWords& w, unsigned long long mask) {
    return a.hmask == nullptr ? mask : (mask | w.m_old);
}
__device__ __forceinline__ void sparse_record(ArgsRef a, const SparseWords& w, long long p0, unsigned long long mask, int lane) {
    if (a.hmask != nullptr && lane == 0 && mask != w.m_old) a.hmask[p0 >> 6] = mask;
}

// `touched`: sparse_touched() of the tile
template <bool FULL, bool NT>
__device__ __forceinline__ void history7_store(ArgsRef a, long long p0, int npts, int lane,
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
// is three 16-byte chunks of the tile image).  Three phases, placed by the tile so that no load is consumed after a younger
// store (one vmcnt counts both: the wave would wait for the store to complete):
//   request()  after the ballot, before the return mapping: the committed rows this tile needs are requested;
//   gather()   before the tile's first store: the rows of this lane's point, ep;
//   store()    after the stress store: the scalar of a touched tile and the rows out.
template <bool FULL, bool NT>
struct SplitRows {
    PackedRows<FULL, NT> pk;  // its chunk registers pk.c also hold the rows of the unpacked layouts (one array: stays in VGPRs)
    unsigned long long rows = 0ull;
    bool packed = false, masked = false;
    bool row_live[3] = {true, true, true};

    __device__ __forceinline__ void request(ArgsRef a, const SparseWords& w, long long p0, int npts, int lane,
                                            unsigned long long touched, bool hist_in_place, double* region) {
        rows = (a.hmask != nullptr || hist_in_place) ? touched : ~0ull;
        if (rows == 0ull) return;
        packed = (a.flags & kFlagPackedHistory) != 0 && a.hmask != nullptr;  // committed run in, trial run out (PackedRows)
        if (packed) {
            pk.ever_in = w.ever;
            const int run_rows = (int)__popcll(w.ever);
            // few touched rows of a long run, no new row, trial run in the committed layout: the rows alone (PackedRows::load_rows)
            if (FULL && w.same_layout && run_rows >= kPackedRowsMinRun && (rows & ~w.ever) == 0ull &&
                kPackedRowsDiv * (int)__popcll(rows) <= run_rows)
                pk.load_rows(a.h1_in, p0, lane, rows, region);
            else
                pk.load(a.h1_in, p0, lane);
            return;
        }
        masked = FULL && rows != ~0ull && (int)__popcll(rows) <= a.masked_max;
        if (masked) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                row_live[k] = rows_granule_live(rows, k * kWave + lane);
                d2 v;
                v.x = 0.0;
                v.y = 0.0;
                if (row_live[k]) v = load16<NT>(a.h1_in + p0 * 6 + 2 * (k * kWave + lane));
                pk.c.v[k] = v;
            }
        } else {
            tile_load<6, FULL, NT>(pk.c, a.h1_in + p0 * 6, npts * 6, lane);
        }
    }
    // d: in, the increment of this lane's point (zero unless it is plastic now); out, the row to be written -- committed row +
    // increment at the plastic points, the committed row (its bits) elsewhere.  Summed here, at once: the committed rows do not
    // stay in registers across the stress store.
    __device__ __forceinline__ void gather(double* region, int lane, unsigned long long mask, double (&d)[6]) {
        if (rows == 0ull) return;
        double ep[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (packed)
            pk.gather(region, lane, ep);
        else if (mask != 0ull)
            transpose_in<6>(pk.c, region, lane, ep);
        else
            return;  // only stale rows: they leave as the chunks they came as
        const bool plastic = ((mask >> lane) & 1ull) != 0ull;
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = plastic ? ep[i] + d[i] : ep[i];  // the others keep their bits (whole granules move: a neighbour's chunks leave as they came)
    }
    __device__ __forceinline__ void store(ArgsRef a, long long p0, int npts, int lane, unsigned long long mask,
                                          bool hist_in_place, double* region, double scalar, const double (&d)[6]) {
        const bool live = FULL || lane < npts;
        if (rows == 0ull) return;
        if (live) a.h0_out[p0 + lane] = scalar;  // one coalesced 512-byte store per touched tile
        if (packed) {
            pk.scatter(a, a.h1_out, p0, lane, mask, region, d);
            return;
        }
        if (mask != 0ull) {
            if (masked) {
                lds_put_point<6>(region, lane, d);
                wave_sync();
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int q = k * kWave + lane;
                    if (row_live[k]) store16<NT>(a.h1_out + p0 * 6 + 2 * q, reinterpret_cast<const d2*>(region)[q]);
                }
                wave_sync();
            } else {
                transpose_out<6, FULL, NT>(d, region, lane, a.h1_out + p0 * 6, npts * 6);
            }
        } else if (masked) {  // only stale rows: restore the committed values
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (row_live[k]) store16<NT>(a.h1_out + p0 * 6 + 2 * (k * kWave + lane), pk.c.v[k]);
        } else if (!hist_in_place) {
            tile_store<6, FULL, NT>(pk.c, a.h1_out + p0 * 6, npts * 6, lane);
        }
    }
};

}  // namespace fcamd
