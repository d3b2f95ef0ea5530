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

// the device code, law by law (each header can be read on its own; all of them are part of this translation unit)
#include "kernels/tile_io.h"
#include "kernels/tangent_writers.h"
#include "kernels/wrapped_io.h"
#include "kernels/history_rows.h"
#include "kernels/law_linear_elasticity.h"
#include "kernels/law_sls.h"
#include "kernels/law_von_mises.h"
#include "kernels/law_comfe_mises.h"
#include "kernels/law_drucker_prager.h"
#include "kernels/law_lowdim.h"

namespace fcamd {

// ---------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------
// PM: the tangent leaves as 8 parameters per point (kFlagTangentParams: the host rebuilds the rows, fcamd_hosttangent.cpp) -- 0 never,
// 1 always (the kernels instantiated for it), 2 decided by the flag at run time (the ragged last tile)
template <int LAW, bool IDX, bool FULL, bool NT, int SPARSE = 0, int PM = 0, bool TWIN = false>
__device__ __forceinline__ void run_tile(ArgsRef a, const StressBases& sb, const Tables* T, double* region,
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
        tile_von_mises<IDX, SPARSE, FULL, NT, PM, TWIN>(a, sb, T, region, rows_lds, p0, npts, lane, st);
    else if constexpr (LAW == LAW_COMFE_DP)
        tile_comfe_dp<false, IDX, FULL, NT, PM>(a, sb, T, region, rows_lds, p0, npts, lane, r0, st);
    else if constexpr (LAW == LAW_COMFE_DP_HYPER)
        tile_comfe_dp<true, IDX, FULL, NT, PM>(a, sb, T, region, rows_lds, p0, npts, lane, r0, st);
    else
        tile_comfe_mises<IDX, FULL, NT, PM>(a, sb, T, region, rows_lds, p0, npts, lane, st);
}

// One full tile of the main kernel.  Indexed kernel: when the 64 parent rows of the tile are
// consecutive (cells of a material are mostly numbered in runs) the coalesced tile body runs on
// shifted base pointers; only tiles with scattered rows pay the per-lane row accesses.
template <int LAW, bool IDX, bool NT, int SPARSE, bool PARAMS = false>
__device__ __forceinline__ void run_full_tile(ArgsRef a, const Tables* T, double* region,
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
    run_tile<LAW, IDX, true, NT, SPARSE, PARAMS ? 1 : 0>(a, sb, T, region, rows_lds, p0, kWave, lane, r0, st);
}

__device__ __forceinline__ void stage_tables(ArgsRef a, Tables* T) {
    const double FCAMD_CONSTANT* src = (const double FCAMD_CONSTANT*)&a.tb;
    double* dst = reinterpret_cast<double*>(T);
    for (int i = threadIdx.x; i < 108; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}

// Per-wave statistics go to one of kCounterSlots copies of the counters (slot = workgroup % slots):
// with tens of thousands of waves, atomics on ONE address serialise at ~10 ns each (measured:
// +0.8 ms at 65k waves); spread over 64 addresses they vanish.  The host sums the slots.
template <int LAW>
__device__ __forceinline__ void flush_stats(ArgsRef a, const WaveStats& st, int lane, int block = (int)blockIdx.x) {
    if constexpr (LAW == LAW_VM3D || LAW == LAW_COMFE_MISES || LAW == LAW_COMFE_DP || LAW == LAW_COMFE_DP_HYPER) {
        const unsigned long long nc = wave_sum(st.nonconv);
        const unsigned long long np = wave_sum(st.plastic);
        const unsigned long long ni = wave_sum(st.iters);
        const unsigned long long nd = wave_sum(st.domain);
        if (lane == 0) {
            unsigned long long* c = a.counters + 4 * (block & (kCounterSlots - 1));
            if (nc) atomicAdd(c + 0, nc);
            if (np) atomicAdd(c + 1, np);
            if (ni) atomicAdd(c + 2, ni);
            if (nd) atomicAdd(c + 3, nd);
        }
    }
}

// Main kernel: all full 64-point tiles.  Persistent: wave w of the grid takes tiles
// w, w + W, w + 2W, ...
// SPARSE (VonMises3D): 0 plain, 1 sparse trial history, 2 sparse on the packed plastic-strain layout (tile_von_mises: HIST)
// Waves per SIMD the register budget is cut for: 4 (128 VGPRs), 3 for the Drucker-Prager laws (their return mapping).  (The
// indexed Maxwell kernel was cut for 3 as well while its per-lane stress rows spilled at 128; with the chunk-major rows it fits,
// and 4 waves are worth 16 % to it: 6.24 -> 5.24 ms at 5e7 points on identical buffers.)
// Build knob of the occupancy experiment (tools/build_variant.py three_waves -DFCAMD_EXTRA_LDS=12288): dynamic LDS that the kernels never
// touch, so that only 3 workgroups (12 waves) fit a CU's 160 KiB instead of 4 -- the register budget alone does not lower the occupancy
// of a kernel that needs fewer registers than it allows.
#ifndef FCAMD_EXTRA_LDS
#define FCAMD_EXTRA_LDS 0
#endif
template <int LAW, bool IDX>
constexpr int kMinBlocks = LAW >= LAW_COMFE_DP ? 3 : 4;

// the work of workgroup `block` of `nblocks` (tables staged): shared by the law's own kernel and the batch kernel
template <int LAW, bool NT, bool IDX, int SPARSE, bool PARAMS = false>
__device__ __forceinline__ void evaluate_blocks(ArgsRef a, const Tables* T, double (*scratch)[kRegionDoubles], int (*rows_all)[kWave],
                                                int block, int nblocks, int tid = (int)threadIdx.x) {
    const int lane = tid & (kWave - 1);
    // wave index as a scalar: tile index, p0 and every array's tile base pointer then live in SGPRs and
    // the per-lane part of an address is a small 32-bit offset (saddr addressing)
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    double* region = scratch[wave];
    int* rows_lds = rows_all[IDX ? wave : 0];
    const int r0 = lane % 18;
    const long long nfull = a.n / kWave;
    WaveStats st;
    const long long wstride = (long long)nblocks * kWavesPerBlock;
    for (long long tile = (long long)block * kWavesPerBlock + wave; tile < nfull; tile += wstride)
        run_full_tile<LAW, IDX, NT, SPARSE, PARAMS>(a, T, region, rows_lds, tile * kWave, lane, r0, st);
    flush_stats<LAW>(a, st, lane, block);
}

// the last, ragged tile (n % 64 points): one wavefront, guarded 8-byte accesses
template <int LAW, bool IDX, int SPARSE>
__device__ __forceinline__ void evaluate_tail_tile(ArgsRef a, const Tables* T, double* region, int* rows_lds, int lane) {
    const long long p0 = (a.n / kWave) * kWave;
    WaveStats st;
    const StressBases sb{a.stress_in, a.stress_out, a.tangent, a.stress_out2};
    run_tile<LAW, IDX, false, false, SPARSE, 2>(a, sb, T, region, rows_lds, p0, (int)(a.n - p0), lane, lane % 18, st);
    flush_stats<LAW>(a, st, lane, 0);
}

template <int LAW, bool NT, bool IDX, int SPARSE = 0, bool PARAMS = false>
__global__ void __launch_bounds__(kBlock, (kMinBlocks<LAW, IDX>)) evaluate_kernel(const EvalArgs) {
    ArgsRef a = kernel_args();
    __shared__ __attribute__((aligned(16))) Tables T;
    __shared__ __attribute__((aligned(16))) double scratch[kWavesPerBlock][kRegionDoubles];
    __shared__ int rows_all[IDX ? kWavesPerBlock : 1][kWave];
    stage_tables(a, &T);
    evaluate_blocks<LAW, NT, IDX, SPARSE, PARAMS>(a, &T, scratch, rows_all, (int)blockIdx.x, (int)gridDim.x);
}

// The synthetic twin of evaluate_kernel<LAW_VM3D, true, false, 2> (kernels/tangent_writers.h: kFlagTwin): the full tiles' request stream
// with the ballots read from a.cache3d and no constitutive arithmetic.  A measurement device (context option "twin_masks"), never a result.
__global__ void __launch_bounds__(kBlock, 4) evaluate_twin_kernel(const EvalArgs) {
    ArgsRef a = kernel_args();
    __shared__ __attribute__((aligned(16))) Tables T;
    __shared__ __attribute__((aligned(16))) double scratch[kWavesPerBlock][kRegionDoubles];
    stage_tables(a, &T);
    int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const long long nfull = a.n / kWave;
    const long long wstride = (long long)gridDim.x * kWavesPerBlock;
    WaveStats st;
    const StressBases sb{a.stress_in, a.stress_out, a.tangent, a.stress_out2};
    for (long long tile = (long long)blockIdx.x * kWavesPerBlock + wave; tile < nfull; tile += wstride) {
        asm volatile("" : "+v"(lane));
        lane &= kWave - 1;
        run_tile<LAW_VM3D, false, true, true, 2, 0, true>(a, sb, &T, scratch[wave], nullptr, tile * kWave, kWave, lane, lane % 18, st);
    }
    flush_stats<LAW_VM3D>(a, st, lane);
}

// Low-dimensional constraints: same persistent structure, DIMS = 1 or 2.
template <int LAW, int DIMS, bool NT>
__global__ void __launch_bounds__(kBlock, 4) evaluate_lowdim_kernel(const EvalArgs) {
    ArgsRef a = kernel_args();
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

constexpr int kUxTrips = 4;  // pairs of points per thread of the uniaxial stream kernel
// Uniaxial constraints: a plain element-wise stream over the whole 64-point tiles (kernels/law_lowdim.h: stream_uniaxial);
// the ragged rest goes to evaluate_lowdim_tail_kernel<LAW, 1>.  NOT persistent: a workgroup takes kBlock * kUxTrips
// consecutive pairs of points and ends, so the resident workgroups sweep the arrays as one front (measured on
// 1e8 points, one process, identical buffers: a grid-stride loop over 16k workgroups 0.536 ms, over 64k 0.496 ms).
template <int LAW, bool NT>
__global__ void __launch_bounds__(kBlock, 8) evaluate_uniaxial_kernel(const EvalArgs) {
    ArgsRef a = kernel_args();
    const long long npairs = (a.n / kWave) * (kWave / 2);
    const long long lo = (long long)blockIdx.x * (kBlock * kUxTrips), hi = lo + kBlock * kUxTrips;
    stream_uniaxial<LAW, NT>(a, hi < npairs ? hi : npairs, lo + threadIdx.x, kBlock);
}

template <int LAW, int DIMS>
__global__ void __launch_bounds__(kWave) evaluate_lowdim_tail_kernel(const EvalArgs) {
    ArgsRef a = kernel_args();
    __shared__ __attribute__((aligned(16))) Tables T;
    __shared__ __attribute__((aligned(16))) double region[kRegionDoubles];
    stage_tables(a, &T);
    const long long p0 = (a.n / kWave) * kWave;
    tile_lowdim<LAW, DIMS, false, false>(a, &T, region, p0, (int)(a.n - p0), (int)threadIdx.x);
}

// Fused wrapper kernels (VonMises3D under UniaxialStrainFrom3D / PlaneStrainFrom3D).
template <int LAW, int WRAP, bool FULL, bool NT>
__device__ __forceinline__ void run_wrapped_tile(ArgsRef a, const Tables* T, double* region, long long p0,
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
__global__ void __launch_bounds__(kBlock, (LAW >= LAW_COMFE_DP ? 3 : 4)) evaluate_wrapped_kernel(const EvalArgs) {
    ArgsRef a = kernel_args();
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
__global__ void __launch_bounds__(kWave) evaluate_wrapped_tail_kernel(const EvalArgs) {
    ArgsRef a = kernel_args();
    __shared__ __attribute__((aligned(16))) Tables T;
    __shared__ __attribute__((aligned(16))) double region[kRegionDoubles];
    stage_tables(a, &T);
    const long long p0 = (a.n / kWave) * kWave;
    WaveStats st;
    run_wrapped_tile<LAW, WRAP, false, false>(a, &T, region, p0, (int)(a.n - p0), (int)threadIdx.x, st);
    flush_stats<LAW>(a, st, (int)threadIdx.x);
}

// Tail kernel: the last, ragged tile (n % 64 points), one wavefront, guarded 8-byte accesses.
template <int LAW, bool IDX, int SPARSE = 0>
__global__ void __launch_bounds__(kWave) evaluate_tail_kernel(const EvalArgs) {
    ArgsRef a = kernel_args();
    __shared__ __attribute__((aligned(16))) Tables T;
    __shared__ __attribute__((aligned(16))) double region[kRegionDoubles];
    __shared__ int rows_lds[kWave];
    stage_tables(a, &T);
    evaluate_tail_tile<LAW, IDX, SPARSE>(a, &T, region, rows_lds, (int)threadIdx.x);
}

// ---------------------------------------------------------------------------------------
// The batch kernel (fcamd_evaluate_batch): the laws of ONE form() in one launch.  The reference calls its laws back to back
// (solver/_solver.py:143-144); as separate launches every small law costs the stream three dispatches (counters, main kernel,
// ragged tile: ~30 us together at 1e4 points, measured), which is what a multi-material Newton iteration of a small mesh then
// consists of.  Here a table in device memory holds one entry per law -- its EvalArgs exactly as its own kernel would get them,
// the kernel variant, and the workgroups it owns -- and workgroup b finds its entry, stages that law's tables and runs the very
// same tile code (evaluate_blocks / evaluate_tail_tile); the arguments are read through the constant address space like kernel
// arguments (ArgsRef).  Bit for bit the results of the separate launches.
// ---------------------------------------------------------------------------------------
constexpr int batch_variant(int law, bool idx, int sparse) { return law * 8 + (idx ? 4 : 0) + sparse; }

// `wave`: the wave's index in its workgroup, a scalar.  The thread id is put together again from it and the lane's position in the
// wave (v_mbcnt) inside every variant: threadIdx.x itself would have to stay in a VGPR from the kernel's entry to the start of every
// variant, and the variants cut for exactly 128 VGPRs made the compiler spill it (8 bytes of scratch per lane, one store and one load
// per workgroup -- harmless, but "no kernel uses scratch" is a rule tests/test_kernel_resources.py enforces without exceptions).
template <int LAW, bool IDX, int SPARSE>
__device__ __forceinline__ void batch_run(ArgsRef a, const Tables* T, double (*scratch)[kRegionDoubles], int (*rows_all)[kWave],
                                          int local, int main_blocks, int wave) {
    const int tid = wave * kWave + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (local < main_blocks)
        evaluate_blocks<LAW, true, IDX, SPARSE>(a, T, scratch, rows_all, local, main_blocks, tid);
    else if (wave == 0)
        evaluate_tail_tile<LAW, IDX, SPARSE>(a, T, scratch[0], rows_all[0], tid);
}

// Two kernels, each with the register budget of the laws it runs (round 6): the Drucker-Prager variants (168 VGPRs, 3 waves per SIMD)
// in one, every other law in the other at the 4 waves per SIMD of its own kernel -- as ONE switch the LE / VonMises3D / SLS entries ran
// at Drucker-Prager's occupancy and the kernel carried 608 SGPR spills (profiles/r05_kernel_resources.md).
template <bool DP>
__global__ void __launch_bounds__(kBlock, DP ? 3 : 4) evaluate_batch_kernel(const BatchEntry* table, int count) {
    __shared__ __attribute__((aligned(16))) Tables T;
    __shared__ __attribute__((aligned(16))) double scratch[kWavesPerBlock][kRegionDoubles];
    __shared__ int rows_all[kWavesPerBlock][kWave];
    const BatchEntry FCAMD_CONSTANT* tab = (const BatchEntry FCAMD_CONSTANT*)table;
    int k = 0;  // the entry this workgroup belongs to (uniform: scalar loads and compares)
    for (int j = 1; j < count; ++j)
        if ((int)blockIdx.x >= tab[j].first_block) k = j;
    ArgsRef a = tab[k].args;
    const int local = (int)blockIdx.x - tab[k].first_block, mb = tab[k].main_blocks;
    stage_tables(a, &T);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
#define FCAMD_BATCH_CASE(LAW, IDX, SPARSE)                                           \
    case batch_variant(LAW, IDX, SPARSE):                                            \
        batch_run<LAW, IDX, SPARSE>(a, &T, scratch, rows_all, local, mb, wave);      \
        break;
#define FCAMD_BATCH_LAW(LAW) FCAMD_BATCH_CASE(LAW, false, 0) FCAMD_BATCH_CASE(LAW, true, 0)
    if constexpr (DP) {
        switch (tab[k].variant) {
            FCAMD_BATCH_LAW(LAW_COMFE_DP)
            FCAMD_BATCH_LAW(LAW_COMFE_DP_HYPER)
            default: break;
        }
    } else {
        switch (tab[k].variant) {
            FCAMD_BATCH_LAW(LAW_LE)
            FCAMD_BATCH_LAW(LAW_VM3D)
            FCAMD_BATCH_CASE(LAW_VM3D, false, 1)
            FCAMD_BATCH_CASE(LAW_VM3D, true, 1)
            FCAMD_BATCH_CASE(LAW_VM3D, false, 2)
            FCAMD_BATCH_CASE(LAW_VM3D, true, 2)
            FCAMD_BATCH_LAW(LAW_MAXWELL)
            FCAMD_BATCH_LAW(LAW_KELVIN)
            FCAMD_BATCH_LAW(LAW_COMFE_LE)
            FCAMD_BATCH_LAW(LAW_COMFE_MISES)
            default: break;
        }
    }
#undef FCAMD_BATCH_LAW
#undef FCAMD_BATCH_CASE
}

// the counters of every counting law of the table back to zero: one workgroup per entry (kCounterSlots x 4 = 256 words)
__global__ void __launch_bounds__(kBlock) batch_zero_counters_kernel(const BatchEntry* table) {
    const BatchEntry FCAMD_CONSTANT* tab = (const BatchEntry FCAMD_CONSTANT*)table;
    unsigned long long* c = tab[blockIdx.x].args.counters;
    if (tab[blockIdx.x].counts && c != nullptr) c[threadIdx.x] = 0ull;
}

// ---------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------
// the variants whose full tiles store the tangent's 8 parameters per point (kFlagTangentParams; contiguous rows only)
template <int LAW, int SPARSE>
static hipError_t launch_params(const EvalArgs& args, int grid, hipStream_t stream) {
    if (args.n >= kWave)
        hipLaunchKernelGGL((evaluate_kernel<LAW, true, false, SPARSE, true>), dim3(grid), dim3(kBlock), FCAMD_EXTRA_LDS, stream, args);
    if (args.n % kWave != 0)
        hipLaunchKernelGGL((evaluate_tail_kernel<LAW, false, SPARSE>), dim3(1), dim3(kWave), 0, stream, args);
    return hipGetLastError();
}

template <int LAW>
static hipError_t launch_law(const EvalArgs& args, int grid, hipStream_t stream) {
    if constexpr (LAW == LAW_VM3D || LAW == LAW_COMFE_MISES || LAW == LAW_COMFE_DP || LAW == LAW_COMFE_DP_HYPER) {
        if ((args.flags & kFlagTangentParams) != 0) {
            if (args.rows) return hipErrorInvalidValue;
            if constexpr (LAW == LAW_VM3D) {
                if (args.hmask) return (args.flags & kFlagPackedHistory) ? launch_params<LAW, 2>(args, grid, stream) : launch_params<LAW, 1>(args, grid, stream);
            }
            return launch_params<LAW, 0>(args, grid, stream);
        }
    }
    if constexpr (LAW == LAW_VM3D) {
        if ((args.flags & kFlagTwin) != 0) {  // the synthetic twin: full tiles of the packed sparse protocol only
            if (!args.hmask || args.rows || !(args.flags & kFlagPackedHistory) || !args.cache3d) return hipErrorInvalidValue;
            if (args.n >= kWave) hipLaunchKernelGGL(evaluate_twin_kernel, dim3(grid), dim3(kBlock), FCAMD_EXTRA_LDS, stream, args);
            return hipGetLastError();
        }
        if (args.hmask && args.rows && (args.flags & kFlagPackedHistory)) {  // ... on the packed plastic-strain layout (local to the law)
            if (args.n >= kWave)
                hipLaunchKernelGGL((evaluate_kernel<LAW, true, true, 2>), dim3(grid), dim3(kBlock), FCAMD_EXTRA_LDS, stream, args);
            if (args.n % kWave != 0)
                hipLaunchKernelGGL((evaluate_tail_kernel<LAW, true, 2>), dim3(1), dim3(kWave), 0, stream, args);
            return hipGetLastError();
        }
        if (args.hmask && args.rows) {  // sparse trial history on a submesh (history is local, stress/tangent indexed)
            if (args.n >= kWave)
                hipLaunchKernelGGL((evaluate_kernel<LAW, true, true, 1>), dim3(grid), dim3(kBlock), FCAMD_EXTRA_LDS, stream, args);
            if (args.n % kWave != 0)
                hipLaunchKernelGGL((evaluate_tail_kernel<LAW, true, 1>), dim3(1), dim3(kWave), 0, stream, args);
            return hipGetLastError();
        }
        if (args.hmask && (args.flags & kFlagPackedHistory)) {  // sparse protocol on the packed plastic-strain layout
            if (args.n >= kWave)
                hipLaunchKernelGGL((evaluate_kernel<LAW, true, false, 2>), dim3(grid), dim3(kBlock), FCAMD_EXTRA_LDS, stream, args);
            if (args.n % kWave != 0)
                hipLaunchKernelGGL((evaluate_tail_kernel<LAW, false, 2>), dim3(1), dim3(kWave), 0, stream, args);
            return hipGetLastError();
        }
        if (args.hmask) {  // sparse trial history
            if (args.n >= kWave)
                hipLaunchKernelGGL((evaluate_kernel<LAW, true, false, 1>), dim3(grid), dim3(kBlock), FCAMD_EXTRA_LDS, stream, args);
            if (args.n % kWave != 0)
                hipLaunchKernelGGL((evaluate_tail_kernel<LAW, false, 1>), dim3(1), dim3(kWave), 0, stream, args);
            return hipGetLastError();
        }
    }
    if (args.rows) {  // stress / tangent rows addressed through a parent-row index
        if (args.n >= kWave)
            hipLaunchKernelGGL((evaluate_kernel<LAW, true, true>), dim3(grid), dim3(kBlock), FCAMD_EXTRA_LDS, stream, args);
        if (args.n % kWave != 0)
            hipLaunchKernelGGL((evaluate_tail_kernel<LAW, true>), dim3(1), dim3(kWave), 0, stream, args);
        return hipGetLastError();
    }
    if (args.n >= kWave) {
        hipLaunchKernelGGL((evaluate_kernel<LAW, true, false>), dim3(grid), dim3(kBlock), FCAMD_EXTRA_LDS, stream, args);
    }
    if (args.n % kWave != 0)
        hipLaunchKernelGGL((evaluate_tail_kernel<LAW, false>), dim3(1), dim3(kWave), 0, stream, args);
    return hipGetLastError();
}

template <int LAW, int DIMS>
static hipError_t launch_lowdim(const EvalArgs& args, int grid, hipStream_t stream) {
    if (args.n >= kWave) {
        if constexpr (DIMS == 1) {  // one workgroup per kBlock * kUxTrips pairs: `grid` does not apply
            const long long npairs = (args.n / kWave) * (kWave / 2), per = kBlock * kUxTrips;
            hipLaunchKernelGGL((evaluate_uniaxial_kernel<LAW, true>), dim3((unsigned)((npairs + per - 1) / per)), dim3(kBlock), 0, stream, args);
        } else
            hipLaunchKernelGGL((evaluate_lowdim_kernel<LAW, DIMS, true>), dim3(grid), dim3(kBlock), FCAMD_EXTRA_LDS, stream, args);
    }
    if (args.n % kWave != 0)
        hipLaunchKernelGGL((evaluate_lowdim_tail_kernel<LAW, DIMS>), dim3(1), dim3(kWave), 0, stream, args);
    return hipGetLastError();
}

// the kernel variant launch_law<LAW> picks for these arguments, as the batch kernel's code
int batch_variant_of(int law, const EvalArgs& args) {
    int sparse = 0;
    if (law == LAW_VM3D && args.hmask) sparse = (args.flags & kFlagPackedHistory) ? 2 : 1;
    return batch_variant(law, args.rows != nullptr, sparse);
}

bool batch_law_is_dp(int law) { return law == LAW_COMFE_DP || law == LAW_COMFE_DP_HYPER; }

hipError_t launch_evaluate_batch(const BatchEntry* table, int count, int total_blocks, bool any_counts, bool dp, hipStream_t stream) {
    static_assert(kCounterSlots * 4 == kBlock, "batch_zero_counters_kernel: one thread per counter word");
    if (count <= 0 || total_blocks <= 0) return hipSuccess;
    if (any_counts) hipLaunchKernelGGL(batch_zero_counters_kernel, dim3(count), dim3(kBlock), 0, stream, table);
    if (dp)
        hipLaunchKernelGGL(evaluate_batch_kernel<true>, dim3(total_blocks), dim3(kBlock), 0, stream, table, count);
    else
        hipLaunchKernelGGL(evaluate_batch_kernel<false>, dim3(total_blocks), dim3(kBlock), 0, stream, table, count);
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
        hipLaunchKernelGGL((evaluate_wrapped_kernel<LAW, WRAP, true>), dim3(grid), dim3(kBlock), FCAMD_EXTRA_LDS, stream, args);
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
    // Measured on MI355X, 1e8 points, identical buffers in one process (round 4, tools/ab_knobs.py): the more workgroups the
    // better up to 256-512 per CU -- VonMises3D mixed 16384: 7.87 ms, 65536: 7.755, 131072: 7.76 (LinearElasticity 7.05 ->
    // 6.97) -- short queues of workgroups rebalance CUs and HBM channels; beyond (262144: 8.04, one tile per wave = 390625:
    // 8.27 ms) the dispatch itself shows.  3e7 points: 16384: 2.40, 65536: 2.34 ms; 1e7: flat (0.83 ms).  grid_for() also keeps
    // at least two tiles per wave.
    (void)law;
    return 512 * num_cu;
}

}  // namespace fcamd

