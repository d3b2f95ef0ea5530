// Auxiliary kernels of libfcamd, everything that is not the constitutive update itself: the device copy behind
// fcamd_copy_device, strain_from_grad_u, the component maps of the 3D <-> 1D/2D
// wrappers, the row gather / scatter of the submesh maps.  Kept out of fcamd_kernels.hip so that the hash that ties PMC
// traffic figures to the EVALUATE kernels (_build.kernel_hash) does not move with them.
#include <hip/hip_runtime.h>

#include "fcamd_host.h"
#include "kernels/tile_io.h"

namespace fcamd {

namespace {

constexpr int kCopyBlock = 256;
constexpr int kCopyUnroll = 4;  // 16-byte chunks in flight per lane

// dst[i] = src[i] over n16 16-byte chunks: non-temporal loads and stores, each wave-instruction one contiguous
// KiB, kCopyUnroll independent loads issued before the first store, grid-stride over tiles of 4 KiB per wave.
__global__ void __launch_bounds__(kCopyBlock) stream_copy_kernel(const d2* __restrict__ src, d2* __restrict__ dst, size_t n16) {
    const size_t tile = (size_t)kCopyBlock * kCopyUnroll;
    const size_t stride = (size_t)gridDim.x * tile;
    for (size_t base = (size_t)blockIdx.x * tile; base < n16; base += stride) {
        d2 v[kCopyUnroll];
#pragma unroll
        for (int k = 0; k < kCopyUnroll; ++k) {
            const size_t i = base + (size_t)k * kCopyBlock + threadIdx.x;
            if (i < n16) v[k] = __builtin_nontemporal_load(src + i);
        }
#pragma unroll
        for (int k = 0; k < kCopyUnroll; ++k) {
            const size_t i = base + (size_t)k * kCopyBlock + threadIdx.x;
            if (i < n16) __builtin_nontemporal_store(v[k], dst + i);
        }
    }
}

}  // namespace

hipError_t launch_stream_copy(void* dst, const void* src, size_t n16, int grid, hipStream_t stream) {
    if (n16 == 0) return hipSuccess;
    hipLaunchKernelGGL(stream_copy_kernel, dim3(grid), dim3(kCopyBlock), 0, stream, static_cast<const d2*>(src), static_cast<d2*>(dst), n16);
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
    hipLaunchKernelGGL((strain_kernel<true>), dim3(grid), dim3(kBlock), 0, stream, grad, strain, n,
                       factor);
    return hipGetLastError();
}

}  // namespace fcamd
