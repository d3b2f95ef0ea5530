// Plain streaming kernels of libfcamd that are not part of the constitutive update: the device copy behind
// fcamd_copy_device.  Kept out of fcamd_kernels.hip so that the hash that ties PMC traffic figures to the evaluate
// kernels (_build.kernel_hash) does not move with them.
#include <hip/hip_runtime.h>

#include "fcamd_host.h"

namespace fcamd {

namespace {

typedef double d2 __attribute__((ext_vector_type(2)));
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

}  // namespace fcamd
