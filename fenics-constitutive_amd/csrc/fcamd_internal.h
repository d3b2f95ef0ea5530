// Internal interface between the C-ABI layer (fcamd_capi.cpp) and the HIP kernels
// (fcamd_kernels.hip).  Not installed; the public boundary is include/fcamd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fcamd {

// Law ids double as template arguments of the kernel (values = fcamd_model_id).
enum Law : int {
    LAW_LE = 1,
    LAW_VM3D = 2,
    LAW_MAXWELL = 3,
    LAW_KELVIN = 4,
    LAW_COMFE_LE = 5,
    LAW_COMFE_MISES = 6,
    LAW_COMFE_DP = 7,        // Drucker-Prager, classic yield surface
    LAW_COMFE_DP_HYPER = 8,  // Drucker-Prager, hyperbolic approximation of the tip
};

// Three 6x6 tables staged into LDS by every workgroup; meaning depends on the law
// (see fill_constants in fcamd_capi.cpp).
struct Tables {
    double a[36];
    double b[36];
    double c[36];
};

// Scalar constants, pre-computed on the host with the reference's own expression
// order so that device results can match NumPy bit for bit.
struct Scalars {
    double s[16];
};

// the device counters exist in kCounterSlots copies of 4 words each (see flush_stats)
constexpr int kCounterSlots = 64;

struct EvalArgs {
    const double* grad;        // [9n]
    const double* stress_in;   // [6n] committed stress (may alias stress_out)
    double* stress_out;        // [6n]
    double* stress_out2;       // nullptr, or a second destination of the stress (the host assembler's page-locked array)
    double* tangent;           // [36n] or nullptr
    const double* h0_in;       // first history field (may alias h0_out) or nullptr
    double* h0_out;
    const double* h1_in;       // second history field or nullptr
    double* h1_out;
    unsigned long long* hmask; // nullptr, or sparse-trial-history mask, one word per 64-point tile (VonMises3D)
    const unsigned long long* emask_in;  // packed plastic-strain history (flags bit 3): EVER mask of the committed array, one word per tile
    unsigned long long* emask_out;       // ... of the trial array (written for touched tiles)
    const int* rows;           // nullptr, or parent row of every point: stress/tangent are parent arrays
    double* cache3d;           // fused 3D->1D/2D wrappers only: the wrapper's cached 3-D stress [6n], in place
    long long n;               // quadrature points
    unsigned long long* counters;  // [kCounterSlots][4]: nonconverged, plastic, newton iterations, reserved
    int masked_max;            // row-masked history access for tiles with at most this many touched rows (else dense)
    int flags;                 // bit 0: sparse-tangent protocol (fcamd_kernels.hip: sparse_tangent_need); bit 2: split history; bit 3: packed plastic-strain history
    Scalars sc;
    Tables tb;
};

// One law of a batch (fcamd_evaluate_batch; table in device memory, read by evaluate_batch_kernel)
struct BatchEntry {
    EvalArgs args;     // exactly what the law's own kernel would get
    int variant;       // batch_variant_of(law, args)
    int first_block;   // the workgroups [first_block, first_block + main_blocks + has_tail) of the launch are this law's
    int main_blocks;   // ... of which the first main_blocks run the full tiles (0: fewer than 64 points)
    int has_tail;      // one more workgroup for the ragged last tile (n % 64 points)
    int counts;        // the law counts (plastic points, Newton iterations, ...): args.counters is zeroed before the launch
    int pad[3];
};
int batch_variant_of(int law, const EvalArgs& args);
// one table holds either Drucker-Prager laws (dp: the kernel cut for 3 waves per SIMD) or the others (4 waves): batch_law_is_dp
bool batch_law_is_dp(int law);
hipError_t launch_evaluate_batch(const BatchEntry* table, int count, int total_blocks, bool any_counts, bool dp, hipStream_t stream);

#ifdef __HIPCC__
// How the device code sees its arguments: a reference into CONSTANT address space (4) -- uniform, invariant loads that the
// compiler keeps in scalar registers and re-loads instead of spilling.  A kernel's by-value EvalArgs parameter IS such memory
// (offset 0 of its kernarg segment: kernel_args()); the batch kernel (fcamd_evaluate_batch: the laws of one form() in one
// launch) reads its EvalArgs from a table in device memory through the same type, so every tile function serves both.
#define FCAMD_CONSTANT __attribute__((address_space(4)))
typedef const EvalArgs FCAMD_CONSTANT& ArgsRef;
typedef const Scalars FCAMD_CONSTANT& ScalarsRef;
__device__ __forceinline__ ArgsRef kernel_args() {
    return *(const EvalArgs FCAMD_CONSTANT*)__builtin_amdgcn_kernarg_segment_ptr();
}
#endif

// Launch the evaluate kernel of `law` on `stream` with `grid` workgroups of 256 threads.
// dims = geometric dimension of the constraint (3: FULL; 2: plane strain/stress; 1: uniaxial)
hipError_t launch_evaluate(int law, int dims, const EvalArgs& args, int grid, hipStream_t stream);
// Fused 3D -> uniaxial-strain (wrap = 1) / plane-strain (wrap = 2) wrapper around VonMises3D or the
// comfe-rs Mises law: grad,
// stress_in/out and tangent are the LOW-dimensional arrays, cache3d the wrapper's 3-D stress, history
// the 3-D law's (in place).
hipError_t launch_evaluate_wrapped(int law, int wrap, const EvalArgs& args, int grid, hipStream_t stream);
// Occupancy-derived default grid (workgroups) for `law` on the current device.
int default_grid(int law, int num_cu);
// strain_from_grad_u, FULL.
hipError_t launch_strain(const double* grad, double* strain, long long n, double factor, int grid,
                         hipStream_t stream);

// out[out_stride*i + omap[k]] = in[in_stride*i + imap[k]], k < K, i < n
struct CopyMap {
    int K, in_stride, out_stride;
    int imap[16], omap[16];
};
hipError_t launch_strided_copy(const double* in, double* out, long long n, const CopyMap& m,
                               hipStream_t stream);

hipError_t launch_map_rows(const double* src, const int* src_idx, double* dst, const int* dst_idx,
                           long long n_rows, int row_size, hipStream_t stream);

}  // namespace fcamd
