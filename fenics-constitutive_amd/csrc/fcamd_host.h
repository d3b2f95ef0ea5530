// Host-side internals shared by the translation units of the C-ABI layer (fcamd_capi.cpp,
// fcamd_multigpu.cpp, fcamd_memory.cpp): context / model structs, error reporting.
// Not installed; the public boundary is include/fcamd.h.
#pragma once
#include "../../include/fcamd.h"
#include "../../include/fcamd_multi.h"

#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "fcamd_internal.h"

namespace fcamd {

// thread-local message behind fcamd_last_error(); returns `status`
int fail(int status, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            (void)hipGetLastError(); /* reported here: do not leave it for a later launch check */ \
            return ::fcamd::fail(FCAMD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                 __FILE__, __LINE__);                                  \
        }                                                                              \
    } while (0)

struct HistField {
    const char* name;
    int dim;
};

struct LawInfo {
    int n_params;
    int n_hist;
    HistField hist[FCAMD_MAX_HISTORY];
    bool needs_del_t;
};

struct Dims {
    int gd2, sd, gdim;
};

// Launch / data-path knobs of a context.  Defaults come from the FCAMD_* environment variables, read
// ONCE when the context is created (never on the launch path); fcamd_context_set_option changes them
// afterwards (experiments: tools/ab_*.py).
struct Options {
    bool batch_kernel = true;  // FCAMD_BATCH_KERNEL  fcamd_evaluate_batch: the small laws of a call leave as one launch of the batch kernel (0: one launch per law)
    int masked_max = -1;       // FCAMD_MASKED_MAX    row-masked history access up to this many touched rows; -1: per-law default
    long long host_chunk = 0;  // FCAMD_HOST_CHUNK    points per chunk of the staged host entries; 0: automatic
    int host_slots = 4;        // FCAMD_HOST_SLOTS    chunk slots in flight (1..4)
    int zero_copy = 1;         // FCAMD_ZERO_COPY     0 keeps page-locked caller arrays on the staged path
    int zero_copy_grad = 1;    // FCAMD_ZERO_COPY_GRAD 0: fcamd_evaluate_resident uploads the gradient by DMA even if page-locked
    long long bounce_max = 256 << 10;  // FCAMD_BOUNCE_MAX  host calls that move at most this many bytes of pageable caller memory go
                                     // through the context's own page-locked scratch (CPU copies); larger ones page-lock the arrays
    // the tangent of the host entries rebuilt on the CPU (fcamd_hosttangent.cpp)
    int host_tangent_threads = -1;            // FCAMD_HOST_TANGENT_THREADS  threads of the expansion; 0: the kernel writes the tangent over PCIe; -1: automatic
    long long host_tangent_min_points = 1 << 16;  // FCAMD_HOST_TANGENT_MIN  calls with fewer points keep the kernel's own tangent stores
    long long host_tangent_chunk = 0;         // FCAMD_HOST_TANGENT_CHUNK  points per chunk of the parameter ring (Mises laws); 0: automatic
    int host_tangent_streams = 1;             // FCAMD_HOST_TANGENT_STREAMS  streams the chunks of the parameter ring alternate between (1..4)
};

// One expansion job of the host tangent (fcamd_hosttangent.cpp): what to write into which array
struct HostTangentJob {
    enum Kind { CONST = 0, MISES = 1, MISES_COMFE = 2, DRUCKER_PRAGER = 3 } kind;
    int prm;                // doubles per plastic point the kernel sends: 8 (Mises laws: B, C, N[6]) or 12 (Drucker-Prager: 5 coefficients, flag, rho s_tr[6])
    int td;                 // doubles per tangent row: stress_strain_dim squared
    const double* table_a;  // Mises: the tables the kernel's tangent writer reads (Tables::a, ::b) ...
    const double* table_b;
    const double* table_c;  // ... constant tangent: the law's tangent table (Tables::c)
    double* tangent;        // the caller's array (CPU-written only)
    double elastic_row[36]; // Mises: the tangent of an elastic point, formed by the expansion from an elastic point's parameters
};
class ExpandPool;

}  // namespace fcamd

struct fcamd_context {
    int device = 0;
    int num_cu = 256;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    int grid_override = 0;
    bool timing = false;
    fcamd::Options opt;
    // host path: kSlots chunk slots, each with its own stream and device buffers
    static constexpr int kSlots = 4;
    hipStream_t hstream[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    double* dchunk[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    size_t dchunk_points = 0;
    int64_t chunk_points = 0;  // points per chunk of the call in progress
    // page-locked caller ranges: host base -> {bytes, address the GPU sees the base at}
    struct Pinned {
        size_t bytes;
        char* dev;
        bool borrowed = false;  // entered after another context of the process had page-locked it (the lock itself is shared)
        char* lock_base = nullptr;  // base of the process-wide page lock this entry holds a reference to (g_registered)
    };
    std::map<char*, Pinned> registered;
    // Guards `registered` and serialises the host entries with (un)registration: Python's garbage collector
    // may unregister a buffer from another thread while the owning thread is inside a host call (ctypes
    // releases the GIL); the unregistration then waits for that call instead of racing it.
    std::recursive_mutex host_mu;
    int last_host_mode = 0;  // FCAMD_HOST_* flags of the last host-entry call
    unsigned long long twin_masks = 0;  // option "twin_masks": VonMises3D launches of the packed sparse protocol run as their synthetic twin (fcamd_kernels.hip)
    // page-locked scratch of the bounce path (hipHostMalloc: pages of its own, locked for its whole life)
    char* bounce = nullptr;
    char* bounce_dev = nullptr;  // the address the GPU sees it at
    size_t bounce_bytes = 0;
    // direct all-gather (fcamd_multigpu.cpp): one copy stream per peer, created on first use
    std::vector<hipStream_t> peer_streams;
    std::vector<hipEvent_t> peer_events;
    // IPC allocations mapped into this process: handle bytes -> {base address here, open count}.  One
    // allocation (e.g. a caching allocator's segment) may back several exported buffers; it is opened once.
    struct IpcMapping {
        void* base;
        int refs;
    };
    std::map<std::string, IpcMapping> ipc_open;
    // fcamd_evaluate_batch: tables of the batch kernel (page-locked host copy + device copy each; a table is uploaded only when it
    // differs from what its slot holds)
    static constexpr int kBatchMax = 32;   // entries per launch of the batch kernel
    static constexpr int kBatchSlots = 16;  // tables kept: two per state (committed / trial copy) for up to eight states of a thread
    struct BatchSlot {
        fcamd::BatchEntry* host = nullptr;
        fcamd::BatchEntry* dev = nullptr;
        size_t bytes = 0;
        uint64_t hash = 0;
        hipEvent_t uploaded = nullptr;  // recorded behind the table's upload, on ...
        hipStream_t stream = nullptr;   // ... this stream (a launch from another stream waits for the event)
    };
    BatchSlot batch_slots[kBatchSlots];
    unsigned batch_next = 0;
    std::vector<fcamd::BatchEntry> batch_build;
    // host tangent (fcamd_hosttangent.cpp): the pool of expansion threads, the page-locked ring of parameter chunks (8 doubles per
    // point) with one event per slot, and what the last host call spent on the expansion
    fcamd::ExpandPool* pool = nullptr;
    char* tparams = nullptr;
    char* tparams_dev = nullptr;
    size_t tparams_bytes = 0;
    static constexpr int kTangentSlots = 16;  // chunks of the parameter ring the GPU may run ahead of the expansion
    hipEvent_t tp_event[kTangentSlots] = {};
    long long last_host_tangent_cpu_us = 0;  // summed busy time of the expansion threads in the last host call (0: kernel-written tangent)
    int last_host_tangent_threads = 0;
};

struct fcamd_model {
    fcamd_context* ctx = nullptr;
    int law = 0;
    int constraint = FCAMD_FULL;
    fcamd::Dims dims{9, 6, 3};
    fcamd::LawInfo info{};
    double params[8] = {0};
    unsigned long long* d_counters = nullptr;  // [kCounterSlots][4] device
    unsigned long long* h_counters = nullptr;  // the same, pinned host
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;     // the last entry was bracketed by ev0 / ev1 (device entries) ...
    float host_ms = -1.0f;  // ... or, host entries: wall-clock of the synchronous call
    int grid_auto = 0;
    // host constants of the law for `const_del_t` (fill_constants): recomputed only when del_t changes
    bool const_valid = false;
    double const_del_t = 0.0;
    fcamd::Scalars sc;
    fcamd::Tables tb;
};

// ---- shared between fcamd_capi.cpp (launching) and fcamd_hostpath.cpp (host entries) ------------------------
namespace fcamd {

constexpr size_t kCounterBytes = (size_t)kCounterSlots * 4 * sizeof(unsigned long long);

// laws whose history changes only at plastic points (elastic points keep theirs bit for bit)
inline bool has_sparse_history(int law) { return law == FCAMD_VON_MISES_3D || law >= FCAMD_COMFE_MISES_PLASTICITY; }

// laws whose reference history is one [scalar, eps_p(6)] row per point (FCAMD_EVAL_SPLIT_HISTORY)
inline bool has_split_history(int law) { return law >= FCAMD_COMFE_MISES_PLASTICITY; }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// the reference's argument checks (models/interfaces.py:82-101) for one call
int validate_call(const fcamd_model* m, double del_t, int64_t n, const void* grad, const void* stress_prev,
                  const void* stress, const void* const* hist_prev, const void* const* hist, int n_hist, int flags = 0);

// context option "peer_access" (fcamd_multigpu.cpp); rounded size of a buffer peers can map (hipIpcOpenMemHandle, see include/fcamd.h)
int enable_peer_access(fcamd_context* c, int peer_device);
size_t ipc_safe_alloc_size(size_t bytes);

// one evaluate launch of `n` points on `stream` (fcamd_capi.cpp)
int enqueue(fcamd_model* m, double del_t, int64_t n, const double* grad, const double* stress_prev, double* stress,
            double* tangent, const double* const* hprev, double* const* hcur, hipStream_t stream, bool reset_counters,
            const int* rows = nullptr, unsigned long long* hmask = nullptr, int flags = 0, double* stress2 = nullptr,
            unsigned long long* counters = nullptr, const unsigned long long* emask_prev = nullptr,
            unsigned long long* emask = nullptr);

// m->sc / m->tb brought up to date for `del_t` (what every launch does; the host tangent reads the tables before its launch)
void constants_for_call(fcamd_model* m, double del_t);

// download and sum the model's counters (synchronises `stream`); the two halves for callers that have a
// synchronisation of their own coming: enqueue the 2 KB download behind the launches, sum after the wait
int read_stats(fcamd_model* m, hipStream_t stream, fcamd_stats* out);
int enqueue_counters_download(fcamd_model* m, hipStream_t stream);
void sum_counters(const fcamd_model* m, fcamd_stats* out);

// release the chunk buffers and the page-locked scratch of the host entries (fcamd_hostpath.cpp)
void free_host_staging(fcamd_context* c);

// Call-scoped page locks of caller memory, one process-wide registry with reference counts (fcamd_hostpath.cpp):
// the host entries take them for pageable caller arrays; the coordinator of a multi-device call (fcamd_multi.cpp) takes
// them ONCE for the whole arrays, and the per-device host entries then find their slices locked already.
bool temp_lock_acquire(char* q, size_t bytes, char** base, char** dev);
void temp_lock_release(char* base);
// A range that another context of this process keeps page-locked with fcamd_register_host_buffer, entered into
// `c`'s registry with the address c's device sees it at (never unlocked from `c`).
int adopt_registered_range(fcamd_context* c, void* ptr, size_t bytes);
// drop c's own registrations from the process-wide registry of registered ranges (context destruction)
void release_registered_ranges(fcamd_context* c);

// ---- host tangent (fcamd_hosttangent.cpp) ----
constexpr int kFlagTangentParamsHost = 32;  // kernels/tangent_writers.h: kFlagTangentParams (library-internal bit of EvalArgs::flags)
int host_tangent_kind(const fcamd_model* m);          // 0: the law keeps the kernel's tangent stores; else 1 + HostTangentJob::Kind
int host_tangent_threads(const fcamd_context* c);     // resolved thread count (0: off)
int host_tangent_threads_shared(int n_contexts);      // the automatic count of ONE of n contexts that work at the same time (fcamd_multi)
bool host_tangent_applies(const fcamd_model* m, int64_t n, int flags);  // would a host entry of n points rebuild the tangent on the CPU?
ExpandPool* host_tangent_pool(fcamd_context* c);      // the context's pool, created / resized on demand (nullptr: off)
void host_tangent_release(fcamd_context* c);          // pool, ring and events (context destruction, option "trim")
HostTangentJob host_tangent_job(const fcamd_model* m, double* tangent);
int host_tangent_ring(fcamd_context* c, int64_t chunk, int slots, int prm);
void pool_begin(ExpandPool* p, const HostTangentJob& job);
int pool_post(ExpandPool* p, int64_t p0, int64_t np, const double* src, const unsigned long long* mask);  // -> ticket
// doubles of one ring slot for chunks of `chunk` points (a multiple of 64): 8 per point + one ballot word per tile
inline size_t host_tangent_slot_doubles(int64_t chunk, int prm) { return (size_t)chunk * (size_t)prm + (((size_t)chunk / 64 + 1) & ~(size_t)1); }  // (slots stay 16-byte aligned)
void pool_wait(ExpandPool* p, int ticket);
void pool_finish(ExpandPool* p);
double pool_busy_seconds(ExpandPool* p);
int pool_threads(ExpandPool* p);

// fcamd_aux_kernels.hip: dst = src over n16 16-byte chunks (non-temporal accesses, grid-stride)
hipError_t launch_stream_copy(void* dst, const void* src, size_t n16, int grid, hipStream_t stream);

}  // namespace fcamd
