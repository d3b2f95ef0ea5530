// Host-memory side of the C ABI (include/fcamd.h): registration of page-locked caller ranges, the host (ndarray)
// entries fcamd_evaluate_host / fcamd_evaluate_resident and the plain copies fcamd_copy_to_device / _to_host.
// Launching, validation and the law constants live in fcamd_capi.cpp (fcamd_host.h declares what is shared).
#include <algorithm>
#include <chrono>
#include <cstring>

#include "fcamd_host.h"

using namespace fcamd;

namespace {

// Address at which the GPU sees the host range [p, p + bytes) when it lies entirely inside ONE
// range page-locked through fcamd_register_host_buffer and keeps the 16-byte alignment the
// kernels' vector accesses need; nullptr otherwise (-> staged path).
double* mapped(const fcamd_context* c, const void* p, size_t bytes) {
    if (!p || c->registered.empty()) return nullptr;
    char* q = static_cast<char*>(const_cast<void*>(p));
    auto it = c->registered.upper_bound(q);
    if (it == c->registered.begin()) return nullptr;
    --it;
    if (q + bytes > it->first + it->second.bytes) return nullptr;
    char* d = it->second.dev + (q - it->first);
    return aligned16(d) ? reinterpret_cast<double*>(d) : nullptr;
}

bool zero_copy_enabled(const fcamd_context* c) { return c->opt.zero_copy != 0; }

// chunk slots of the host-staged entries: sizes the slots, creates the streams, returns the chunk
// length.  Enough chunks in flight to keep both DMA directions busy.  Measured on MI355X / PCIe
// gen5 (bench.py (host_path_figures)): page-locked caller arrays like many small chunks in flight
// (4 x 128 Ki points: 116 Mpts/s); pageable arrays are staged by the runtime and prefer large
// chunks (512 Ki points: 93 Mpts/s).  FCAMD_HOST_CHUNK / FCAMD_HOST_SLOTS override (experiments).
// `staging` = false: every per-chunk array is read / written by the kernel in the caller's page-locked
// memory (zero copy), the chunks only pipeline the small stress download behind the next launch: no
// device buffers, and large chunks (2 Mi points: 472 instead of 404 Mpts/s for the resident pass).
int prepare_chunks(fcamd_context* c, const void* probe_host_ptr, int64_t n, int64_t* chunk_out, bool staging = true,
                   bool locked = false) {
    {
        const bool pinned = locked || mapped(c, probe_host_ptr, 8) != nullptr ||
                            c->registered.count(static_cast<char*>(const_cast<void*>(probe_host_ptr))) != 0;
        c->chunk_points = c->opt.host_chunk > 0 ? std::max<int64_t>(64, (c->opt.host_chunk / 64) * 64)
                                                : (!staging ? (1 << 21) : (pinned ? (1 << 17) : (1 << 19)));
    }
    const int nslots = c->opt.host_slots;
    const int64_t chunk = std::min<int64_t>(c->chunk_points, ((n + 63) / 64) * 64);
    if (staging && chunk > 0 && ((size_t)chunk > c->dchunk_points || !c->dchunk[nslots - 1])) {
        for (int i = 0; i < fcamd_context::kSlots; ++i) {
            if (c->dchunk[i]) HIP_TRY(hipFree(c->dchunk[i]));
            c->dchunk[i] = nullptr;
        }
        c->dchunk_points = 0;
        for (int i = 0; i < nslots; ++i)
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->dchunk[i]), (size_t)chunk * 66 * sizeof(double)));
        c->dchunk_points = (size_t)chunk;
    }
    for (int i = 0; i < nslots; ++i)
        if (!c->hstream[i]) HIP_TRY(hipStreamCreateWithFlags(&c->hstream[i], hipStreamNonBlocking));
    *chunk_out = chunk;
    return FCAMD_OK;
}

// wait for all chunk streams, read the counters, map them to the reference's error conventions.
// `downloaded`: every launch of the call went to hstream[0] and the download of the counters is already queued
// behind them (one wait instead of two -- 10 us of a 60 us call at 1e3 points).  Laws that count nothing (linear
// elasticity, SLS) skip the download altogether.
int finish_chunks(fcamd_model* m, fcamd_stats* stats, bool downloaded = false) {
    fcamd_context* c = m->ctx;
    for (int i = 0; i < fcamd_context::kSlots; ++i)
        if (c->hstream[i]) HIP_TRY(hipStreamSynchronize(c->hstream[i]));
    fcamd_stats local;
    std::memset(&local, 0, sizeof(local));
    if (has_sparse_history(m->law)) {  // = the laws with a plastic branch: the only ones that count
        if (downloaded) {
            sum_counters(m, &local);
        } else {
            int st = read_stats(m, c->hstream[0], &local);
            if (st != FCAMD_OK) return st;
        }
    }
    if (stats) *stats = local;
    if (local.n_domain > 0)
        return fail(FCAMD_ERR_DOMAIN, "non-differentiable tip of Drucker-Prager surface reached");
    if (local.n_nonconverged > 0)
        return fail(FCAMD_ERR_NONCONVERGED,
                    m->law >= FCAMD_COMFE_DRUCKER_PRAGER ? "Plasticity3D: Newton-Raphson did not converge."
                                                         : "Newton-Raphson method did not converge for plastic multiplier.");
    return FCAMD_OK;
}

// single-stream paths: queue the counters' download behind the launches on hstream[0], then finish
int finish_single_stream(fcamd_model* m, fcamd_stats* stats) {
    if (has_sparse_history(m->law)) {
        int st = enqueue_counters_download(m, m->ctx->hstream[0]);
        if (st != FCAMD_OK) return st;
    }
    return finish_chunks(m, stats, /*downloaded=*/true);
}

// Ranges page-locked through fcamd_register_host_buffer by ANY context of the process (base -> bytes).  A host entry of
// another context that meets such a range must use it as it is: this runtime lets a second hipHostRegister of a
// registered address "succeed", and the hipHostUnregister that ends that call then takes the owner's lock away.
//
// g_pages_mu guards BOTH process-wide tables of page locks -- the registered ranges here and the call-scoped locks
// (g_temp, below) -- and is held across "look the range up in both, then hipHostRegister / hipHostUnregister": a thread
// that registers an array while a host entry of another thread holds a call-scoped lock on it would otherwise lock it a
// second time, and the end of that call would take the lock away.  Order: a context's host_mu first, g_pages_mu second.
// A registered range is ONE page lock shared by every context that has entered it (round 4, ADVICE r3: the first context
// used to own the lock and the others kept dangling entries when it let go): reference-counted, released by hipHostUnregister
// when the last context leaves -- unregister, context destroy -- in any order.
std::recursive_mutex g_pages_mu;
struct SharedLock {
    size_t bytes;
    int refs;
};
std::map<char*, SharedLock> g_registered;

struct TempLock {
    size_t bytes;
    int refs;
};
std::map<char*, TempLock> g_temp;  // host base -> page lock held by one or more calls in progress

void note_registered(char* base, size_t bytes) {
    std::lock_guard<std::recursive_mutex> g(g_pages_mu);
    g_registered[base] = {bytes, 1};
}
// one context leaves the shared page lock that starts at `base`; the last one unlocks the pages
void release_registered(char* base) {
    std::lock_guard<std::recursive_mutex> g(g_pages_mu);
    auto it = g_registered.find(base);
    if (it == g_registered.end()) return;
    if (--it->second.refs > 0) return;
    g_registered.erase(it);
    if (hipHostUnregister(base) != hipSuccess) (void)hipGetLastError();  // best effort: the memory may be gone already
}
// base of the registered range that contains [q, q + bytes), taking a reference to it; nullptr if there is none
char* share_registered(char* q, size_t bytes) {
    std::lock_guard<std::recursive_mutex> g(g_pages_mu);
    auto it = g_registered.upper_bound(q);
    if (it == g_registered.begin()) return nullptr;
    auto lo = std::prev(it);
    if (q + bytes > lo->first + lo->second.bytes) return nullptr;
    ++lo->second.refs;
    return lo->first;
}
// does [q, q + bytes) touch a call-scoped page lock of a host entry in progress?
bool overlaps_call_scoped_lock(char* q, size_t bytes) {
    std::lock_guard<std::recursive_mutex> g(g_pages_mu);
    auto it = g_temp.upper_bound(q);
    if (it != g_temp.begin()) {
        auto lo = std::prev(it);
        if (q < lo->first + lo->second.bytes) return true;
    }
    return it != g_temp.end() && it->first < q + bytes;
}
// [q, q + bytes) relative to the registered ranges of the process: 1 inside one, -1 overlaps one partly, 0 disjoint
int in_process_registry(char* q, size_t bytes) {
    std::lock_guard<std::recursive_mutex> g(g_pages_mu);
    auto it = g_registered.upper_bound(q);
    if (it != g_registered.begin()) {
        auto lo = std::prev(it);
        if (q < lo->first + lo->second.bytes) return q + bytes <= lo->first + lo->second.bytes ? 1 : -1;
    }
    if (it != g_registered.end() && it->first < q + bytes) return -1;
    return 0;
}

}  // namespace

namespace fcamd {

// a context is going away: it leaves every page lock it holds a reference to
void release_registered_ranges(fcamd_context* c) {
    for (auto& kv : c->registered)
        if (kv.second.lock_base) release_registered(kv.second.lock_base);
    c->registered.clear();
}

int adopt_registered_range(fcamd_context* c, void* ptr, size_t bytes) {
    if (!c || !ptr || bytes == 0) return fail(FCAMD_ERR_BAD_ARG, "NULL argument");
    std::lock_guard<std::recursive_mutex> lock(c->host_mu);
    HIP_TRY(hipSetDevice(c->device));
    {   // entered already (a repeated call): the old entry's reference goes back first
        auto old = c->registered.find(static_cast<char*>(ptr));
        if (old != c->registered.end()) {
            if (old->second.lock_base) release_registered(old->second.lock_base);
            c->registered.erase(old);
        }
    }
    char* lock_base = share_registered(static_cast<char*>(ptr), bytes);
    if (!lock_base) return fail(FCAMD_ERR_BAD_ARG, "the range is not inside one that a context of this process has registered");
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, ptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        release_registered(lock_base);
        return fail(FCAMD_ERR_HIP, "device %d cannot see the page-locked range", c->device);
    }
    c->registered[static_cast<char*>(ptr)] = {bytes, static_cast<char*>(dev), true, lock_base};
    return FCAMD_OK;
}

// release the staging buffers of the pageable host path (up to 4 slots x 512 Ki points x 66 doubles)
void free_host_staging(fcamd_context* c) {
    for (int i = 0; i < fcamd_context::kSlots; ++i) {
        if (c->dchunk[i]) (void)hipFree(c->dchunk[i]);
        c->dchunk[i] = nullptr;
    }
    c->dchunk_points = 0;
    if (c->bounce) (void)hipHostFree(c->bounce);
    c->bounce = c->bounce_dev = nullptr;
    c->bounce_bytes = 0;
}

}  // namespace fcamd

extern "C" {

int fcamd_register_host_buffer(fcamd_context* c, void* ptr, size_t bytes) {
    if (!c || !ptr || bytes == 0) return fail(FCAMD_ERR_BAD_ARG, "NULL argument");
    std::lock_guard<std::recursive_mutex> lock(c->host_mu);
    std::lock_guard<std::recursive_mutex> pages(g_pages_mu);
    HIP_TRY(hipSetDevice(c->device));
    char* base = static_cast<char*>(ptr);
    if (overlaps_call_scoped_lock(base, bytes))
        return fail(FCAMD_ERR_BAD_ARG, "a host call in progress on another thread holds a page lock on this range: register it when no evaluate runs on it");
    if (c->registered.count(base)) {
        // Same address again: either a repeated call or a NEW buffer that landed where a freed, still-registered one was.
        // This context gives its reference up.  If it was the LAST holder the pages are unlocked here and locked afresh
        // below -- a stale registration would DMA through the old pages.  If other contexts of the process still hold the
        // lock, in_process_registry() below finds the range and this context re-enters the SAME lock: re-pinning is
        // impossible while a range is shared, so a buffer that was freed and allocated again at this address must be
        // unregistered by every context that had registered it before it is registered again (include/fcamd.h).
        if (c->registered[base].lock_base) release_registered(c->registered[base].lock_base);
        c->registered.erase(base);
    }
    {   // page-locked already by another context of this process (one process driving several GPUs, several threads with a
        // context each): this context enters the range with its own device's view of it and shares the page lock
        const int r = in_process_registry(base, bytes);
        if (r > 0) return adopt_registered_range(c, ptr, bytes);
        if (r < 0) return fail(FCAMD_ERR_BAD_ARG, "the range overlaps one that another context of this process has registered");
    }
    HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    note_registered(base, bytes);
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, ptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        dev = nullptr;  // page-locked but not mapped: DMA path only
    }
    c->registered[base] = {dev ? bytes : 0, static_cast<char*>(dev), false, base};
    return FCAMD_OK;
}

int fcamd_host_device_pointer(fcamd_context* c, const void* host_ptr, size_t bytes, void** device_ptr) {
    if (!c || !host_ptr || !device_ptr) return fail(FCAMD_ERR_BAD_ARG, "NULL argument");
    std::lock_guard<std::recursive_mutex> lock(c->host_mu);
    double* d = mapped(c, host_ptr, bytes);
    if (!d) return fail(FCAMD_ERR_BAD_ARG, "host range is not inside a registered, mapped buffer (or not 16-byte aligned)");
    *device_ptr = d;
    return FCAMD_OK;
}

int fcamd_unregister_host_buffer(fcamd_context* c, void* ptr) {
    if (!c || !ptr) return fail(FCAMD_ERR_BAD_ARG, "NULL argument");
    // waits for a host entry in progress on another thread (it holds host_mu for the whole synchronous call)
    std::lock_guard<std::recursive_mutex> lock(c->host_mu);
    std::lock_guard<std::recursive_mutex> pages(g_pages_mu);
    auto it = c->registered.find(static_cast<char*>(ptr));
    if (it == c->registered.end()) return FCAMD_OK;
    HIP_TRY(hipSetDevice(c->device));
    char* lock_base = it->second.lock_base;
    c->registered.erase(it);
    if (lock_base) release_registered(lock_base);  // the last context to leave unlocks the pages
    return FCAMD_OK;
}

}  // extern "C"

namespace {

// A host entry that fails half-way must not return while copies into or out of the CALLER's arrays are
// still in flight on other chunk streams (the caller may free or reuse them as soon as it sees the error).
int drain_and_return(fcamd_context* c, int status) {
    for (int i = 0; i < fcamd_context::kSlots; ++i)
        if (c->hstream[i]) (void)hipStreamSynchronize(c->hstream[i]);
    (void)hipGetLastError();
    return status;
}

// as HIP_TRY, inside the chunk loops of the host entries
#define HIP_TRY_DRAIN(c, expr)                                                                          \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            (void)hipGetLastError();                                                                    \
            return drain_and_return(c, fail(FCAMD_ERR_HIP, "%s failed: %s (%s:%d)", #expr,              \
                                            hipGetErrorString(e_), __FILE__, __LINE__));                \
        }                                                                                               \
    } while (0)

// wall-clock of a synchronous host entry, reported by fcamd_model_last_kernel_ms when timing is on
struct HostTimer {
    fcamd_model* m;
    std::chrono::steady_clock::time_point t0;
    explicit HostTimer(fcamd_model* m_) : m(m_), t0(std::chrono::steady_clock::now()) {
        m->timed = false;
        m->host_ms = -1.0f;
    }
    ~HostTimer() {
        if (m->ctx->timing)
            m->host_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
};

// ---- pageable caller arrays ---------------------------------------------------------------------------------
// The host entries never hand PAGEABLE caller memory to hipMemcpy*.  The HIP runtime moves such memory (above
// 1 MiB) in pieces it page-locks on the fly and remembers in a cache keyed by address and size.  On this stack a
// page lock is an attribute of the process's pages (hsa_amd_memory_lock = KFD SVM "accessible in place"; the GPU
// sees the memory at its host address): memory that appears LATER at a remembered address -- an array that was
// freed and allocated again, a heap that shrank and grew -- carries no such attribute, the cache still calls it
// locked, and the DMA engine faults ("Memory access fault by GPU node-N on address <host address>";
// tools/hsa_lock_probe.c, tools/pageable_copy_probe.py).  Instead:
//   * calls that move at most `bounce_max` bytes (256 KiB: where the CPU copies cost what the page locks of a call cost,
//     round-2 probe bounce_crossover_probe.py (git history)): the CPU copies inputs into / results out of the context's own
//     page-locked scratch (hipHostMalloc) and the kernel runs on the scratch;
//   * larger calls: the caller's arrays are page-locked for the duration of the call (hipHostRegister: the
//     attribute is set on the pages that are there NOW), the kernel runs directly on them, they are unlocked on
//     return.  Measured against the runtime's pageable path, VonMises3D, 1e7 points: 254 instead of 270 ms on
//     arrays never seen before, 80.7 instead of 103.9 ms on arrays used before (round-2 probe temp_register_probe.py (git history));
//   * arrays that cannot be locked (a range that overlaps somebody else's registration): the scratch again, in chunks.
// Page locks taken for the duration of a call are shared by all contexts of the process: two threads (each with a
// context of its own) may pass the SAME read-only array -- the gradient -- at the same time, and two concurrent
// hipHostRegister calls on one address both "succeed" in this runtime, the second hipHostUnregister then aborts
// ("Memobj map does not have ptr").  One registry (g_temp, above), one mutex (g_pages_mu), reference counts.

// address at which the CURRENT device sees the page-locked host address q (every device of the process can reach a
// page-locked range, each at an address of its own: one process may drive several GPUs, fcamd_multi.cpp)
char* device_view(char* q) {
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, q, 0) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return static_cast<char*>(d);
}

}  // namespace

namespace fcamd {

// Takes (or shares) the call-scoped page lock that covers [q, q + bytes).  *base = the key to hand to
// temp_lock_release (nullptr: the range was page-locked by somebody else and is used as it is), *dev = the address the
// current device sees q at.  false: the range cannot be locked.
bool temp_lock_acquire(char* q, size_t bytes, char** base, char** dev) {
    *base = *dev = nullptr;
    std::lock_guard<std::recursive_mutex> g(g_pages_mu);
    {   // page-locked for good by a context of this process (fcamd_register_host_buffer): use it as it is
        const int r = in_process_registry(q, bytes);
        if (r < 0) return false;
        if (r > 0) return (*dev = device_view(q)) != nullptr;
    }
    {   // inside a range another call in progress (or the coordinator of a multi-device call) has locked: share it
        auto it = g_temp.upper_bound(q);
        if (it != g_temp.begin()) {
            --it;
            if (q < it->first + it->second.bytes) {
                if (q + bytes > it->first + it->second.bytes) return false;
                char* d = device_view(q);
                if (!d) return false;
                ++it->second.refs;
                *base = it->first;
                *dev = d;
                return true;
            }
        }
    }
    hipError_t e = hipHostRegister(q, bytes, hipHostRegisterDefault);
    if (e == hipSuccess) {
        char* d = device_view(q);
        if (!d) {
            (void)hipHostUnregister(q);
            return false;
        }
        g_temp[q] = {bytes, 1};
        *base = q;
        *dev = d;
        return true;
    }
    (void)hipGetLastError();
    if (e == hipErrorHostMemoryAlreadyRegistered) {
        // page-locked by somebody else (the application's own hipHostRegister / hipHostMalloc, a range registered
        // with another context): usable, for as long as that somebody keeps it, if the whole range is one mapping
        void *d0 = nullptr, *d1 = nullptr;
        if (hipHostGetDevicePointer(&d0, q, 0) == hipSuccess && hipHostGetDevicePointer(&d1, q + bytes - 1, 0) == hipSuccess &&
            static_cast<char*>(d1) - static_cast<char*>(d0) == static_cast<ptrdiff_t>(bytes - 1)) {
            *dev = static_cast<char*>(d0);
            return true;
        }
        (void)hipGetLastError();
    }
    return false;
}

// Drops one reference; the last one unlocks the pages.  The caller has made sure that nothing of its own is still in
// flight on the range.
void temp_lock_release(char* base) {
    if (!base) return;
    std::lock_guard<std::recursive_mutex> g(g_pages_mu);
    auto it = g_temp.find(base);
    if (it == g_temp.end()) return;
    if (--it->second.refs == 0) {
        (void)hipHostUnregister(base);
        g_temp.erase(it);
    }
    (void)hipGetLastError();
}

}  // namespace fcamd

namespace {

class CallerArrays {
  public:
    explicit CallerArrays(fcamd_context* c) : c_(c) {}
    CallerArrays(const CallerArrays&) = delete;
    CallerArrays& operator=(const CallerArrays&) = delete;
    ~CallerArrays() { release(); }

    // Makes [p, p + bytes) GPU-accessible for the call and returns the address the GPU sees it at in *dev;
    // false: it cannot be page-locked (-> bounce).  A range inside a fcamd_register_host_buffer range is used as is.
    bool lock(const void* p, size_t bytes, char** dev) {
        *dev = nullptr;
        if (!p || bytes == 0) return true;
        char* q = static_cast<char*>(const_cast<void*>(p));
        if (!c_->registered.empty()) {
            auto it = c_->registered.upper_bound(q);
            if (it != c_->registered.begin()) {
                --it;
                if (q < it->first + it->second.bytes) {  // starts inside a registered range ...
                    if (q + bytes > it->first + it->second.bytes || !it->second.dev) return false;  // ... must end there
                    *dev = it->second.dev + (q - it->first);
                    return true;
                }
            }
        }
        char* base = nullptr;
        if (!temp_lock_acquire(q, bytes, &base, dev)) return false;
        if (base) temp_.push_back(base);
        return true;
    }

    bool temp_locked() const { return !temp_.empty(); }

    // give back what this call locked -- only once nothing is in flight on the context's chunk streams
    void release() {
        if (temp_.empty()) return;
        for (int i = 0; i < fcamd_context::kSlots; ++i)
            if (c_->hstream[i]) (void)hipStreamSynchronize(c_->hstream[i]);
        for (char* q : temp_) temp_lock_release(q);
        temp_.clear();
    }

  private:
    fcamd_context* c_;
    std::vector<char*> temp_;  // bases in g_temp this call holds a reference on
};

int ensure_bounce(fcamd_context* c, size_t bytes) {
    if (bytes <= c->bounce_bytes) return FCAMD_OK;
    if (c->bounce) HIP_TRY(hipHostFree(c->bounce));
    c->bounce = c->bounce_dev = nullptr;
    c->bounce_bytes = 0;
    void* h = nullptr;
    HIP_TRY(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipHostFree(h);
        return fail(FCAMD_ERR_HIP, "the page-locked scratch buffer is not mapped into the device's address space");
    }
    c->bounce = static_cast<char*>(h);
    c->bounce_dev = static_cast<char*>(d);
    c->bounce_bytes = bytes;
    return FCAMD_OK;
}

// carves 256-byte aligned arrays out of the scratch: host address and the address the GPU sees
struct BounceLayout {
    size_t used = 0;
    size_t take(size_t bytes) {
        const size_t at = used;
        used += (bytes + 255) & ~(size_t)255;
        return at;
    }
};

constexpr size_t kBounceChunkBytes = (size_t)64 << 20;  // chunk size of the bounce path when the arrays cannot be locked

// points per chunk of a bounce pass over n points of `bytes_per_point` bytes each
int64_t bounce_chunk(const fcamd_context* c, int64_t n, size_t bytes_per_point) {
    const int64_t all = ((n + 63) / 64) * 64;
    if ((size_t)n * bytes_per_point <= (size_t)c->opt.bounce_max) return all;
    const int64_t chunk = (int64_t)(kBounceChunkBytes / bytes_per_point / 64) * 64;
    return std::max<int64_t>(64, std::min<int64_t>(chunk, all));
}

}  // namespace


namespace {

// ---- host tangent (fcamd_hosttangent.cpp) ------------------------------------------------------------------------------------
// Is the tangent of this call rebuilt on the CPU?  Returns the pool (nullptr: the kernel writes the tangent itself).
ExpandPool* host_tangent_for(fcamd_model* m, int64_t n, const double* tangent) {
    fcamd_context* c = m->ctx;
    c->last_host_tangent_cpu_us = 0;
    c->last_host_tangent_threads = 0;
    if (!tangent || !host_tangent_applies(m, n, 0)) return nullptr;
    return host_tangent_pool(c);
}

void host_tangent_done(fcamd_context* c, ExpandPool* pool) {
    pool_finish(pool);  // nothing of this call writes the caller's tangent array after the entry has returned
    c->last_host_tangent_cpu_us = (long long)(pool_busy_seconds(pool) * 1e6);
    c->last_host_tangent_threads = pool_threads(pool);
    c->last_host_mode |= FCAMD_HOST_TANGENT_CPU;
}

// Constant tangent: ONE launch without a tangent array while the pool fills the caller's array from the law's table.
template <class Launch>
int run_const_tangent(fcamd_model* m, ExpandPool* pool, int64_t n, double* tangent, fcamd_stats* stats, Launch&& launch) {
    fcamd_context* c = m->ctx;
    pool_begin(pool, host_tangent_job(m, tangent));  // (the tables of m were brought up to date for this del_t by the caller)
    pool_post(pool, 0, n, nullptr, nullptr);
    int st = launch(0, n, nullptr, c->hstream[0]);
    if (st == FCAMD_OK) st = finish_single_stream(m, stats);
    else st = drain_and_return(c, st);
    host_tangent_done(c, pool);
    return st;
}

// Plasticity laws: the kernel stores the tangent parameters of the plastic points (8 doubles, Drucker-Prager: 12) and every tile's
// ballot into a ring of page-locked chunks (kFlagTangentParams); the pool expands chunk k while the GPU works on the chunks behind it.
// `launch(p0, np, params, stream)` enqueues the kernel of one chunk.  All HIP calls stay on the calling thread: it posts a chunk to
// the pool when the chunk's event has completed and reuses a slot when the pool has expanded what the slot held.
template <class Launch>
int run_param_chunks(fcamd_model* m, ExpandPool* pool, int64_t n, double* tangent, fcamd_stats* stats, Launch&& launch) {
    fcamd_context* c = m->ctx;
    // Chunks.  The GPU is the slower side of the pipeline (VonMises3D at 1e7 points: 33 ms on the link against 16 x 17 ms of expansion),
    // so a call takes the kernels' time plus the expansion of the LAST chunk (2.3 ns per point of it) plus what the chunk boundaries cost:
    // a kernel's last waves drain over the link before the next kernel of the stream may start -- ~150 us per boundary at 1e7 points
    // (38 chunks of 256 Ki points 37.7 ms, 10 of 1 Mi 33.4 ms, same run-ahead), nothing measurable at 1e6 (chunks of 64 Ki / 128 Ki /
    // 256 Ki points: 4.02 / 4.17 / 4.28 ms: there the tail decides).  So: a twelfth of the call, 64 Ki .. 1 Mi points, the last chunk cut
    // in halves down to 64 Ki points; and as many ring slots as 256 MiB of page-locked memory hold (4 .. 16): the GPU runs that far
    // ahead of the expansion when a thread of the pool is held up by another tenant of the host.
    int64_t chunk = c->opt.host_tangent_chunk > 0 ? c->opt.host_tangent_chunk : std::max<int64_t>(1 << 16, std::min<int64_t>(1 << 20, (n / 12 + 63) / 64 * 64));
    chunk = std::max<int64_t>(64, chunk / 64 * 64);
    chunk = std::min<int64_t>(chunk, (n + 63) / 64 * 64);
    std::vector<int64_t> start;  // chunk k = points [start[k], start[k + 1])
    {
        const int64_t taper_min = c->opt.host_tangent_chunk > 0 ? chunk : (1 << 16);
        int64_t p = 0;
        while (p < n) {
            start.push_back(p);
            const int64_t left = n - p;
            int64_t take = chunk;
            if (left <= chunk) take = left > 2 * taper_min ? (left / 2 + 63) / 64 * 64 : left;  // the tail: halves down to taper_min
            p += std::min(take, left);
        }
        start.push_back(n);
    }
    const HostTangentJob job = host_tangent_job(m, tangent);
    const int prm = job.prm;
    const int nslots = (int)std::max<int64_t>(4, std::min<int64_t>(fcamd_context::kTangentSlots, ((int64_t)256 << 20) / (chunk * prm * 8)));
    int st = host_tangent_ring(c, chunk, nslots, prm);
    if (st != FCAMD_OK) return st;
    pool_begin(pool, job);
    // option "host_tangent_streams" > 1: the chunks alternate between streams (they are independent, the counters atomic).  Measured: it
    // helps chunks of 256 Ki points and less and loses at the automatic sizes -- two kernels share the link, each chunk completes later
    // and the call ends with two expansions instead of one -- so the default is ONE stream.
    const int nstreams = std::max(1, std::min(c->opt.host_tangent_streams, fcamd_context::kSlots));  // (hstream[] has kSlots entries)
    for (int i = 1; i < nstreams; ++i)
        if (!c->hstream[i]) HIP_TRY(hipStreamCreateWithFlags(&c->hstream[i], hipStreamNonBlocking));
    if (nstreams > 1) HIP_TRY(hipStreamSynchronize(c->hstream[0]));  // the counters' reset (queued by the caller) before any chunk counts
    const int64_t nchunks = (int64_t)start.size() - 1;
    std::vector<int> ticket((size_t)nchunks, -1);
    int64_t posted = 0;  // chunks [0, posted) have completed on the GPU and are with the pool
    const size_t slot_doubles = host_tangent_slot_doubles(chunk, prm);
    auto slot_host = [&](int64_t k) { return reinterpret_cast<const double*>(c->tparams) + (size_t)(k % nslots) * slot_doubles; };
    auto slot_dev = [&](int64_t k) { return reinterpret_cast<double*>(c->tparams_dev) + (size_t)(k % nslots) * slot_doubles; };
    auto points = [&](int64_t k) { return start[(size_t)k + 1] - start[(size_t)k]; };
    auto post = [&](int64_t k) {
        const int64_t np = points(k);
        // the ballots of a launch of np points lie behind its prm * roundup(np, 64) parameter doubles (tangent_writers.h: store_tangent_params)
        const unsigned long long* words = reinterpret_cast<const unsigned long long*>(slot_host(k) + (size_t)prm * (size_t)((np + 63) / 64 * 64));
        ticket[(size_t)k] = pool_post(pool, start[(size_t)k], np, slot_host(k), words);
    };
    hipError_t err = hipSuccess;
    bool counters_queued = false;
    for (int64_t k = 0; k < nchunks && st == FCAMD_OK && err == hipSuccess; ++k) {
        // whatever has completed goes to the pool first
        while (posted < k && hipEventQuery(c->tp_event[posted % nslots]) == hipSuccess) post(posted++);
        (void)hipGetLastError();  // (hipErrorNotReady of the query)
        if (k >= nslots) {  // the slot's previous chunk: completed, posted, expanded
            while (posted <= k - nslots && err == hipSuccess) {
                err = hipEventSynchronize(c->tp_event[posted % nslots]);
                if (err == hipSuccess) post(posted++);
            }
            if (err != hipSuccess) break;
            pool_wait(pool, ticket[(size_t)(k - nslots)]);
        }
        hipStream_t s = c->hstream[k % nstreams];
        st = launch(start[(size_t)k], points(k), slot_dev(k), s);
        if (st == FCAMD_OK) err = hipEventRecord(c->tp_event[k % nslots], s);
        // one stream: the counters ride behind the last chunk (no extra round trip after the last event)
        if (st == FCAMD_OK && err == hipSuccess && k + 1 == nchunks && nstreams == 1 && has_sparse_history(m->law)) {
            st = enqueue_counters_download(m, s);
            counters_queued = true;
        }
    }
    if (st == FCAMD_OK && err == hipSuccess) {
        while (posted < nchunks && err == hipSuccess) {
            err = hipEventSynchronize(c->tp_event[posted % nslots]);
            if (err == hipSuccess) post(posted++);
        }
        if (err == hipSuccess && !counters_queued && has_sparse_history(m->law)) st = enqueue_counters_download(m, c->hstream[0]);  // every chunk has completed
    }
    if (err != hipSuccess) {
        (void)hipGetLastError();
        st = fail(FCAMD_ERR_HIP, "host tangent pipeline: %s", hipGetErrorString(err));
    }
    if (st != FCAMD_OK) st = drain_and_return(c, st);
    else st = finish_chunks(m, stats, /*downloaded=*/true);
    host_tangent_done(c, pool);
    return st;
}

}  // namespace

extern "C" {

// Host (ndarray) entry.  The kernel runs directly on the caller's arrays (zero copy): on ranges registered with
// fcamd_register_host_buffer as they are, on pageable arrays after page-locking them for the duration of the call;
// small calls and arrays that cannot be locked go through the context's page-locked scratch (CallerArrays above).
// With the "zero_copy" option off (or an array off the 16-byte grid): chunked H2D -> kernel -> D2H over up to four
// chunk slots on four streams -- DMA from / into the page-locked arrays.
int fcamd_evaluate_host(fcamd_model* m, double t, double del_t, int64_t n, const double* grad,
                        double* stress, double* tangent, double* const* hist, int n_hist,
                        fcamd_stats* stats) {
    (void)t;
    int st = validate_call(m, del_t, n, grad, stress, stress,
                           reinterpret_cast<const void* const*>(hist),
                           reinterpret_cast<const void* const*>(hist), n_hist);
    if (st != FCAMD_OK) return st;
    fcamd_context* c = m->ctx;
    std::lock_guard<std::recursive_mutex> lock(c->host_mu);
    HostTimer timer(m);
    HIP_TRY(hipSetDevice(c->device));
    if (stats) std::memset(stats, 0, sizeof(*stats));
    const size_t GD2 = (size_t)m->dims.gd2, SD = (size_t)m->dims.sd, TD = SD * SD;
    const int NH = m->info.n_hist;
    c->last_host_mode = 0;
    if (!c->hstream[0]) HIP_TRY(hipStreamCreateWithFlags(&c->hstream[0], hipStreamNonBlocking));
    if (n == 0) {
        if (has_sparse_history(m->law)) HIP_TRY(hipMemsetAsync(m->d_counters, 0, kCounterBytes, c->hstream[0]));
        return finish_single_stream(m, stats);
    }
    const size_t N = (size_t)n;
    size_t hist_doubles = 0;
    for (int k = 0; k < NH; ++k) hist_doubles += (size_t)m->info.hist[k].dim;
    const size_t bytes_per_point = (GD2 + SD + (tangent ? TD : 0) + hist_doubles) * sizeof(double);

    // the tangent rebuilt on the CPU (fcamd_hosttangent.cpp): the caller's tangent array is then neither locked nor mapped
    ExpandPool* pool = host_tangent_for(m, n, tangent);

    // every array inside the caller's registered ranges: nothing to lock, whatever the size
    bool all_registered = mapped(c, grad, N * GD2 * sizeof(double)) && mapped(c, stress, N * SD * sizeof(double)) &&
                          (pool || !tangent || mapped(c, tangent, N * TD * sizeof(double)));
    for (int k = 0; k < NH && all_registered; ++k)
        all_registered = mapped(c, hist[k], N * (size_t)m->info.hist[k].dim * sizeof(double)) != nullptr;

    CallerArrays arrays(c);
    char *z_grad = nullptr, *z_stress = nullptr, *z_tan = nullptr, *z_hist[FCAMD_MAX_HISTORY] = {nullptr, nullptr};
    bool locked = false;
    if (all_registered || N * bytes_per_point > (size_t)c->opt.bounce_max) {
        locked = arrays.lock(grad, N * GD2 * sizeof(double), &z_grad) && arrays.lock(stress, N * SD * sizeof(double), &z_stress);
        for (int k = 0; k < NH && locked; ++k)
            locked = arrays.lock(hist[k], N * (size_t)m->info.hist[k].dim * sizeof(double), &z_hist[k]);
        if (locked && pool) {  // the zero-copy launch needs the other arrays on the 16-byte grid; else: today's paths, tangent and all
            bool ok = aligned16(z_grad) && aligned16(z_stress);
            for (int k = 0; k < NH; ++k) ok = ok && aligned16(z_hist[k]);
            if (!ok) pool = nullptr;
        }
        if (locked && !pool) locked = arrays.lock(tangent, tangent ? N * TD * sizeof(double) : 0, &z_tan);
        if (!locked) arrays.release();
    }
    if (!locked) pool = nullptr;

    if (!locked) {
        // bounce: CPU copies through the context's page-locked scratch, one launch per chunk
        c->last_host_mode = FCAMD_HOST_BOUNCE;
        const int64_t chunk = bounce_chunk(c, n, bytes_per_point);
        BounceLayout lay;
        const size_t o_grad = lay.take((size_t)chunk * GD2 * sizeof(double)), o_stress = lay.take((size_t)chunk * SD * sizeof(double));
        const size_t o_tan = tangent ? lay.take((size_t)chunk * TD * sizeof(double)) : 0;
        size_t o_hist[FCAMD_MAX_HISTORY] = {0, 0};
        for (int k = 0; k < NH; ++k) o_hist[k] = lay.take((size_t)chunk * (size_t)m->info.hist[k].dim * sizeof(double));
        st = ensure_bounce(c, lay.used);
        if (st != FCAMD_OK) return st;
        hipStream_t s = c->hstream[0];
        if (has_sparse_history(m->law)) HIP_TRY(hipMemsetAsync(m->d_counters, 0, kCounterBytes, s));
        for (int64_t p0 = 0; p0 < n; p0 += chunk) {
            const size_t np = (size_t)std::min<int64_t>(chunk, n - p0);
            std::memcpy(c->bounce + o_grad, grad + GD2 * p0, np * GD2 * sizeof(double));
            std::memcpy(c->bounce + o_stress, stress + SD * p0, np * SD * sizeof(double));
            double* d_hist[FCAMD_MAX_HISTORY] = {nullptr, nullptr};
            for (int k = 0; k < NH; ++k) {
                const size_t d = (size_t)m->info.hist[k].dim;
                std::memcpy(c->bounce + o_hist[k], hist[k] + d * p0, np * d * sizeof(double));
                d_hist[k] = reinterpret_cast<double*>(c->bounce_dev + o_hist[k]);
            }
            double* d_stress = reinterpret_cast<double*>(c->bounce_dev + o_stress);
            st = enqueue(m, del_t, (int64_t)np, reinterpret_cast<const double*>(c->bounce_dev + o_grad), d_stress, d_stress,
                         tangent ? reinterpret_cast<double*>(c->bounce_dev + o_tan) : nullptr, d_hist, d_hist, s, false);
            if (st != FCAMD_OK) return drain_and_return(c, st);
            if (p0 + chunk >= n && has_sparse_history(m->law))  // last chunk: the counters ride on the same wait
                HIP_TRY_DRAIN(c, hipMemcpyAsync(m->h_counters, m->d_counters, kCounterBytes, hipMemcpyDeviceToHost, s));
            HIP_TRY_DRAIN(c, hipStreamSynchronize(s));
            std::memcpy(stress + SD * p0, c->bounce + o_stress, np * SD * sizeof(double));
            if (tangent) std::memcpy(tangent + TD * p0, c->bounce + o_tan, np * TD * sizeof(double));
            for (int k = 0; k < NH; ++k) {
                const size_t d = (size_t)m->info.hist[k].dim;
                std::memcpy(hist[k] + d * p0, c->bounce + o_hist[k], np * d * sizeof(double));
            }
        }
        return finish_chunks(m, stats, /*downloaded=*/true);
    }

    if (arrays.temp_locked()) c->last_host_mode |= FCAMD_HOST_TEMP_LOCK;
    bool aligned = aligned16(z_grad) && aligned16(z_stress) && (!tangent || aligned16(z_tan));
    for (int k = 0; k < NH; ++k) aligned = aligned && aligned16(z_hist[k]);
    // Zero copy: one launch directly on the (page-locked) caller arrays -- the GPU reads the inputs and writes the
    // results over PCIe itself, both directions at once, no staging buffers.  Measured on MI355X / PCIe gen5
    // (round-2 probe zero_copy_probe.py (git history), VonMises3D): 140 instead of 117 Mpts/s at 1e7 points (55 GB/s of device-to-host
    // traffic), 40 instead of 145 us per call at 1e3 points.
    if (zero_copy_enabled(c) && aligned) {
        c->last_host_mode |= FCAMD_HOST_ZERO_COPY_IN | FCAMD_HOST_ZERO_COPY_OUT;
        hipStream_t s = c->hstream[0];
        if (has_sparse_history(m->law)) HIP_TRY(hipMemsetAsync(m->d_counters, 0, kCounterBytes, s));
        double* zh[FCAMD_MAX_HISTORY] = {reinterpret_cast<double*>(z_hist[0]), reinterpret_cast<double*>(z_hist[1])};
        double* zs = reinterpret_cast<double*>(z_stress);
        if (pool) {  // the kernel works on the other arrays in place; the tangent rows are the CPU's (fcamd_hosttangent.cpp)
            constants_for_call(m, del_t);
            auto launch = [&](int64_t p0, int64_t np, double* params, hipStream_t on) {
                double* h[FCAMD_MAX_HISTORY] = {nullptr, nullptr};
                for (int k = 0; k < NH; ++k) h[k] = zh[k] + (size_t)m->info.hist[k].dim * p0;
                return enqueue(m, del_t, np, reinterpret_cast<const double*>(z_grad) + GD2 * p0, zs + SD * p0, zs + SD * p0, params, h, h, on,
                               false, nullptr, nullptr, params ? kFlagTangentParamsHost : 0);
            };
            if (host_tangent_kind(m) == 1 + HostTangentJob::CONST) return run_const_tangent(m, pool, n, tangent, stats, launch);
            return run_param_chunks(m, pool, n, tangent, stats, launch);
        }
        st = enqueue(m, del_t, n, reinterpret_cast<const double*>(z_grad), zs, zs, reinterpret_cast<double*>(z_tan), zh, zh, s, false);
        if (st != FCAMD_OK) return drain_and_return(c, st);
        return finish_single_stream(m, stats);
    }

    // chunked DMA pipeline between the page-locked caller arrays and device buffers
    int64_t chunk = 0;
    st = prepare_chunks(c, grad, n, &chunk, /*staging=*/true, /*locked=*/true);
    if (st != FCAMD_OK) return st;
    const int nslots = c->opt.host_slots;

    HIP_TRY(hipMemsetAsync(m->d_counters, 0, kCounterBytes, c->hstream[0]));
    HIP_TRY(hipStreamSynchronize(c->hstream[0]));

    int slot = 0;
    for (int64_t p0 = 0; p0 < n; p0 += chunk, slot = (slot + 1) % nslots) {
        const int64_t np = std::min<int64_t>(chunk, n - p0);
        hipStream_t s = c->hstream[slot];
        // device layout of a slot (each sub-array starts 16-byte aligned: chunk is a multiple of 64)
        double* d_grad = c->dchunk[slot];
        double* d_stress = d_grad + 10 * c->dchunk_points;  // slots sized for FULL (9 -> 10: keeps 16-B alignment)
        double* d_tan = d_stress + 6 * c->dchunk_points;
        double* d_hist[FCAMD_MAX_HISTORY] = {nullptr, nullptr};
        double* cur = d_tan + 36 * c->dchunk_points;
        for (int k = 0; k < NH; ++k) {
            d_hist[k] = cur;
            cur += (size_t)m->info.hist[k].dim * c->dchunk_points;
            // keep 16-byte alignment for odd per-point dimensions (alpha: 1, comfe history: 7)
            if ((reinterpret_cast<uintptr_t>(cur) & 15u) != 0) cur += 1;
        }
        HIP_TRY_DRAIN(c, hipMemcpyAsync(d_grad, grad + GD2 * p0, (size_t)np * GD2 * sizeof(double), hipMemcpyHostToDevice, s));
        HIP_TRY_DRAIN(c, hipMemcpyAsync(d_stress, stress + SD * p0, (size_t)np * SD * sizeof(double), hipMemcpyHostToDevice, s));
        for (int k = 0; k < NH; ++k) {
            const size_t d = (size_t)m->info.hist[k].dim;
            HIP_TRY_DRAIN(c, hipMemcpyAsync(d_hist[k], hist[k] + d * p0, (size_t)np * d * sizeof(double), hipMemcpyHostToDevice, s));
        }
        st = enqueue(m, del_t, np, d_grad, d_stress, d_stress, tangent ? d_tan : nullptr, d_hist, d_hist, s, false);
        if (st != FCAMD_OK) return drain_and_return(c, st);
        HIP_TRY_DRAIN(c, hipMemcpyAsync(stress + SD * p0, d_stress, (size_t)np * SD * sizeof(double), hipMemcpyDeviceToHost, s));
        if (tangent)
            HIP_TRY_DRAIN(c, hipMemcpyAsync(tangent + TD * p0, d_tan, (size_t)np * TD * sizeof(double), hipMemcpyDeviceToHost, s));
        for (int k = 0; k < NH; ++k) {
            const size_t d = (size_t)m->info.hist[k].dim;
            HIP_TRY_DRAIN(c, hipMemcpyAsync(hist[k] + d * p0, d_hist[k], (size_t)np * d * sizeof(double), hipMemcpyDeviceToHost, s));
        }
    }
    return finish_chunks(m, stats);
}

int fcamd_evaluate_resident(fcamd_model* m, double t, double del_t, int64_t n, const fcamd_eval_args* x,
                            double* stress_host, double* tangent_host, fcamd_stats* stats) {
    (void)t;
    if (!x) return fail(FCAMD_ERR_BAD_ARG, "state is NULL");
    if (x->parent_rows || x->stress2 || x->tangent || x->wrapper_constraint)
        return fail(FCAMD_ERR_UNSUPPORTED, "fcamd_evaluate_resident: parent_rows / stress2 / a device tangent / the wrapper form are options of fcamd_evaluate_device_ex");
    const double* grad = x->grad_del_u;  // HOST array
    const double* stress_prev = x->stress_prev;
    double* stress = x->stress;
    const double* const* hist_prev = x->history_prev;
    double* const* hist = x->history;
    const int n_hist = x->n_hist, flags = x->flags;
    uint64_t* history_mask = x->history_mask;
    const bool packed = (flags & FCAMD_EVAL_PACKED_HISTORY) != 0;
    const unsigned long long* emask_prev = packed ? reinterpret_cast<const unsigned long long*>(x->packed_mask_prev) : nullptr;
    unsigned long long* emask = packed ? reinterpret_cast<unsigned long long*>(x->packed_mask) : nullptr;
    int st = validate_call(m, del_t, n, grad, stress_prev, stress,
                           reinterpret_cast<const void* const*>(hist_prev),
                           reinterpret_cast<const void* const*>(hist), n_hist, flags);
    if (st != FCAMD_OK) return st;
    if (history_mask && !has_sparse_history(m->law))
        return fail(FCAMD_ERR_UNSUPPORTED, "sparse trial history exists for the plasticity laws only");
    if (flags & ~(FCAMD_EVAL_SPARSE_TANGENT | FCAMD_EVAL_SPLIT_HISTORY | FCAMD_EVAL_PACKED_HISTORY))
        return fail(FCAMD_ERR_UNSUPPORTED, "unknown FCAMD_EVAL_* flag in 0x%x (2, FCAMD_EVAL_DELTA_HISTORY of ABI 0.3, was removed in 0.4)", flags);
    if (packed && n > 0) {  // as fcamd_evaluate_device_ex
        if (!history_mask || !emask_prev || !emask || emask == emask_prev)
            return fail(FCAMD_ERR_BAD_ARG, "FCAMD_EVAL_PACKED_HISTORY needs history_mask and two mask arrays, packed_mask_prev and packed_mask");
        if (m->law != FCAMD_VON_MISES_3D && !((flags & FCAMD_EVAL_SPLIT_HISTORY) && has_split_history(m->law)))
            return fail(FCAMD_ERR_UNSUPPORTED, "FCAMD_EVAL_PACKED_HISTORY: VonMises3D, or a comfe-rs plasticity law with FCAMD_EVAL_SPLIT_HISTORY");
        const int kd = (flags & FCAMD_EVAL_SPLIT_HISTORY) ? 1 : 0;
        if (hist && hist_prev && hist[kd] == hist_prev[kd])
            return fail(FCAMD_ERR_BAD_ARG, "FCAMD_EVAL_PACKED_HISTORY needs a trial plastic-strain array of its own");
    }
    if (!aligned16(stress) || !aligned16(stress_prev))
        return fail(FCAMD_ERR_ALIGN, "device arrays must be 16-byte aligned");
    // history arrays of the state: the law's fields, or -- FCAMD_EVAL_SPLIT_HISTORY -- [scalar (n), eps_p rows (6 n)]
    const bool split = (flags & FCAMD_EVAL_SPLIT_HISTORY) != 0;
    const int NH = split ? 2 : m->info.n_hist;
    size_t hdim[FCAMD_MAX_HISTORY] = {0, 0};
    for (int k = 0; k < NH; ++k) hdim[k] = split ? (k == 0 ? 1 : 6) : (size_t)m->info.hist[k].dim;
    for (int k = 0; k < NH; ++k)
        if (!aligned16(hist[k]) || !aligned16(hist_prev[k]))
            return fail(FCAMD_ERR_ALIGN, "device history arrays must be 16-byte aligned");
    fcamd_context* c = m->ctx;
    std::lock_guard<std::recursive_mutex> lock(c->host_mu);
    HostTimer timer(m);
    HIP_TRY(hipSetDevice(c->device));
    if (stats) std::memset(stats, 0, sizeof(*stats));
    HIP_TRY(hipStreamSynchronize(c->stream));  // the state arrays may have work queued on the caller's stream
    const size_t GD2 = (size_t)m->dims.gd2, SD = (size_t)m->dims.sd, TD = SD * SD;
    c->last_host_mode = 0;
    if (!c->hstream[0]) HIP_TRY(hipStreamCreateWithFlags(&c->hstream[0], hipStreamNonBlocking));
    if (n == 0) {
        if (has_sparse_history(m->law)) HIP_TRY(hipMemsetAsync(m->d_counters, 0, kCounterBytes, c->hstream[0]));
        return finish_single_stream(m, stats);
    }
    const size_t N = (size_t)n;
    const size_t bytes_per_point = (GD2 + (stress_host ? SD : 0) + (tangent_host ? TD : 0)) * sizeof(double);
    // the tangent rebuilt on the CPU (fcamd_hosttangent.cpp) -- not under the sparse-tangent protocol, whose untouched rows stay as
    // they are in the caller's array, and for the 3-D laws' one-launch pass only
    ExpandPool* pool = host_tangent_for(m, n, tangent_host);
    if ((flags & FCAMD_EVAL_SPARSE_TANGENT) || m->dims.gdim != 3) pool = nullptr;
    const bool all_registered = mapped(c, grad, N * GD2 * sizeof(double)) &&
                                (!stress_host || mapped(c, stress_host, N * SD * sizeof(double))) &&
                                (pool || !tangent_host || mapped(c, tangent_host, N * TD * sizeof(double)));
    // the host arrays of the pass: ranges the caller registered as they are, pageable ones page-locked for the
    // duration of the call, small passes through the context's page-locked scratch (CallerArrays)
    CallerArrays arrays(c);
    char *l_grad = nullptr, *l_stress = nullptr, *l_tan = nullptr;
    bool locked = false;
    if (all_registered || N * bytes_per_point > (size_t)c->opt.bounce_max) {
        locked = arrays.lock(grad, N * GD2 * sizeof(double), &l_grad) &&
                 arrays.lock(stress_host, stress_host ? N * SD * sizeof(double) : 0, &l_stress);
        if (locked && pool && !(c->opt.zero_copy_grad && aligned16(l_grad) && aligned16(l_stress))) pool = nullptr;  // no one-launch pass
        if (locked && !pool) locked = arrays.lock(tangent_host, tangent_host ? N * TD * sizeof(double) : 0, &l_tan);
        if (!locked) arrays.release();
    }
    if (!locked) pool = nullptr;
    if (!locked) {
        c->last_host_mode = FCAMD_HOST_BOUNCE;
        const int64_t chunk = bounce_chunk(c, n, bytes_per_point);
        BounceLayout lay;
        const size_t o_grad = lay.take((size_t)chunk * GD2 * sizeof(double));
        const size_t o_stress = stress_host ? lay.take((size_t)chunk * SD * sizeof(double)) : 0;
        const size_t o_tan = tangent_host ? lay.take((size_t)chunk * TD * sizeof(double)) : 0;
        st = ensure_bounce(c, lay.used);
        if (st != FCAMD_OK) return st;
        hipStream_t s = c->hstream[0];
        if (has_sparse_history(m->law)) HIP_TRY(hipMemsetAsync(m->d_counters, 0, kCounterBytes, s));
        const bool second_store = stress_host && m->dims.gdim == 3;  // the 3-D kernels can store the stress twice
        for (int64_t p0 = 0; p0 < n; p0 += chunk) {
            const size_t np = (size_t)std::min<int64_t>(chunk, n - p0);
            std::memcpy(c->bounce + o_grad, grad + GD2 * p0, np * GD2 * sizeof(double));
            const double* hp[FCAMD_MAX_HISTORY] = {nullptr, nullptr};
            double* hc[FCAMD_MAX_HISTORY] = {nullptr, nullptr};
            for (int k = 0; k < NH; ++k) {
                const size_t d = hdim[k];
                hp[k] = hist_prev[k] + d * p0;
                hc[k] = hist[k] + d * p0;
            }
            // the scratch holds no previous tangent: every row is written (no sparse tangent)
            st = enqueue(m, del_t, (int64_t)np, reinterpret_cast<const double*>(c->bounce_dev + o_grad), stress_prev + SD * p0,
                         stress + SD * p0, tangent_host ? reinterpret_cast<double*>(c->bounce_dev + o_tan) : nullptr, hp, hc, s, false,
                         nullptr, history_mask ? reinterpret_cast<unsigned long long*>(history_mask) + p0 / 64 : nullptr,
                         flags & ~FCAMD_EVAL_SPARSE_TANGENT, second_store ? reinterpret_cast<double*>(c->bounce_dev + o_stress) : nullptr,
                         nullptr, emask_prev ? emask_prev + p0 / 64 : nullptr, emask ? emask + p0 / 64 : nullptr);
            if (st != FCAMD_OK) return drain_and_return(c, st);
            if (stress_host && !second_store)
                HIP_TRY_DRAIN(c, hipMemcpyAsync(c->bounce + o_stress, stress + SD * p0, np * SD * sizeof(double), hipMemcpyDeviceToHost, s));
            if (p0 + chunk >= n && has_sparse_history(m->law))  // last chunk: the counters ride on the same wait
                HIP_TRY_DRAIN(c, hipMemcpyAsync(m->h_counters, m->d_counters, kCounterBytes, hipMemcpyDeviceToHost, s));
            HIP_TRY_DRAIN(c, hipStreamSynchronize(s));
            if (stress_host) std::memcpy(stress_host + SD * p0, c->bounce + o_stress, np * SD * sizeof(double));
            if (tangent_host) std::memcpy(tangent_host + TD * p0, c->bounce + o_tan, np * TD * sizeof(double));
        }
        return finish_chunks(m, stats, /*downloaded=*/true);
    }
    if (arrays.temp_locked()) c->last_host_mode |= FCAMD_HOST_TEMP_LOCK;
    const bool zc = zero_copy_enabled(c);
    const double* z_grad = (zc && c->opt.zero_copy_grad && aligned16(l_grad)) ? reinterpret_cast<const double*>(l_grad) : nullptr;
    double* z_tan = (zc && tangent_host && aligned16(l_tan)) ? reinterpret_cast<double*>(l_tan) : nullptr;
    c->last_host_mode |= (z_grad ? FCAMD_HOST_ZERO_COPY_IN : 0) | (z_tan ? FCAMD_HOST_ZERO_COPY_OUT : 0);
    // Everything the pass moves lies in page-locked caller memory and the law is a 3-D one (whose stress
    // store can feed two destinations): ONE launch reads the gradient from and writes stress and tangent
    // to the host arrays while it updates the device-resident state -- no chunks, no copies.
    double* z_stress = (zc && stress_host && m->dims.gdim == 3 && aligned16(l_stress)) ? reinterpret_cast<double*>(l_stress) : nullptr;
    if (z_grad && (z_tan || !tangent_host || pool) && (z_stress || !stress_host) && m->dims.gdim == 3) {
        hipStream_t s = c->hstream[0];
        if (has_sparse_history(m->law)) HIP_TRY(hipMemsetAsync(m->d_counters, 0, kCounterBytes, s));
        if (pool) {  // state on the device, gradient and stress over the link in place, the tangent rows from the CPU
            if (z_stress) c->last_host_mode |= FCAMD_HOST_ZERO_COPY_OUT;  // (the stress is the kernel's own store into the caller's array)
            constants_for_call(m, del_t);
            auto launch = [&](int64_t p0, int64_t np, double* params, hipStream_t on) {
                const double* hp[FCAMD_MAX_HISTORY] = {nullptr, nullptr};
                double* hc[FCAMD_MAX_HISTORY] = {nullptr, nullptr};
                for (int k = 0; k < NH; ++k) {
                    hp[k] = hist_prev[k] + hdim[k] * p0;
                    hc[k] = hist[k] + hdim[k] * p0;
                }
                return enqueue(m, del_t, np, z_grad + GD2 * p0, stress_prev + SD * p0, stress + SD * p0, params, hp, hc, on, false, nullptr,
                               history_mask ? reinterpret_cast<unsigned long long*>(history_mask) + p0 / 64 : nullptr,
                               flags | (params ? kFlagTangentParamsHost : 0), z_stress ? z_stress + SD * p0 : nullptr, nullptr,
                               emask_prev ? emask_prev + p0 / 64 : nullptr, emask ? emask + p0 / 64 : nullptr);
            };
            if (host_tangent_kind(m) == 1 + HostTangentJob::CONST) return run_const_tangent(m, pool, n, tangent_host, stats, launch);
            return run_param_chunks(m, pool, n, tangent_host, stats, launch);
        }
        st = enqueue(m, del_t, n, z_grad, stress_prev, stress, z_tan, hist_prev, hist, s, false, nullptr,
                     reinterpret_cast<unsigned long long*>(history_mask), flags, z_stress, nullptr, emask_prev, emask);
        if (st != FCAMD_OK) return drain_and_return(c, st);
        return finish_single_stream(m, stats);
    }
    int64_t chunk = 0;
    st = prepare_chunks(c, grad, n, &chunk, /*staging=*/!(z_grad && (z_tan || !tangent_host)), /*locked=*/true);
    if (st != FCAMD_OK) return st;
    const int nslots = c->opt.host_slots;
    HIP_TRY(hipMemsetAsync(m->d_counters, 0, kCounterBytes, c->hstream[0]));
    HIP_TRY(hipStreamSynchronize(c->hstream[0]));
    int slot = 0;
    for (int64_t p0 = 0; p0 < n; p0 += chunk, slot = (slot + 1) % nslots) {
        const int64_t np = std::min<int64_t>(chunk, n - p0);
        hipStream_t s = c->hstream[slot];
        double* d_grad = c->dchunk[slot];  // unused (possibly null) when gradient and tangent are zero copy
        double* d_tan = d_grad ? d_grad + 10 * c->dchunk_points : nullptr;
        // chunk offsets are multiples of 64 points: every sub-array stays 16-byte aligned and the
        // per-tile mask words line up
        const double* hp[FCAMD_MAX_HISTORY] = {nullptr, nullptr};
        double* hc[FCAMD_MAX_HISTORY] = {nullptr, nullptr};
        for (int k = 0; k < NH; ++k) {
            const size_t d = hdim[k];
            hp[k] = hist_prev[k] + d * p0;
            hc[k] = hist[k] + d * p0;
        }
        // page-locked, GPU-mapped caller arrays are read / written by the kernel itself (zero copy)
        const double* k_grad = z_grad ? z_grad + GD2 * p0 : d_grad;
        double* k_tan = !tangent_host ? nullptr : (z_tan ? z_tan + TD * p0 : d_tan);
        if (!z_grad)
            HIP_TRY_DRAIN(c, hipMemcpyAsync(d_grad, grad + GD2 * p0, (size_t)np * GD2 * sizeof(double), hipMemcpyHostToDevice, s));
        st = enqueue(m, del_t, np, k_grad, stress_prev + SD * p0, stress + SD * p0, k_tan,
                     hp, hc, s, false, nullptr,
                     history_mask ? reinterpret_cast<unsigned long long*>(history_mask) + p0 / 64 : nullptr,
                     z_tan ? flags : (flags & ~FCAMD_EVAL_SPARSE_TANGENT),  // the staging buffer of a chunk holds no previous tangent: full rows
                     nullptr, nullptr, emask_prev ? emask_prev + p0 / 64 : nullptr, emask ? emask + p0 / 64 : nullptr);
        if (st != FCAMD_OK) return drain_and_return(c, st);
        if (stress_host)
            HIP_TRY_DRAIN(c, hipMemcpyAsync(stress_host + SD * p0, stress + SD * p0, (size_t)np * SD * sizeof(double),
                                            hipMemcpyDeviceToHost, s));
        if (tangent_host && !z_tan)
            HIP_TRY_DRAIN(c, hipMemcpyAsync(tangent_host + TD * p0, d_tan, (size_t)np * TD * sizeof(double),
                                            hipMemcpyDeviceToHost, s));
    }
    return finish_chunks(m, stats);
}

}  // extern "C"

namespace {

// One synchronous copy between caller host memory and device memory, ordered after the work queued on the
// context stream, by the rules of the host entries (CallerArrays): never the runtime's pageable-copy path.
int host_copy(fcamd_context* c, char* dev, char* host, size_t bytes, bool to_device) {
    if (!c || (bytes && (!dev || !host))) return fail(FCAMD_ERR_BAD_ARG, "NULL argument");
    if (bytes == 0) return FCAMD_OK;
    std::lock_guard<std::recursive_mutex> lock(c->host_mu);
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    CallerArrays arrays(c);
    char* z = nullptr;
    if ((mapped(c, host, 8) || bytes > (size_t)c->opt.bounce_max) && arrays.lock(host, bytes, &z)) {
        const hipError_t e = to_device ? hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s)
                                       : hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s);
        const hipError_t e2 = hipStreamSynchronize(s);  // before the arrays are unlocked, whatever happened
        if (e != hipSuccess || e2 != hipSuccess) {
            (void)hipGetLastError();
            return fail(FCAMD_ERR_HIP, "copy between page-locked host memory and the device failed: %s",
                        hipGetErrorString(e != hipSuccess ? e : e2));
        }
        return FCAMD_OK;
    }
    arrays.release();
    const size_t piece = std::min(bytes, std::max<size_t>((size_t)c->opt.bounce_max, kBounceChunkBytes));
    int st = ensure_bounce(c, piece);
    if (st != FCAMD_OK) return st;
    for (size_t off = 0; off < bytes; off += piece) {
        const size_t nb = std::min(piece, bytes - off);
        if (to_device) {
            std::memcpy(c->bounce, host + off, nb);
            HIP_TRY(hipMemcpyAsync(dev + off, c->bounce, nb, hipMemcpyHostToDevice, s));
            HIP_TRY(hipStreamSynchronize(s));
        } else {
            HIP_TRY(hipMemcpyAsync(c->bounce, dev + off, nb, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            std::memcpy(host + off, c->bounce, nb);
        }
    }
    return FCAMD_OK;
}

}  // namespace

extern "C" {

static int copy_device(fcamd_context* c, void* dst_device, const void* src_device, size_t bytes) {
    if (!c || (bytes && (!dst_device || !src_device))) return fail(FCAMD_ERR_BAD_ARG, "NULL argument");
    if (bytes == 0) return FCAMD_OK;
    HIP_TRY(hipSetDevice(c->device));
    if (!aligned16(dst_device) || !aligned16(src_device))
        return fail(FCAMD_ERR_ALIGN, "device arrays must be 16-byte aligned");
    const size_t n16 = bytes / 16;
    // 16 KiB per workgroup and pass.  round-2 probe stream_copy_probe.py (git history), 8 GiB buffers, three buffer pairs: 5.0 - 5.6 TB/s with 1024
    // workgroups, 5.5 - 5.9 with 8192, 5.8 - 6.0 with 262144 (torch's copy: 4.7 - 5.0): many short workgroups
    const size_t tiles = (n16 + 1023) / 1024;
    const int grid = c->grid_override > 0 ? c->grid_override : (int)std::min<size_t>(tiles, (size_t)c->num_cu * 1024);
    if (n16) HIP_TRY(launch_stream_copy(dst_device, src_device, n16, std::max(grid, 1), c->stream));
    if (bytes % 16)
        HIP_TRY(hipMemcpyAsync(static_cast<char*>(dst_device) + 16 * n16, static_cast<const char*>(src_device) + 16 * n16, bytes % 16,
                               hipMemcpyDeviceToDevice, c->stream));
    return FCAMD_OK;
}

int fcamd_copy(fcamd_context* c, void* dst, const void* src, size_t bytes, int kind) {
    switch (kind) {
        case FCAMD_COPY_TO_DEVICE: return host_copy(c, static_cast<char*>(dst), static_cast<char*>(const_cast<void*>(src)), bytes, true);
        case FCAMD_COPY_TO_HOST: return host_copy(c, static_cast<char*>(const_cast<void*>(src)), static_cast<char*>(dst), bytes, false);
        case FCAMD_COPY_DEVICE: return copy_device(c, dst, src, bytes);
        default: return fail(FCAMD_ERR_BAD_ARG, "unknown copy kind %d", kind);
    }
}

}  // extern "C"
