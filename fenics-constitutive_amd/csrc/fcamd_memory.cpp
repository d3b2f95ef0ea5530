// Device-memory placement part of the C ABI (include/fcamd.h, "device memory").
//
// On MI355X the time of the evaluate kernels follows WHERE the driver places the arrays the kernel
// writes ("Run-to-run variance": the same tangent array costs 8.3 or 9.9 ms depending on the
// allocation it came from, carried by HBM write-credit / L2 tag stalls).  A plain hipMalloc gives the
// caller no say in that.  fcamd_device_alloc_set builds a whole working set through the virtual-memory
// API instead: one address range per array, backed by physical handles of `granule` bytes that are
// created in an interleaved order over all arrays of the set (proportionally to their sizes), so that the
// physical placement of the arrays relative to each other is a property of the call, not of the
// allocator's history.  Whether that removes the lottery is an experiment (round-2 probe vmm_placement_probe.py (git history),
// profiles/r02_placement_vmm.md), not a promise.
#include <algorithm>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "fcamd_host.h"

using namespace fcamd;

namespace {

struct VmmArray {
    int device = 0;
    size_t bytes = 0;     // reserved / mapped size (multiple of the granule)
    size_t granule = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
};

std::mutex g_vmm_mu;
std::unordered_map<void*, VmmArray> g_vmm;  // base address -> its mapping
std::unordered_set<void*> g_plain;        // FCAMD_ALLOC_IPC blocks (plain hipMalloc)

// returns the first HIP error met (everything is attempted regardless)
hipError_t release(void* base, VmmArray& a) {
    hipError_t first = hipSuccess;
    for (size_t g = 0; g < a.handles.size(); ++g) {
        const hipError_t e = hipMemUnmap(static_cast<char*>(base) + g * a.granule, a.granule);
        if (e != hipSuccess && first == hipSuccess) first = e;
    }
    for (auto h : a.handles) {
        const hipError_t e = hipMemRelease(h);
        if (e != hipSuccess && first == hipSuccess) first = e;
    }
    // The address range is NOT returned (hipMemAddressFree): on this stack a range that is reserved again at
    // the same address and mapped to new handles serves stale data -- the second of two identical sets built
    // after freeing the first read back wrong values in every trial (round-2 probe vmm_placement_probe.py (git history) --selftest
    // reproduces it with the free enabled).  Address space is plentiful (the physical memory IS released);
    // a range is simply never reused.
    a.handles.clear();
    (void)hipGetLastError();
    return first;
}

}  // namespace

extern "C" {

int fcamd_device_alloc_set(fcamd_context* c, int n_arrays, const size_t* bytes, size_t granule_bytes, int order,
                           void** ptrs) {
    if (!c || !bytes || !ptrs || n_arrays <= 0) return fail(FCAMD_ERR_BAD_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    if (order == FCAMD_ALLOC_IPC) {  // plain hipMalloc blocks that peers can map (fcamd_ipc_export)
        for (int k = 0; k < n_arrays; ++k) ptrs[k] = nullptr;
        for (int k = 0; k < n_arrays; ++k) {
            if (bytes[k] == 0) continue;
            const hipError_t e = hipMalloc(&ptrs[k], ipc_safe_alloc_size(bytes[k]));
            if (e != hipSuccess) {
                (void)hipGetLastError();
                for (int j = 0; j < k; ++j)
                    if (ptrs[j]) (void)hipFree(ptrs[j]);
                return fail(FCAMD_ERR_HIP, "hipMalloc of an IPC buffer of %zu bytes failed: %s", bytes[k], hipGetErrorString(e));
            }
            std::lock_guard<std::mutex> lock(g_vmm_mu);
            g_plain.insert(ptrs[k]);
        }
        return FCAMD_OK;
    }
    hipMemAllocationProp prop;
    std::memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = c->device;
    size_t min_granule = 0;
    HIP_TRY(hipMemGetAllocationGranularity(&min_granule, &prop, hipMemAllocationGranularityRecommended));
    if (min_granule == 0) min_granule = 4096;
    // default: 2 MiB handles (the page-table fragment size of large allocations).  The runtime's own
    // "recommended" granularity is 4 KiB on this stack -- a million handles for a 4 GB array.
    size_t granule = granule_bytes ? granule_bytes : std::max<size_t>(min_granule, (size_t)2 << 20);
    granule = ((granule + min_granule - 1) / min_granule) * min_granule;

    std::vector<VmmArray> arrs((size_t)n_arrays);
    std::vector<void*> bases((size_t)n_arrays, nullptr);
    std::vector<size_t> total((size_t)n_arrays), done((size_t)n_arrays, 0);
    auto cleanup = [&]() {
        for (int k = 0; k < n_arrays; ++k) {
            // only the granules created so far are mapped: unmap them one by one
            for (size_t g = 0; g < arrs[k].handles.size(); ++g)
                (void)hipMemUnmap(static_cast<char*>(bases[k]) + g * granule, granule);
            for (auto h : arrs[k].handles) (void)hipMemRelease(h);
            // the address range is kept (never reused), as in release(): see there
        }
        (void)hipGetLastError();
    };
#define VMM_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            cleanup();                                                                              \
            return fail(FCAMD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
        }                                                                                           \
    } while (0)

    size_t all = 0;
    for (int k = 0; k < n_arrays; ++k) {
        if (bytes[k] == 0) {
            cleanup();
            return fail(FCAMD_ERR_BAD_ARG, "array %d has zero bytes", k);
        }
        total[k] = (bytes[k] + granule - 1) / granule;
        all += total[k];
        arrs[k].device = c->device;
        arrs[k].granule = granule;
        arrs[k].bytes = total[k] * granule;
        arrs[k].handles.reserve(total[k]);
        VMM_TRY(hipMemAddressReserve(&bases[k], arrs[k].bytes, granule, nullptr, 0));
    }
    hipMemAccessDesc acc;
    std::memset(&acc, 0, sizeof(acc));
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = c->device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    // creation order of the physical handles:
    //   order 0 (FCAMD_ALLOC_SEQUENTIAL): array after array;
    //   order 1 (FCAMD_ALLOC_INTERLEAVED): always the array that is furthest behind its share, so the
    //   handles of all arrays are created side by side in proportion to their sizes
    for (size_t step = 0; step < all; ++step) {
        int k = -1;
        if (order == FCAMD_ALLOC_INTERLEAVED) {
            double worst = 2.0;
            for (int j = 0; j < n_arrays; ++j) {
                if (done[j] >= total[j]) continue;
                const double frac = (double)done[j] / (double)total[j];
                if (frac < worst) worst = frac, k = j;
            }
        } else {
            for (int j = 0; j < n_arrays && k < 0; ++j)
                if (done[j] < total[j]) k = j;
        }
        hipMemGenericAllocationHandle_t h;
        VMM_TRY(hipMemCreate(&h, granule, &prop, 0));
        arrs[k].handles.push_back(h);
        char* va = static_cast<char*>(bases[k]) + done[k] * granule;
        VMM_TRY(hipMemMap(va, granule, 0, h, 0));
        // Access is granted per mapped handle.  One hipMemSetAccess over a range of several handles is
        // accepted by this runtime but maps the range as if its handles had been created back to back:
        // with an interleaved creation order the arrays of the set then alias each other's physical memory
        // (found with distinct fill patterns, round-2 probe vmm_placement_probe.py (git history) --selftest).
        VMM_TRY(hipMemSetAccess(va, granule, &acc, 1));
        ++done[k];
    }
#undef VMM_TRY
    {
        std::lock_guard<std::mutex> lock(g_vmm_mu);
        for (int k = 0; k < n_arrays; ++k) {
            g_vmm[bases[k]] = std::move(arrs[k]);
            ptrs[k] = bases[k];
        }
    }
    return FCAMD_OK;
}

int fcamd_device_free(fcamd_context* c, void* ptr) {
    if (!c) return fail(FCAMD_ERR_BAD_ARG, "context is NULL");
    if (!ptr) return FCAMD_OK;
    VmmArray a;
    bool plain = false;
    {
        // The block leaves the table BEFORE the runtime gets it back: once hipFree has returned, another thread's hipMalloc may be
        // handed the same address, and an erase after the free would untrack that new, live block.
        std::lock_guard<std::mutex> lock(g_vmm_mu);
        plain = g_plain.erase(ptr) != 0;
    }
    if (plain) {  // an FCAMD_ALLOC_IPC block.  hipFree waits for the device: not under the process-wide lock
        hipError_t e = hipSetDevice(c->device);
        if (e == hipSuccess) e = hipFree(ptr);
        if (e != hipSuccess) {  // still allocated: tracked again, so that the free can be tried again
            (void)hipGetLastError();
            std::lock_guard<std::mutex> lock(g_vmm_mu);
            g_plain.insert(ptr);
            return fail(FCAMD_ERR_HIP, "hipFree of an FCAMD_ALLOC_IPC block failed: %s", hipGetErrorString(e));
        }
        return FCAMD_OK;
    }
    {
        std::lock_guard<std::mutex> lock(g_vmm_mu);
        auto it = g_vmm.find(ptr);
        if (it == g_vmm.end()) return fail(FCAMD_ERR_BAD_ARG, "pointer was not returned by fcamd_device_alloc_set");
        a = std::move(it->second);
        g_vmm.erase(it);
    }
    HIP_TRY(hipSetDevice(a.device));
    HIP_TRY(hipDeviceSynchronize());  // nothing may still be using the range
    const hipError_t e = release(ptr, a);
    if (e != hipSuccess) return fail(FCAMD_ERR_HIP, "releasing a VMM array failed: %s", hipGetErrorString(e));
    return FCAMD_OK;
}

}  // extern "C"
