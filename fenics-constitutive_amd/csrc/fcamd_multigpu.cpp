// Multi-GPU part of the C ABI (include/fcamd.h, "multi-GPU"): contiguous 64-aligned shards of the
// quadrature-point axis, buffers that peers can map, and the direct (one-hop) all-gather of stress /
// tangent slices over xGMI as world-1 concurrent peer copies.  The evaluation itself
// needs no collective -- every point is independent -- so nothing here touches the kernels.
//
// The reference's only inter-rank exchange is dolfinx's ghost forwarding
// (src/fenics_constitutive/solver/_solver.py:146-147, solver/maps.py:47,59,101,123); the gather exists
// for the single-assembler mode of BASELINE.json's north_star only.
#include <algorithm>
#include <cstring>

#include "fcamd_host.h"

using namespace fcamd;

namespace {

constexpr int64_t kTile = 64;  // one wavefront tile: slices start on tile boundaries

int64_t slot_points(int64_t n, int world) {
    int64_t per = (n + world - 1) / world;
    return ((per + kTile - 1) / kTile) * kTile;
}

// hipIpcOpenMemHandle of this stack (ROCm 7.2, dmabuf IPC) never returns -- both processes spin in user space --
// when the exported ALLOCATION's size has bit 31 set, i.e. (size mod 4 GiB) >= 2 GiB: 2, 3, 6, 7, 10, 10.5, 10.73,
// 11 GiB hang; 1, 4, 5, 5.4, 8, 9, 12, 13, 16, 32, 64 GiB map in under a millisecond (tools/ipc_open_probe.py).
// Such an allocation is refused at export, and fcamd_ipc_alloc rounds a request up to the next safe size.
constexpr size_t kFourGiB = (size_t)4 << 30, kTwoGiB = (size_t)2 << 30;
bool ipc_size_is_safe(size_t bytes) { return (bytes % kFourGiB) < kTwoGiB; }
size_t ipc_safe_size(size_t bytes) { return ipc_size_is_safe(bytes) ? bytes : ((bytes + kFourGiB - 1) / kFourGiB) * kFourGiB; }

int ensure_peer_streams(fcamd_context* c, int world) {
    while ((int)c->peer_streams.size() < world) {
        hipStream_t s = nullptr;
        hipEvent_t e = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        c->peer_streams.push_back(s);
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->peer_events.push_back(e);
    }
    return FCAMD_OK;
}

}  // namespace

extern "C" {

int fcamd_shard_bounds(int64_t n, int world, int rank, int64_t* lo, int64_t* hi, int64_t* slot) {
    if (!lo || !hi) return fail(FCAMD_ERR_BAD_ARG, "NULL argument");
    if (n < 0 || world <= 0 || rank < 0 || rank >= world)
        return fail(FCAMD_ERR_BAD_ARG, "n >= 0, world > 0 and 0 <= rank < world expected");
    const int64_t per = slot_points(n, world);
    *lo = std::min<int64_t>((int64_t)rank * per, n);
    *hi = std::min<int64_t>(*lo + per, n);
    if (slot) *slot = per;
    return FCAMD_OK;
}

int fcamd_gather_chunk_plan(int64_t slot_pts, int world, int values_per_point, size_t budget_bytes, int n_buffers,
                            int64_t* chunk_points, int64_t* n_chunks) {
    if (!chunk_points || !n_chunks) return fail(FCAMD_ERR_BAD_ARG, "NULL argument");
    if (slot_pts < 0 || world <= 0 || values_per_point <= 0 || n_buffers <= 0)
        return fail(FCAMD_ERR_BAD_ARG, "bad plan arguments");
    if (slot_pts == 0) {
        *chunk_points = 0;
        *n_chunks = 0;
        return FCAMD_OK;
    }
    // one chunk buffer holds `world` slices of chunk_points points
    const size_t per_point = (size_t)world * (size_t)values_per_point * sizeof(double) * (size_t)n_buffers;
    int64_t c = (int64_t)(budget_bytes / per_point);
    c = (c / kTile) * kTile;
    if (c < kTile)
        return fail(FCAMD_ERR_SIZE, "gather budget of %zu bytes holds not even one 64-point tile per rank (%zu bytes needed)",
                    budget_bytes, per_point * (size_t)kTile);
    c = std::min<int64_t>(c, ((slot_pts + kTile - 1) / kTile) * kTile);
    // equalise the chunks: same count, smallest tile-aligned length
    const int64_t k = (slot_pts + c - 1) / c;
    c = (((slot_pts + k - 1) / k + kTile - 1) / kTile) * kTile;
    *chunk_points = c;
    *n_chunks = (slot_pts + c - 1) / c;
    return FCAMD_OK;
}

int fcamd_ipc_export(fcamd_context* c, const void* device_ptr, unsigned char handle[FCAMD_IPC_HANDLE_BYTES],
                     size_t* offset_bytes) {
    if (!c || !device_ptr || !handle || !offset_bytes) return fail(FCAMD_ERR_BAD_ARG, "NULL argument");
    static_assert(sizeof(hipIpcMemHandle_t) <= FCAMD_IPC_HANDLE_BYTES, "IPC handle does not fit");
    HIP_TRY(hipSetDevice(c->device));
    // the handle names the allocation; the pointer may lie inside it (a caching allocator's sub-block)
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    HIP_TRY(hipMemGetAddressRange(&base, &size, const_cast<void*>(device_ptr)));
    if (!ipc_size_is_safe(size))
        return fail(FCAMD_ERR_UNSUPPORTED,
                    "the allocation behind this pointer has %zu bytes: (size mod 4 GiB) >= 2 GiB, which hipIpcOpenMemHandle of this "
                    "ROCm stack cannot map (it never returns); allocate shared buffers with fcamd_device_alloc_set(FCAMD_ALLOC_IPC)", size);
    hipIpcMemHandle_t h;
    HIP_TRY(hipIpcGetMemHandle(&h, base));
    std::memset(handle, 0, FCAMD_IPC_HANDLE_BYTES);
    std::memcpy(handle, &h, sizeof(h));
    *offset_bytes = (size_t)(static_cast<const char*>(device_ptr) - static_cast<const char*>(base));
    return FCAMD_OK;
}

int fcamd_ipc_open(fcamd_context* c, const unsigned char handle[FCAMD_IPC_HANDLE_BYTES], size_t offset_bytes,
                   void** device_ptr) {
    if (!c || !handle || !device_ptr) return fail(FCAMD_ERR_BAD_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    std::lock_guard<std::recursive_mutex> lock(c->host_mu);
    const std::string key(reinterpret_cast<const char*>(handle), FCAMD_IPC_HANDLE_BYTES);
    auto it = c->ipc_open.find(key);
    if (it == c->ipc_open.end()) {
        hipIpcMemHandle_t h;
        std::memcpy(&h, handle, sizeof(h));
        void* base = nullptr;
        HIP_TRY(hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess));
        it = c->ipc_open.emplace(key, fcamd_context::IpcMapping{base, 0}).first;
    }
    ++it->second.refs;
    *device_ptr = static_cast<char*>(it->second.base) + offset_bytes;
    return FCAMD_OK;
}

int fcamd_ipc_close(fcamd_context* c, void* device_ptr, size_t offset_bytes) {
    if (!c || !device_ptr) return fail(FCAMD_ERR_BAD_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    std::lock_guard<std::recursive_mutex> lock(c->host_mu);
    void* base = static_cast<char*>(device_ptr) - offset_bytes;
    for (auto it = c->ipc_open.begin(); it != c->ipc_open.end(); ++it) {
        if (it->second.base != base) continue;
        if (--it->second.refs == 0) {
            c->ipc_open.erase(it);
            HIP_TRY(hipIpcCloseMemHandle(base));
        }
        return FCAMD_OK;
    }
    return fail(FCAMD_ERR_BAD_ARG, "pointer was not returned by fcamd_ipc_open of this context");
}

}  // extern "C"

namespace fcamd {
size_t ipc_safe_alloc_size(size_t bytes) { return ipc_safe_size(bytes); }

int enable_peer_access(fcamd_context* c, int peer_device) {
    if (!c) return fail(FCAMD_ERR_BAD_ARG, "context is NULL");
    if (peer_device == c->device) return FCAMD_OK;
    HIP_TRY(hipSetDevice(c->device));
    int can = 0;
    HIP_TRY(hipDeviceCanAccessPeer(&can, c->device, peer_device));
    if (!can) return fail(FCAMD_ERR_UNSUPPORTED, "device %d cannot access device %d", c->device, peer_device);
    const hipError_t e = hipDeviceEnablePeerAccess(peer_device, 0);
    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
        (void)hipGetLastError();
        return fail(FCAMD_ERR_HIP, "hipDeviceEnablePeerAccess(%d) failed: %s", peer_device, hipGetErrorString(e));
    }
    (void)hipGetLastError();
    return FCAMD_OK;
}
}  // namespace fcamd

extern "C" {

int fcamd_allgather_direct(fcamd_context* c, int world, int rank, void* const* gathered, const int* devices,
                           size_t slot_bytes, size_t offset_bytes, size_t bytes, int flags) {
    if (!c || !gathered) return fail(FCAMD_ERR_BAD_ARG, "NULL argument");
    if (world <= 0 || rank < 0 || rank >= world) return fail(FCAMD_ERR_BAD_ARG, "0 <= rank < world expected");
    if (offset_bytes + bytes > slot_bytes) return fail(FCAMD_ERR_SIZE, "offset + bytes exceeds the slot");
    for (int p = 0; p < world; ++p)
        if (!gathered[p]) return fail(FCAMD_ERR_BAD_ARG, "gathered[%d] is NULL", p);
    if (world == 1 || bytes == 0) return FCAMD_OK;
    HIP_TRY(hipSetDevice(c->device));
    int st = ensure_peer_streams(c, world);
    if (st != FCAMD_OK) return st;
    // the slot is produced on the context's stream (the evaluate kernel): the copies start after it
    hipEvent_t ready = c->peer_events[rank];
    HIP_TRY(hipEventRecord(ready, c->stream));
    const bool pull = (flags & FCAMD_GATHER_PULL) != 0;
    // staggered peer order: in step s rank r talks to r+s, so at any time every rank is the target of one
    // copy per step and the 7 transfers of a rank leave over 7 different xGMI links
    for (int shift = 1; shift < world; ++shift) {
        const int p = (rank + shift) % world;
        hipStream_t s = c->peer_streams[p];
        HIP_TRY(hipStreamWaitEvent(s, ready, 0));
        // push: my slot -> the same slot of peer p's buffer;  pull: peer p's slot of its buffer -> my buffer
        const int slot = pull ? p : rank;
        char* dst = static_cast<char*>(gathered[pull ? rank : p]) + (size_t)slot * slot_bytes + offset_bytes;
        const char* src = static_cast<const char*>(gathered[pull ? p : rank]) + (size_t)slot * slot_bytes + offset_bytes;
        if (devices && devices[p] != devices[rank]) {
            const int dst_dev = pull ? devices[rank] : devices[p], src_dev = pull ? devices[p] : devices[rank];
            HIP_TRY(hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, s));
        } else {
            HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
        }
    }
    return FCAMD_OK;
}

int fcamd_allgather_direct_wait(fcamd_context* c, int host_sync) {
    if (!c) return fail(FCAMD_ERR_BAD_ARG, "context is NULL");
    HIP_TRY(hipSetDevice(c->device));
    for (size_t p = 0; p < c->peer_streams.size(); ++p) {
        if (host_sync) {
            HIP_TRY(hipStreamSynchronize(c->peer_streams[p]));
        } else {  // stream-ordered: later work on the context's stream starts after this rank's copies
            HIP_TRY(hipEventRecord(c->peer_events[p], c->peer_streams[p]));
            HIP_TRY(hipStreamWaitEvent(c->stream, c->peer_events[p], 0));
        }
    }
    return FCAMD_OK;
}

}  // extern "C"
