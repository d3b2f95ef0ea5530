/*
 * fcamd_multi.h -- the multi-GPU forms of the C ABI of libfcamd (core: fcamd.h).
 *
 * UNMEASURED ON MULTI-GPU HARDWARE.  Everything below has run on ONE MI355X only -- several contexts on one device, two ranks
 * sharing a device over HIP IPC, eight device contexts in one process, world 8 over gloo on the CPU (tests/test_gpu_sharded.py,
 * test_gpu_multidevice.py, test_sharded_gloo.py) -- because no multi-GPU node was available in any round of this build: no scaling
 * curve exists, RCCL has seen world 1, fcamd_allgather_direct has never crossed xGMI.  The entry points are kept apart from the
 * measured core (fcamd.h: 26 symbols) so that a reader of the boundary sees which is which.
 *
 * Reference analogue: a rank-per-GPU dolfinx run needs none of this (every rank assembles its own cells; only the ghost
 * scatter_forward of solver/_solver.py:146-147 remains); the all-gather and the fcamd_multi handle serve ONE assembling process
 * that drives several GPUs (BASELINE.json north_star: contiguous slices, all-gather of stress and tangent).
 */
#ifndef FCAMD_MULTI_H
#define FCAMD_MULTI_H

#include "fcamd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- multi-GPU: contiguous shards + all-gather ------------------------ */
/* The quadrature-point axis [0, n) is cut into `world` contiguous slices that start on 64-point
   (wavefront-tile) boundaries and are padded to one common length, the SLOT (`slot_points`, nullable):
   rank r owns [lo, hi) = [min(r*slot, n), min(r*slot + slot, n)); trailing ranks may be empty.  Every
   array slices by dim*lo.  Evaluation needs no collective and the history stays sharded; the only
   exchange is the optional all-gather of stress (6/pt) and tangent (36/pt) for ONE assembling process
   (under dolfinx/MPI every rank assembles its own cells: the reference's only exchange is its ghost
   forwarding, solver/_solver.py:146-147).  A gathered buffer has world*slot*dim doubles, rank r's slice
   in slot r; because every rank before the last non-empty one is full, its first dim*n doubles are the
   global array. */
FCAMD_API int fcamd_shard_bounds(int64_t n, int world, int rank, int64_t* lo, int64_t* hi, int64_t* slot_points);

/* Chunked gather for shards whose gathered tangent does not fit next to the working set (config 5:
   8 x 1e8 points = 230 GB of gathered tangent per GPU): the assembler consumes the gathered array chunk
   by chunk, chunk k = points [k*chunk, (k+1)*chunk) of EVERY rank's slot, through `n_buffers` chunk
   buffers of world*chunk*values_per_point doubles each (2 = double buffering).  Returns the largest
   tile-aligned chunk length whose buffers fit into `budget_bytes` (then equalised over the chunks) and
   the number of chunks; FCAMD_ERR_SIZE if not even one tile per rank fits -- the budget is checked up
   front, before anything is allocated. */
FCAMD_API int fcamd_gather_chunk_plan(int64_t slot_points, int world, int values_per_point, size_t budget_bytes,
                                      int n_buffers, int64_t* chunk_points, int64_t* n_chunks);

/* One rank per process: a peer's gathered buffer is mapped through HIP IPC.  fcamd_ipc_export gives
   the handle of the ALLOCATION `device_ptr` lies in plus its offset inside it (the pointer may be a
   sub-block of a caching allocator, e.g. a torch tensor); the 64 bytes + offset travel to the peers by
   any host channel (torch.distributed.all_gather_object, MPI); fcamd_ipc_open maps it there.  A mapping
   is closed with the same offset it was opened with, before the owner frees the memory.
   Buffers meant to be mapped by peers come from fcamd_device_alloc_set(order = FCAMD_ALLOC_IPC): on this ROCm stack
   hipIpcOpenMemHandle never returns for an allocation whose size has bit 31 set ((size mod 4 GiB) >= 2 GiB;
   measured, tools/ipc_open_probe.py), so such a request is rounded up to the next multiple of 4 GiB there, and
   fcamd_ipc_export refuses (FCAMD_ERR_UNSUPPORTED) a pointer whose allocation has such a size instead of
   letting the peers hang. */
#define FCAMD_IPC_HANDLE_BYTES 64
FCAMD_API int fcamd_ipc_export(fcamd_context* ctx, const void* device_ptr, unsigned char handle[FCAMD_IPC_HANDLE_BYTES],
                               size_t* offset_bytes);
FCAMD_API int fcamd_ipc_open(fcamd_context* ctx, const unsigned char handle[FCAMD_IPC_HANDLE_BYTES], size_t offset_bytes,
                             void** device_ptr);
FCAMD_API int fcamd_ipc_close(fcamd_context* ctx, void* device_ptr, size_t offset_bytes);

/* Direct (one-hop) all-gather, in place: `gathered[p]` is the address, in THIS process, of rank p's
   gathered buffer (own allocation for p == rank; an IPC mapping or an allocation on another device of
   this process otherwise -- context option "peer_access" = that device lets the context's device reach it), `devices[p]`
   its HIP device ordinal (NULL: all on the context's device or
   reachable by unified addressing).  Bytes [offset_bytes, offset_bytes + bytes) of a slot of slot_bytes
   are exchanged (offset / bytes select one chunk of a chunked gather).
     push (default):    this rank's slot is copied into the same slot of every peer's buffer;
     FCAMD_GATHER_PULL: every peer's slot is copied from the peer's buffer into this rank's buffer.
   world-1 copies on world-1 streams, issued in staggered peer order (step s: rank r <-> rank r+s), so
   the transfers of one GPU leave over different xGMI links at once -- a ring all-gather forwards every
   slice world-1 times over one link per step.  The copies start after the work queued on the context's
   stream (the evaluate that produces the slot).  Asynchronous: fcamd_allgather_direct_wait(host_sync=1)
   blocks until THIS rank's copies have completed, (host_sync=0) makes the context's stream wait for
   them.  Completion of the peers' copies into / out of this rank's buffer is the caller's cross-rank
   barrier: after wait + barrier every rank's buffer is complete (push); before a pull every rank must
   have passed a barrier after producing its slot.  Before a PUSH every rank must know that no peer is still reading
   the previous contents of its buffer (the push overwrites them): synchronise the consumers' streams and pass a
   barrier first, or alternate between two gathered buffers (sharded.py: allgather_peer does the former). */
#define FCAMD_GATHER_PULL 1
FCAMD_API int fcamd_allgather_direct(fcamd_context* ctx, int world, int rank, void* const* gathered, const int* devices,
                                     size_t slot_bytes, size_t offset_bytes, size_t bytes, int flags);
FCAMD_API int fcamd_allgather_direct_wait(fcamd_context* ctx, int host_sync);

/* ---- one process, several GPUs: the single assembler's host entry (SURVEY 8b last row, 8e) ------------- */
/* north_star's single-process mode: ONE dolfinx process assembles, its arrays are host memory (views of
   Function.x.array, solver/_lawonsubmesh.py:87-94), several GPUs evaluate.  A gather over xGMI would end in HBM, one
   PCIe link away from the assembler; here the gather never happens: a fcamd_multi owns one context + one model handle
   per device and one worker thread per device; a call cuts [0, n) with fcamd_shard_bounds (world = the devices used)
   and every device runs the host entry of the single-GPU ABI -- fcamd_evaluate_host resp. fcamd_evaluate_resident -- on
   ITS slice of the caller's arrays, concurrently, over its own PCIe link, results straight into the assembler's
   arrays.  The caller's arrays are page-locked once per call for all devices (ranges registered with
   fcamd_multi_register_host_buffer: never); nothing is copied between devices; history that is resident
   (fcamd_multi_state) never leaves its device.  Results are bit-identical to the single-GPU entries (every point is
   independent, the slices start on tile boundaries).  `devices` may name a device more than once (two contexts on one
   GPU: how the path is tested on a one-GPU box).  Calls of fewer than FCAMD_MULTI_MIN_POINTS points per device use
   fewer devices (launch latency, not PCIe, bounds them).  Thread-compatible like a context: one call at a time. */
typedef struct fcamd_multi fcamd_multi;
#define FCAMD_MULTI_MAX_DEVICES 64
#define FCAMD_MULTI_MIN_POINTS 8192
FCAMD_API int fcamd_multi_create(const int* devices, int n_devices, int model_id, int constraint, const double* params,
                                 int n_params, fcamd_multi** out);
FCAMD_API int fcamd_multi_destroy(fcamd_multi* mg);  /* also destroys the fcamd_multi_state objects still alive on it */
/* How a call over n points is spread: the number of devices it uses and the slice [lo, hi) of device slot k in that
   call (every output nullable; k is ignored when lo and hi are NULL). */
FCAMD_API int fcamd_multi_plan(const fcamd_multi* mg, int64_t n, int k, int* n_used, int64_t* lo, int64_t* hi);
/* fcamd_evaluate_host over all devices: same arguments, same status codes (the first failing slice's), `stats` =
   the sums over the slices.  In place on the caller's host arrays. */
FCAMD_API int fcamd_multi_evaluate_host(fcamd_multi* mg, double t, double del_t, int64_t n, const double* grad_del_u,
                                        double* stress, double* tangent, double* const* history, int n_hist,
                                        fcamd_stats* stats);
/* Page-lock a caller range once for ALL devices of the handle (fcamd_register_host_buffer for every context, one
   lock): calls on it skip the per-call page lock.  bytes = 0 UNREGISTERS the range that starts at ptr -- do it before
   freeing the memory. */
FCAMD_API int fcamd_multi_register_host_buffer(fcamd_multi* mg, void* ptr, size_t bytes);
/* Options.  set: an option of every context of the handle (fcamd_context_set_option), or the handle's own "min_points"
   (default FCAMD_MULTI_MIN_POINTS: a call over n points uses at most n / min_points devices).  get (read-only):
   "n_devices"; "last_host_mode" (FCAMD_HOST_* flags of the last call, OR-ed over the devices used), "last_n_used"; any other name: the
   option as the FIRST device's context holds it (fcamd_context_get_option).  A handle's contexts share the host's CPUs: their automatic
   "host_tangent_threads" (fcamd.h) is the process's usable CPUs divided by the number of devices, at most 16 each. */
FCAMD_API int fcamd_multi_set_option(fcamd_multi* mg, const char* name, long long value);
FCAMD_API int fcamd_multi_get_option(const fcamd_multi* mg, const char* name, long long* value);

/* Device-resident increment state over several GPUs (SURVEY 8f-1 for the single assembler): device slot k keeps the
   committed and the trial copy of stress and history of ITS slice of the n points (fcamd_shard_bounds(n, devices, k))
   in its own HBM -- allocated here, owned by the state.  Per Newton iteration only the gradient goes up and stress /
   tangent come down, over every device's own PCIe link at once:
     _set       committed state <- host arrays (initial conditions, restart; NULL stress / history: zeros); the trial
                history is set equal to it and the sparse-history masks are cleared;
     _evaluate  trial <- law(committed, grad_del_u_host), stress_host / tangent_host (nullable) receive the trial stress
                and tangent: fcamd_evaluate_resident on every slice.  `flags`: FCAMD_EVAL_SPARSE_TANGENT as there
                (the CALLER knows whether tangent_host still holds the previous call's tangent); the sparse trial-history
                protocol is always on for the plasticity laws, FCAMD_EVAL_SPLIT_HISTORY is a property of the state
                (`flags` of _create; the laws with 7-double history rows);
     _commit    trial becomes committed (pointer swap per device; solver/_solver.py:149-159).  Refused
                (FCAMD_ERR_NONCONVERGED / _DOMAIN / _BAD_ARG) when the last evaluate failed or none happened;
     _get       host arrays <- committed (trial = 0) or trial (trial = 1) state, in the law's reference layout.
   History arrays at the interface are always the law's history_dim fields (7-double rows for the comfe-rs laws). */
typedef struct fcamd_multi_state fcamd_multi_state;
FCAMD_API int fcamd_multi_state_create(fcamd_multi* mg, int64_t n, int flags, fcamd_multi_state** out);
FCAMD_API int fcamd_multi_state_destroy(fcamd_multi_state* st);
FCAMD_API int fcamd_multi_state_set(fcamd_multi_state* st, const double* stress_host, const double* const* history_host, int n_hist);
FCAMD_API int fcamd_multi_state_get(fcamd_multi_state* st, int trial, double* stress_host, double* const* history_host, int n_hist);
FCAMD_API int fcamd_multi_state_evaluate(fcamd_multi_state* st, double t, double del_t, const double* grad_del_u_host,
                                         double* stress_host, double* tangent_host, int flags, fcamd_stats* stats);
FCAMD_API int fcamd_multi_state_commit(fcamd_multi_state* st);


/* ---- shorthands (static inline: no symbols) ---- */
FCAMD_INLINE int fcamd_shard_slot_points(int64_t n, int world, int64_t* per_rank) {
    int64_t lo, hi;
    return fcamd_shard_bounds(n, world, 0, &lo, &hi, per_rank);
}
FCAMD_INLINE int fcamd_ipc_alloc(fcamd_context* ctx, size_t bytes, void** device_ptr) {
    return fcamd_device_alloc_set(ctx, 1, &bytes, 0, FCAMD_ALLOC_IPC, device_ptr);
}
FCAMD_INLINE int fcamd_ipc_free(fcamd_context* ctx, void* device_ptr) { return fcamd_device_free(ctx, device_ptr); }
FCAMD_INLINE int fcamd_multi_unregister_host_buffer(fcamd_multi* mg, void* ptr) { return fcamd_multi_register_host_buffer(mg, ptr, 0); }
FCAMD_INLINE int fcamd_multi_device_count(const fcamd_multi* mg, int* n_devices) {
    long long v = 0;
    const int st = fcamd_multi_get_option(mg, "n_devices", &v);
    if (st != FCAMD_OK) return st;
    if (!n_devices) return FCAMD_ERR_BAD_ARG;
    *n_devices = (int)v;
    return FCAMD_OK;
}
FCAMD_INLINE int fcamd_multi_bounds(const fcamd_multi* mg, int64_t n, int k, int64_t* lo, int64_t* hi) {
    return (lo && hi) ? fcamd_multi_plan(mg, n, k, (int*)0, lo, hi) : FCAMD_ERR_BAD_ARG;
}
FCAMD_INLINE int fcamd_multi_last_host_mode(const fcamd_multi* mg, int* mode, int* n_used) {
    long long m = 0, u = 0;
    int st = fcamd_multi_get_option(mg, "last_host_mode", &m);
    if (st == FCAMD_OK) st = fcamd_multi_get_option(mg, "last_n_used", &u);
    if (st != FCAMD_OK) return st;
    if (mode) *mode = (int)m;
    if (n_used) *n_used = (int)u;
    return FCAMD_OK;
}

#ifdef __cplusplus
}
#endif

#endif /* FCAMD_MULTI_H */
