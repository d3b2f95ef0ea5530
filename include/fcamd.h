/*
 * fcamd.h -- C ABI of the MI355X (gfx950) quadrature-point constitutive-update engine.
 *
 * Drop-in boundary for the hot path of BAMresearch/fenics-constitutive:
 *     IncrSmallStrainModel.evaluate(t, del_t, grad_del_u, stress, tangent, history)
 *         reference: src/fenics_constitutive/models/interfaces.py:82-101
 * The entry points below are what a native backend for that interface binds, in the
 * same position as the reference's PyO3 layer (bindings/src/lib.rs:45-152, class
 * new()/evaluate()/history_dim/constraint) and its raw-pointer precedent
 * (comfe-rs/src/linear_elasticity.rs:77-96, examples/elasticity_cpp/src/main.cpp:35-50).
 *
 * Conventions (all taken from the reference):
 *   - every array is flat, C-contiguous IEEE float64, point-major AoS:
 *       grad_del_u[9*i + 3*r + c]   (row-major 3x3 of nabla_grad(u - u_prev))
 *       stress[6*i + k]             Mandel order xx,yy,zz,xy,xz,yz
 *       tangent[36*i + 6*r + c]
 *       history[k][dim_k*i + j]     one array per history field
 *   - stress / tangent / history are overwritten (in-place semantics of evaluate()).
 *   - no C++ types, no exceptions: every function returns an fcamd_status.
 *   - a context is bound to one device and one stream; use one context per thread.
 *
 * Plain pointers and sizes only; nothing from torch or numpy appears here.
 *
 * Version 0.4 (round 4): the boundary is THREE evaluate entries -- fcamd_evaluate_host (the ndarray call),
 * fcamd_evaluate_device_ex (device arrays, every option in one argument struct), fcamd_evaluate_resident (host
 * assembler on a device-resident state) --, plus (round 5) fcamd_evaluate_batch, the laws of one form() in one call: 46 exported
 * symbols in all (0.3: 69) -- 26 here, the 20 of the multi-GPU forms in fcamd_multi.h (round 6).  The narrower forms of 0.3
 * (fcamd_evaluate_device, _from, _from_sparse, _indexed, _wrapped, the single-value getters and setters) are
 * `static inline` shorthands at the end of this header: same names, same arguments, no symbols.
 */
#ifndef FCAMD_H
#define FCAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Every EXPORTED entry point of libfcamd.so carries FCAMD_API. */
#ifndef FCAMD_API
#define FCAMD_API
#endif

#define FCAMD_VERSION_MAJOR 0
#define FCAMD_VERSION_MINOR 4

/* ---- status codes (mapped to Python exceptions by the ctypes shim) -------- */
typedef enum fcamd_status {
    FCAMD_OK = 0,
    FCAMD_ERR_SIZE = 1,          /* inconsistent array sizes -> AssertionError
                                    (linear_elasticity_model.py:36-40; interfaces.rs:402-415) */
    FCAMD_ERR_NULL_HISTORY = 2,  /* history missing -> ValueError (spring_maxwell_model.py:63-65) */
    FCAMD_ERR_DEL_T = 3,         /* del_t <= 0 -> AssertionError (spring_kelvin_model.py:72) */
    FCAMD_ERR_NONCONVERGED = 4,  /* Newton > 100 iterations -> RuntimeError
                                    (mises_plasticity_isotropic_hardening.py:141-143) */
    FCAMD_ERR_HIP = 5,           /* HIP runtime error (see fcamd_last_error) */
    FCAMD_ERR_BAD_ARG = 6,       /* null pointer, unknown model id, bad parameter count -> ValueError */
    FCAMD_ERR_ALIGN = 7,         /* device pointer not 16-byte aligned */
    FCAMD_ERR_UNSUPPORTED = 8,   /* constraint / layout not implemented -> NotImplementedError */
    FCAMD_ERR_DOMAIN = 9         /* a point left the law's domain: "non-differentiable tip of Drucker-Prager
                                    surface reached" (assert!, drucker_prager_classic.rs:82) -> RuntimeError */
} fcamd_status;

/* ---- constitutive laws on the hot path ---------------------- */
typedef enum fcamd_model_id {
    /* LinearElasticityModel, FULL -- models/linear_elasticity_model.py:26-45.
       params[2]: E, nu (the tangent D is built on the host as get_elastic_tangent does,
       utils.py:25-51). */
    FCAMD_LINEAR_ELASTICITY = 1,
    /* VonMises3D -- models/mises_plasticity_isotropic_hardening.py:57-175.
       params[5]: p_ka, p_mu, p_y0, p_y00, p_w.  history: eps_n(6), alpha(1). */
    FCAMD_VON_MISES_3D = 2,
    /* SpringMaxwellModel, FULL -- models/spring_maxwell_model.py:40-88.
       params[4]: E0, E1, tau, nu.  history: strain_visco(6), strain(6). */
    FCAMD_SPRING_MAXWELL = 3,
    /* SpringKelvinModel, FULL -- models/spring_kelvin_model.py:43-88. params/history as Maxwell. */
    FCAMD_SPRING_KELVIN = 4,
    /* comfe-rs LinearElasticity3D -- comfe-rs/src/linear_elasticity.rs:42-75. params[2]: mu, kappa. */
    FCAMD_COMFE_LINEAR_ELASTICITY = 5,
    /* comfe-rs MisesPlasticity3D (linear hardening) -- comfe-rs/src/mises_plasticity.rs:58-126.
       params[4]: mu, kappa, y_0, h.  history: "history"(7) = [alpha, plastic_strain(6)]. */
    FCAMD_COMFE_MISES_PLASTICITY = 6,
    /* comfe-rs IsotropicPlasticityModel3D<DruckerPrager3D> -- general return mapping
       comfe-rs/src/plasticity/general.rs:105-266 with drucker_prager_classic.rs:62-116.
       params[5]: mu, kappa, a, b, b_flow.  history: "history"(7) = [alpha, plastic_strain(6)]. */
    FCAMD_COMFE_DRUCKER_PRAGER = 7,
    /* ... <DruckerPragerHyperbolic3D> -- drucker_prager_hyperbolic.rs:64-114.
       params[6]: mu, kappa, a, b, d, b_flow.  history as above. */
    FCAMD_COMFE_DRUCKER_PRAGER_HYPERBOLIC = 8
} fcamd_model_id;

/* StressStrainConstraint values -- models/interfaces.py:14-28.  Linear elasticity and the two SLS
   laws have kernels for all five (array widths per point then follow stress_strain_dim /
   geometric_dim: 1/1, 4/4, 6/9); the plasticity laws are FULL only, as in the reference. */
typedef enum fcamd_constraint {
    FCAMD_UNIAXIAL_STRAIN = 1,
    FCAMD_UNIAXIAL_STRESS = 2,
    FCAMD_PLANE_STRAIN = 3,
    FCAMD_PLANE_STRESS = 4,
    FCAMD_FULL = 5
} fcamd_constraint;

#define FCAMD_MAX_HISTORY 2

typedef struct fcamd_context fcamd_context;
typedef struct fcamd_model fcamd_model;

/* Per-call statistics (device counters, read back after the launch has completed). */
typedef struct fcamd_stats {
    uint64_t n_nonconverged; /* points whose Newton iteration exceeded 100 steps */
    uint64_t n_plastic;      /* points that took the plastic branch */
    uint64_t n_newton_iters; /* total Newton iterations over all plastic points */
    uint64_t n_domain;       /* points that left the law's domain (Drucker-Prager tip) */
    double kernel_ms;        /* filled by fcamd_model_last_stats only: time of the model's last entry with the context
                                option "timing" on (HIP events around the kernel(s) of a device entry; wall-clock of a
                                synchronous host entry, copies included); -1 when it was not timed */
} fcamd_stats;

/* ---- lifecycle --------------------------------------------------------------- */

/* Create a context on HIP device `device`.  `stream` is a hipStream_t to launch on
   (e.g. torch's current stream), or NULL to let the context own a private stream. */
FCAMD_API int fcamd_context_create(int device, void* stream, fcamd_context** out);
FCAMD_API int fcamd_context_destroy(fcamd_context* ctx);
/* Re-bind the launch stream (borrowed; not destroyed with the context). */
FCAMD_API int fcamd_context_set_stream(fcamd_context* ctx, void* stream);
FCAMD_API int fcamd_context_synchronize(fcamd_context* ctx);

/* Model handle = law id + constraint + host-precomputed constants.
   Replaces the PyO3 constructor `Py*::new(parameters)` (bindings/src/lib.rs:60-75).
   `params` holds the law's parameters in the order documented at fcamd_model_id. */
FCAMD_API int fcamd_model_create(fcamd_context* ctx, int model_id, int constraint,
                                 const double* params, int n_params, fcamd_model** out);
FCAMD_API int fcamd_model_destroy(fcamd_model* model);

/* The getters of the reference's native model classes in one call (bindings/src/lib.rs:131-148: `history_dim`,
   `constraint`, `stress_strain_dim`, `geometric_dim`; models/interfaces.py:103-143): the StressStrainConstraint value,
   the array widths per point, and the law's history fields (name and per-point dimension, in the order every entry
   takes the history pointers).  The names point into the library (valid for its lifetime). */
typedef struct fcamd_model_info {
    int model_id;
    int constraint; /* fcamd_constraint = StressStrainConstraint value */
    int stress_strain_dim;
    int geometric_dim;
    int n_history;
    const char* history_name[FCAMD_MAX_HISTORY];
    int history_dim[FCAMD_MAX_HISTORY];
} fcamd_model_info;
FCAMD_API int fcamd_model_get_info(const fcamd_model* model, fcamd_model_info* info);

/* ---- the hot path ------------------------------------------------------------- */

/* Host evaluate: the ndarray entry.  Pointers are host arrays laid out as above; results are
   written back in place.  Two data paths, chosen per call:
     zero copy -- every array of the call lies inside ranges page-locked with
                  fcamd_register_host_buffer (and is 16-byte aligned): ONE kernel launch runs
                  directly on the caller's arrays; the GPU reads the inputs and writes the
                  results over PCIe itself, both directions at once, no staging buffers;
     pageable  -- arrays the caller did not register are page-locked for the duration of the call
                  (FCAMD_HOST_TEMP_LOCK) and treated the same way; calls that move at most
                  "bounce_max" bytes (256 KiB), and arrays that cannot be locked, go through the
                  context's own page-locked scratch with CPU copies instead (FCAMD_HOST_BOUNCE).
                  Pageable caller memory is never handed to the HIP runtime's copy path (its cache
                  of on-the-fly page locks goes stale when memory is freed and allocated again);
     staged    -- option "zero_copy" = 0, or an array off the 16-byte grid: chunk by chunk through
                  device buffers (four chunk slots on four streams: H2D / kernel / D2H of one chunk
                  overlap the others'), DMA from / into the page-locked arrays.
   Both produce bit-identical results.  Synchronous.  Validates like the reference and returns
   the matching status; on FCAMD_ERR_NONCONVERGED the outputs hold the values the kernel
   produced (the reference raises mid-loop). `stats` may be NULL.  `tangent` may be NULL (the Rust
   entry allows it, comfe-rs/src/interfaces.rs:383-394). */
FCAMD_API int fcamd_evaluate_host(fcamd_model* model, double t, double del_t, int64_t n,
                                  const double* grad_del_u, double* stress, double* tangent,
                                  double* const* history, int n_hist, fcamd_stats* stats);

/* Device evaluate (roofline path): all pointers are device pointers, 16-byte aligned; asynchronous on the
   context's stream.  One argument struct carries every form of the call:
     in place        stress_prev == stress, history_prev == history: the reference's contract;
     out of place    reads the committed state (stress_prev, history_prev), writes the trial state (stress,
                     history): fuses the copies the reference performs before every call
                     (solver/_lawonsubmesh.py:58-61 stress_local <- stress.previous; solver/_history.py:64-79
                     history_1 <- history_0);
     parent_rows     submesh-indexed form (multi-material problems, FULL laws): the gather of the committed stress
                     and the scatter of stress and tangent that the reference performs around evaluate
                     (solver/_lawonsubmesh.py:58-70 with SubSpaceMap, solver/maps.py:82-123) are folded into the
                     kernel's addressing.  `grad_del_u` and the history arrays are local to the n points of this
                     law (as in the reference); stress_prev, stress (stress2) and tangent are the PARENT arrays and
                     point i uses their row parent_rows[i] (int32 device array, every row at most once);
     history_mask    sparse trial-history protocol for device-resident Newton loops (plasticity laws).  Contract: on
                     entry the trial history arrays equal the committed ones except at the points whose bit is set
                     in history_mask (one uint64 per 64-point tile, bit l = point 64*tile + l; all zero initially).
                     On return the trial history is exactly what the plain out-of-place call would have written --
                     but only plastic points (new value) and points that were plastic at the previous call and are
                     elastic now (restored to the committed value) are touched, and history_mask holds the new plastic
                     set.  Elastic points cost no history traffic.  A commit by swapping the committed and trial
                     pointers keeps the contract (the mask then marks the points where the new trial array is stale);
     wrapper_constraint, stress_3d
                     fused form of the reference's 3D -> 1D/2D wrappers around LinearElasticityModel and the
                     plasticity laws (UniaxialStrainFrom3D / PlaneStrainFrom3D, models/utils.py:211-412):
                     grad_del_u, stress, tangent are the LOW-dimensional arrays (1 / 1 / 1 doubles per point for
                     FCAMD_UNIAXIAL_STRAIN, 4 / 4 / 16 for FCAMD_PLANE_STRAIN), stress_3d (6 n) is the wrapper's
                     cached 3-D stress whose unmapped components persist from call to call (utils.py:253-266;
                     zero-initialised by the caller), the history is the 3-D law's, everything in place
                     (stress_prev == stress, history_prev == history).  One kernel replaces map -> evaluate -> map;
                     no 3-D gradient or tangent array exists.  Other laws: FCAMD_ERR_UNSUPPORTED (use
                     fcamd_convert_device around the plain call);
     flags           FCAMD_EVAL_* below. */
typedef struct fcamd_eval_args {
    const double* grad_del_u;
    const double* stress_prev;
    double* stress;
    double* tangent;                    /* nullable */
    const double* const* history_prev;
    double* const* history;
    int n_hist;
    const int32_t* parent_rows;         /* nullable */
    uint64_t* history_mask;             /* nullable */
    int flags;                          /* FCAMD_EVAL_* */
    double* stress2;                    /* nullable: second destination of the stress rows, addressed like
                                           `stress` (the host assembler's page-locked array, see
                                           fcamd_host_device_pointer) */
    uint64_t* counters;                 /* nullable: caller-owned device counters of this launch instead of the
                                           model's own (FCAMD_COUNTER_WORDS words, reset by the call) -- every
                                           resident state keeps its own, so that states sharing one model
                                           handle cannot read each other's non-convergence */
    const uint64_t* packed_mask_prev;   /* FCAMD_EVAL_PACKED_HISTORY: the EVER mask of the committed plastic-strain array, */
    uint64_t* packed_mask;              /* ... of the trial array (written); one word per 64-point tile each */
    int wrapper_constraint;             /* 0, or FCAMD_UNIAXIAL_STRAIN / FCAMD_PLANE_STRAIN: the fused 3D wrapper form */
    double* stress_3d;                  /* the wrapper's cached 3-D stress (6 n), with wrapper_constraint */
} fcamd_eval_args;
/* Layout of a counter buffer: FCAMD_COUNTER_SLOTS slots of 4 words {non-converged points, plastic
   points, Newton iterations, points outside the law's domain}; the totals (fcamd_stats) are the sums
   over the slots (waves add to slot = workgroup % slots, which keeps the atomics off one address). */
#define FCAMD_COUNTER_SLOTS 64
#define FCAMD_COUNTER_WORDS (4 * FCAMD_COUNTER_SLOTS)
/* Sparse-tangent protocol (plasticity laws, needs history_mask and a tangent array): the caller owns
   `tangent` across evaluates and it holds the tangent written by the PREVIOUS evaluate with the same
   history_mask (the first one without this flag).  The tangent of an elastic point is one constant for
   all points and calls, so only the rows of points that are plastic now or were plastic at the previous
   evaluate are written; rows of points that stay elastic -- 288 of their 464 bytes -- are not touched.
   Same array contents as without the flag (tests/test_gpu_resident.py). */
#define FCAMD_EVAL_SPARSE_TANGENT 1
/* (2 was FCAMD_EVAL_DELTA_HISTORY in 0.3: increments in the trial array plus a commit kernel.  Removed in 0.4 --
   FCAMD_EVAL_PACKED_HISTORY gives the same contiguous accesses with a pointer-swap commit.) */
/* Split history (the laws whose reference history is ONE [scalar, eps_p(6)] row of 7 doubles per point: comfe-rs
   MisesPlasticity3D -- alpha --, DruckerPrager3D / DruckerPragerHyperbolic3D -- the hardening variable;
   comfe-rs/src/plasticity/mises_plasticity.rs:58-126, general.rs:105-266).  eps_p only accumulates, the stress update
   never reads it back, but inside the 7-double row every point pays 56 bytes of history reads for the scalar.  With this
   flag the history of the call is TWO arrays, history[0] = the scalars (n doubles), history[1] = the eps_p rows (6 n
   doubles), n_hist = 2: elastic points then read 8 bytes (Mises) of history and write none.  A layout for
   device-resident states (ResidentState keeps it and assembles the reference's rows on demand), not of the interface
   arrays; FULL 3-D only. */
#define FCAMD_EVAL_SPLIT_HISTORY 4
/* Packed plastic-strain history (with history_mask; VonMises3D: history[0] = eps_n; the comfe-rs plasticity laws with
   FCAMD_EVAL_SPLIT_HISTORY: history[1] = the eps_p rows; with parent_rows: VonMises3D only -- the history is local to the law's n points either way).  The plastic-strain array only accumulates
   (models/mises_plasticity_isotropic_hardening.py:161, comfe-rs/src/mises_plasticity.rs:112, plasticity/general.rs:243)
   and is +0.0 wherever a point has never been plastic, so a device-resident state keeps BOTH its copies packed per
   64-point tile: the rows of the points whose row is not all +0.0 -- the tile's EVER mask, `packed_mask_prev[tile]` for the
   committed copy, `packed_mask[tile]` for the trial copy -- lie at the head of the tile's slot, the k-th set bit (ascending
   point order) owning doubles [6 (64 tile + k), 6 (64 tile + k) + 6); the other rows of the slot are undefined.  A touched
   tile (sparse protocol: a point plastic now or at the previous evaluate) reads the committed run and writes the trial run
   (ever_trial = ever_committed | plastic now) as one contiguous stream each; untouched tiles keep trial == committed (run
   and mask word), so the commit is still a swap of pointers -- arrays and mask arrays.  Same values, bit for bit, as the
   unpacked sparse protocol; the scalar history keeps its layout.  ResidentState packs / unpacks at set_state / history. */
#define FCAMD_EVAL_PACKED_HISTORY 8
FCAMD_API int fcamd_evaluate_device_ex(fcamd_model* model, double t, double del_t, int64_t n,
                                       const fcamd_eval_args* args);

/* The laws of ONE form() in one call.  The reference calls its laws back to back, one LawOnSubMesh.evaluate per material
   (solver/_solver.py:143-144); here `count` device calls -- models[k] over n[k] points with args[k], each exactly as
   fcamd_evaluate_device_ex takes it (any form but the fused wrapper one), all of one context, one t / del_t -- are checked
   first (if one is refused nothing is launched and its status is returned), then enqueued on the context's stream from this
   one call.  Laws that fill the device on their own (4096 points per compute unit and more) keep their own launches; the
   others -- as separate launches three dispatches each: counters, main kernel, ragged tile -- leave as ONE launch of a batch
   kernel that reads every law's arguments from a table in device memory.  The table is uploaded only when it changes, so the
   Newton iterations of an increment (same arrays, new gradient values) cost two dispatches whatever the number of laws.
   Precondition (the reference's, too): the laws write disjoint rows of the arrays they share.  Results are bit for bit those
   of the same calls made one by one.  Context option "batch_kernel" (FCAMD_BATCH_KERNEL, 1): 0 = one launch per law; with the
   context option "timing" on, the laws are launched (and timed) one by one. */
FCAMD_API int fcamd_evaluate_batch(int count, fcamd_model* const* models, const int64_t* n, const fcamd_eval_args* args,
                                   double t, double del_t);

/* Resident-state evaluate for a host assembler (replaces the per-iteration copies of
   solver/_lawonsubmesh.py:58-61,84-95 and solver/_history.py:64-79): the committed and trial
   copies of stress and history are DEVICE arrays of n points (committed is read, trial is written), while what must
   cross PCIe in every Newton iteration stays on the host: the gradient (gd2*n) is uploaded chunk by chunk, each chunk is
   evaluated on the device-resident state, and the chunk's trial stress and tangent are
   downloaded into `stress_host` (sd*n) / `tangent_host` (sd*sd*n) while the next chunks are in
   flight (either may be NULL).  The tangent never exists as an n-sized device array.  72 B/pt
   up and 336 B/pt down instead of 176 + 392, and no host-side state copies.
   `state` describes the device-resident arrays exactly as for fcamd_evaluate_device_ex -- stress_prev / stress,
   history_prev / history, n_hist, history_mask, flags, packed_mask_prev / packed_mask -- with ONE difference:
   `state->grad_del_u` is the HOST gradient array; tangent, parent_rows, stress2, wrapper_constraint must be NULL / 0 and
   `counters` is ignored (the call is synchronous and reports through `stats`).
   The host arrays are handled as in fcamd_evaluate_host: ranges registered with
   fcamd_register_host_buffer as they are, pageable arrays page-locked for the duration of the call
   (passes that move at most "bounce_max" bytes: through the context's page-locked scratch).  A
   page-locked gradient / `tangent_host` is read / written by the kernel itself (zero copy)
   instead of passing through the chunk buffers; when all host arrays of the call are page-locked
   (3-D laws) the whole pass is one launch that writes the stress both to the device-resident trial
   array and to `stress_host`.
   FCAMD_EVAL_SPARSE_TANGENT applies the sparse-tangent protocol to `tangent_host` when the kernel writes it directly:
   only the rows of plastic / formerly plastic points cross PCIe (the caller's array must still hold the previous
   call's tangent); ignored on the chunked and scratch paths, which write every row.  Synchronous; waits for work queued
   on the context stream before touching the state arrays.  Status and `stats` as fcamd_evaluate_host. */
FCAMD_API int fcamd_evaluate_resident(fcamd_model* model, double t, double del_t, int64_t n, const fcamd_eval_args* state,
                                      double* stress_host, double* tangent_host, fcamd_stats* stats);

/* Mandel strain from displacement gradient, FULL (utils.py:132-151,187-208).
   rust_factor = 0: factor 1/2**0.5 (Python); 1: FRAC_1_SQRT_2 (mandel.rs:147). */
FCAMD_API int fcamd_strain_from_grad_u_device(fcamd_context* ctx, int64_t n, const double* grad_u,
                                              double* strain, int rust_factor);

/* Component maps of the reference's 3D->1D/2D wrappers (models/utils.py:276-297, 362-412):
   strided copies between a low-dimensional AoS array and its 3-D counterpart.  "TO_3D" kinds
   write only the mapped components of dst (the others keep their values, as the reference's
   cached 3-D arrays do). */
typedef enum fcamd_convert_kind {
    FCAMD_GRAD_1D_TO_3D = 1,     /* grad3d[9i+0]        <- grad1d[i]            utils.py:276-279 */
    FCAMD_STRESS_1D_TO_3D = 2,   /* stress3d[6i+0]      <- stress1d[i]          utils.py:281-284 */
    FCAMD_STRESS_3D_TO_1D = 3,   /* stress1d[i]         <- stress3d[6i+0]       utils.py:286-289 */
    FCAMD_TANGENT_3D_TO_1D = 4,  /* tangent1d[i]        <- tangent3d[36i+0]     utils.py:291-294 */
    FCAMD_GRAD_2D_TO_3D = 5,     /* grad3d[9i+{0,1,3,4}] <- grad2d[4i+{0,1,2,3}] utils.py:362-375 */
    FCAMD_STRESS_2D_TO_3D = 6,   /* stress3d[6i+0..3]   <- stress2d[4i+0..3]    utils.py:377-383 */
    FCAMD_STRESS_3D_TO_2D = 7,   /* stress2d[4i+0..3]   <- stress3d[6i+0..3]    utils.py:385-388 */
    FCAMD_TANGENT_3D_TO_2D = 8   /* tangent2d[16i+4r+c] <- tangent3d[36i+6r+c], r,c<4  utils.py:390-412 */
} fcamd_convert_kind;
FCAMD_API int fcamd_convert_device(fcamd_context* ctx, int kind, int64_t n, const double* src, double* dst);

/* Row gather/scatter between a parent quadrature array and a per-material submesh array:
       dst[row_size*dst_idx[r] + k] = src[row_size*src_idx[r] + k],  r < n_rows, k < row_size
   = SubSpaceMap.map_to_parent / map_to_sub (solver/maps.py:82-123:
   "parent_array[self.parent] = sub_array[self.sub]" with rows of 6 (stress) or 36 (tangent)
   doubles).  Index arrays are int32 device arrays (dolfinx dof indices are int32); NULL means
   the identity (IdentityMap, solver/maps.py:29-59). */
FCAMD_API int fcamd_map_rows_device(fcamd_context* ctx, int64_t n_rows, int row_size, const double* src,
                                    const int32_t* src_idx, double* dst, const int32_t* dst_idx);

/* Synchronise the stream and read the counters accumulated by the last fcamd_evaluate_device_ex launch of this model
   (and `kernel_ms`, see fcamd_stats: the counterpart of the reference's Timer("constitutive-law-evaluation") around
   evaluate, solver/_lawonsubmesh.py:86). */
FCAMD_API int fcamd_model_last_stats(fcamd_model* model, fcamd_stats* stats);

/* Caller arrays are stable across Newton iterations (views of Function.x.array,
   solver/_lawonsubmesh.py:87-94): page-lock and map them once.  fcamd_evaluate_host /
   fcamd_evaluate_resident then run their kernels directly on them (zero copy; any sub-range of
   a registered range qualifies), and the staged path DMAs without the runtime's bounce
   buffers.  The caller must unregister a buffer BEFORE freeing it: on this stack a page lock is an
   attribute of the pages, new memory that appears at the address of a freed, still-registered buffer
   does not carry it, and a launch on it ends in a GPU memory fault.  Registering is an optimisation
   only: arrays that are not registered are page-locked for the duration of each call.  A range that ANOTHER context of
   the process has registered already (several GPUs or threads, one array) is entered into this context's registry with
   its own device's view of it; the page lock is shared and reference-counted -- it is released when the LAST of the
   contexts unregisters the range (or is destroyed), in any order; while a range is shared it cannot be pinned AGAIN
   (registering the same address once more re-enters the existing lock), so EVERY context must unregister a buffer before
   it is freed.  Threads: a range
   on which a host entry of another thread is running right now (it holds a call-scoped page lock) is refused with
   FCAMD_ERR_BAD_ARG -- register between calls; unregistering a range while another thread's call uses it is the
   caller's error, like freeing it. */
FCAMD_API int fcamd_register_host_buffer(fcamd_context* ctx, void* ptr, size_t bytes);
FCAMD_API int fcamd_unregister_host_buffer(fcamd_context* ctx, void* ptr);
/* Data path of the last fcamd_evaluate_host / fcamd_evaluate_resident call of a context (context option
   "last_host_mode", read-only): a bit mask of the flags below (0 = everything staged). */
#define FCAMD_HOST_ZERO_COPY_IN 1  /* inputs read by the kernel from the caller's host arrays */
#define FCAMD_HOST_ZERO_COPY_OUT 2 /* results written by the kernel into the caller's host arrays */
#define FCAMD_HOST_TEMP_LOCK 4     /* pageable caller arrays were page-locked for the duration of the call */
#define FCAMD_HOST_BOUNCE 8        /* pageable caller arrays were moved by the CPU through the context's page-locked scratch */
#define FCAMD_HOST_TANGENT_CPU 16  /* the tangent rows were written by the CPU ("host_tangent_threads", below), not sent over the link */
/* Address at which fcamd_evaluate_device_ex can read / write the host range
   [host_ptr, host_ptr + bytes): it must lie inside one range registered with
   fcamd_register_host_buffer and be 16-byte aligned (FCAMD_ERR_BAD_ARG otherwise).  With it a device
   launch can take the host assembler's arrays as operands -- e.g. the PARENT stress / tangent arrays
   of a multi-material problem as `stress2` / `tangent` with `parent_rows`:
   the reference's map_to_parent copies (solver/maps.py:82-101) then happen inside the kernel, over
   PCIe.  The caller synchronises (fcamd_context_synchronize) before the host reads the results. */
FCAMD_API int fcamd_host_device_pointer(fcamd_context* ctx, const void* host_ptr, size_t bytes, void** device_ptr);

/* Copies, ordered after the work queued on the context stream.
     FCAMD_COPY_TO_DEVICE / FCAMD_COPY_TO_HOST   synchronous, between the caller's HOST memory and device memory.  Like the
         host entries they never hand pageable memory to the HIP runtime's copy path (whose cache of on-the-fly page locks is
         keyed by address and goes stale when memory is freed and allocated again -- a GPU memory fault on this stack,
         DESIGN.md 6): up to "bounce_max" bytes go through the context's own page-locked scratch, larger ranges are
         page-locked for the duration of the copy, registered ranges are used as they are.  What the reference does with
         `array[:] = other` between NumPy arrays (solver/_history.py:64-79) is, for a device-resident state, one of these;
     FCAMD_COPY_DEVICE   device to device, asynchronous, with the access pattern of the evaluate kernels: 16 bytes per lane,
         non-temporal loads and stores, one contiguous KiB per wave instruction -- "trial = committed" of a device-resident
         state, and what bench.py reports as the achievable copy rate of the box next to the 8 TB/s peak.
         Both pointers 16-byte aligned. */
#define FCAMD_COPY_TO_DEVICE 1
#define FCAMD_COPY_TO_HOST 2
#define FCAMD_COPY_DEVICE 3
FCAMD_API int fcamd_copy(fcamd_context* ctx, void* dst, const void* src, size_t bytes, int kind);

/* ---- multi-GPU: contiguous shards + all-gather, one process driving several GPUs ---------------------
   The 20 entry points of the multi-GPU forms -- fcamd_shard_bounds / fcamd_gather_chunk_plan, fcamd_ipc_*, fcamd_allgather_direct*,
   fcamd_multi_*, fcamd_multi_state_* -- are declared in fcamd_multi.h (same library).  They have run on ONE GPU only (several contexts
   on one device, two ranks sharing a device): no multi-GPU node was available in any round, so nothing there is measured on the
   hardware it is for.  This header is the measured core: lifecycle, the four evaluate entries, the device helpers of the path, stats,
   host-buffer registration, copies, device memory, options. */

/* ---- device memory -------------------------------------------------------------- */
/* A working set whose physical placement is chosen by the call: `n_arrays` address ranges of bytes[k]
   bytes (rounded up to the granule), backed by physical handles of `granule_bytes` (0 = 2 MiB; rounded up
   to the device's allocation granularity) created either array after array (FCAMD_ALLOC_SEQUENTIAL) or
   interleaved over all arrays in proportion to their sizes (FCAMD_ALLOC_INTERLEAVED) through
   hipMemAddressReserve / hipMemCreate / hipMemMap.  Background: on MI355X the kernel time follows where
   the written arrays live.  FCAMD_ALLOC_IPC: plain hipMalloc blocks that peers can map
   (fcamd_ipc_export; sizes rounded as described there, granule_bytes ignored).  ptrs[k] receives the base of array k;
   each array is released on its own with fcamd_device_free (which synchronises the device). */
#define FCAMD_ALLOC_SEQUENTIAL 0
#define FCAMD_ALLOC_INTERLEAVED 1
#define FCAMD_ALLOC_IPC 2
FCAMD_API int fcamd_device_alloc_set(fcamd_context* ctx, int n_arrays, const size_t* bytes, size_t granule_bytes, int order,
                                     void** ptrs);
FCAMD_API int fcamd_device_free(fcamd_context* ctx, void* ptr);
/* Note: the ADDRESS RANGES of a VMM set are never returned to the runtime (on this stack a range that is reserved again
   and mapped to new handles serves stale data); the physical memory is.  A process that builds and frees many sets
   consumes virtual address space only (47 bits of it are plentiful: 1e4 sets of 64 GB). */

/* ---- tuning / introspection -------------------------------------------------- */
/* Context options (name, value).  Launch / data-path knobs (experiments; the defaults are the measured optimum) and
   their FCAMD_* environment defaults, which are read ONCE, when the context is created:
     "masked_max" (FCAMD_MASKED_MAX, -1 = per law), "batch_kernel" (FCAMD_BATCH_KERNEL, 1: fcamd_evaluate_batch),
     "host_chunk" (FCAMD_HOST_CHUNK, 0 = automatic), "host_slots" (FCAMD_HOST_SLOTS, 4),
     "bounce_max" (FCAMD_BOUNCE_MAX, 256 KiB: host calls up to this size go through the page-locked scratch),
     "zero_copy" (FCAMD_ZERO_COPY, 1), "zero_copy_grad" (FCAMD_ZERO_COPY_GRAD, 1);
   "host_tangent_threads" (FCAMD_HOST_TANGENT_THREADS, -1): the host entries do not send the tangent -- 288 of the 336-392 bytes per
       point that come down the link -- when the host can rebuild it, as the reference does (np.tile(D.flatten(), n),
       linear_elasticity_model.py:45; C = ka xioi + B xpp + C N (x) N, mises_plasticity_isotropic_hardening.py:170-175): for the laws
       with a constant tangent the kernel writes none and this many threads fill the caller's array from the law's table; for
       VonMises3D and the comfe-rs Mises law the kernel sends, for the plastic points, the 8 doubles per point its own tangent writer
       starts from, and every tile's plastic ballot; the threads expand them chunk by chunk behind the kernel, with the kernel's expression in the kernel's operation order: the
       array holds bit for bit what the kernel would have written.  0: the kernel writes the tangent over PCIe (ABI 0.4's path);
       -1: automatic = the CPUs the process may run on (affinity mask, cgroup quota) less one, at most 16 -- and 0 if that is fewer than 6
       (a rank bound to one or two cores cannot expand as fast as the link delivers); reading the option returns the resolved count.
       "host_tangent_min_points" (FCAMD_HOST_TANGENT_MIN, 65536): smaller calls keep the kernel's tangent stores (laws with a constant
       tangent: half of it, 32768 -- their one-shot fill pays off earlier than the parameter pipeline);
       "host_tangent_chunk" (FCAMD_HOST_TANGENT_CHUNK, 0 = automatic: n / 12, 64 Ki .. 1 Mi points, the last chunk cut in halves down to
       64 Ki points): points per chunk of the parameter ring (4 .. 16 slots, at most 256 MiB page-locked; a value: uniform chunks of that size);
       "host_tangent_streams" (FCAMD_HOST_TANGENT_STREAMS, 1): streams the chunk launches alternate between (2 helps chunks of
       256 Ki points and less, 1 is best at the automatic size);
       "last_host_tangent_cpu_us" / "last_host_tangent_threads" (get only): summed busy time and number of the expansion threads in
       the context's last host entry (0: the kernel wrote the tangent).  The Drucker-Prager laws send 12 doubles per plastic point (the five
       coefficients of their isotropic tangent form, the plastic flag, rho s_tr) and the ballots; elastic points get elastic_tangent() itself.
   "grid": the launch grid (number of 256-thread workgroups; 0 = automatic);
   "timing": 1 = every fcamd_evaluate_device_ex is bracketed by HIP events on the context's stream and every host entry
       by a wall clock; fcamd_model_last_stats reports the time (fcamd_stats.kernel_ms) -- the counterpart of the
       reference's Timer("constitutive-law-evaluation") around evaluate (solver/_lawonsubmesh.py:86); default 0;
   "peer_access" (set only): value = a HIP device ordinal whose memory the context's device may access directly from
       now on (one process driving several GPUs: fcamd_allgather_direct);
   "trim" (set only): release the staging buffers the pageable host path keeps between calls (up to ~1.1 GB), the expansion threads and
       the parameter ring of the host tangent;
   "twin_masks" (set only; a MEASUREMENT DEVICE, never a result): value = device address of one 64-bit word per 64-point tile, the plastic
       ballots recorded from a real evaluate (the history_mask array after it); while it is non-zero, the context's VonMises3D launches of
       the packed sparse protocol (FCAMD_EVAL_PACKED_HISTORY, n >= 64 points per CU x 64, no parent_rows) run as their SYNTHETIC TWIN --
       the same loads and stores at the same addresses with the ballots read from the recording and no constitutive arithmetic: what the
       memory system alone takes for the step (bench.py: roofline.mem_floor_ms).  The values such a launch writes mean nothing; 0 = off;
   "last_host_mode" (get only): FCAMD_HOST_* flags of the context's last host entry. */
FCAMD_API int fcamd_context_set_option(fcamd_context* ctx, const char* name, long long value);
FCAMD_API int fcamd_context_get_option(fcamd_context* ctx, const char* name, long long* value);

/* Number of HIP devices visible to the process (0 and FCAMD_OK when there is none). */
FCAMD_API int fcamd_device_count(int* count);
FCAMD_API const char* fcamd_last_error(void);
FCAMD_API int fcamd_version(void);

/* ================================================================================================================
 * Conveniences: `static inline` shorthands of the entries above (the names and arguments of ABI 0.3).  No symbols.
 * SOURCE compatibility only, not behaviour: a shorthand that finds its own argument wrong returns FCAMD_ERR_BAD_ARG without
 * setting fcamd_last_error() (the text then still describes an earlier call); fcamd_model_last_kernel_ms goes through
 * fcamd_model_last_stats, i.e. it downloads the counters and synchronises the context's stream (0.3 read two events).
 * A binary built against the 0.3 header must not be run against this library: fcamd_stats grew by `kernel_ms` and every
 * entry that takes a fcamd_stats* writes the whole struct -- check fcamd_version() (ABI 0.4 = 4) before the first call.
 * ================================================================================================================ */
/* zero initialiser of a struct in both languages (C: {0}; C++: {} -- keeps -Wextra quiet in a C++ caller) */
#ifdef __cplusplus
#define FCAMD_ZERO_INIT {}
#else
#define FCAMD_ZERO_INIT {0}
#endif
#if defined(__GNUC__) || defined(__clang__)
#define FCAMD_INLINE static inline __attribute__((unused))
#else
#define FCAMD_INLINE static inline
#endif

FCAMD_INLINE const char* fcamd_status_string(int status) {
    switch (status) {
        case FCAMD_OK: return "ok";
        case FCAMD_ERR_SIZE: return "Stress, strain, and tangent lengths do not match";
        case FCAMD_ERR_NULL_HISTORY: return "history must not be None";
        case FCAMD_ERR_DEL_T: return "Time step must be defined and positive.";
        case FCAMD_ERR_NONCONVERGED: return "Newton-Raphson method did not converge for plastic multiplier.";
        case FCAMD_ERR_HIP: return "HIP runtime error";
        case FCAMD_ERR_BAD_ARG: return "bad argument";
        case FCAMD_ERR_ALIGN: return "device arrays must be 16-byte aligned";
        case FCAMD_ERR_UNSUPPORTED: return "constraint / layout not implemented";
        case FCAMD_ERR_DOMAIN: return "non-differentiable tip of Drucker-Prager surface reached";
        default: return "unknown status";
    }
}

/* history_dim and the other getters, one value at a time (models/interfaces.py:103-143; bindings/src/lib.rs:131-148) */
FCAMD_INLINE int fcamd_model_history_count(const fcamd_model* model, int* n_fields) {
    fcamd_model_info i;
    const int st = fcamd_model_get_info(model, &i);
    if (st != FCAMD_OK) return st;
    if (!n_fields) return FCAMD_ERR_BAD_ARG;
    *n_fields = i.n_history;
    return FCAMD_OK;
}
FCAMD_INLINE int fcamd_model_history_field(const fcamd_model* model, int k, const char** name, int* dim) {
    fcamd_model_info i;
    const int st = fcamd_model_get_info(model, &i);
    if (st != FCAMD_OK) return st;
    if (!name || !dim || k < 0 || k >= i.n_history) return FCAMD_ERR_BAD_ARG;
    *name = i.history_name[k];
    *dim = i.history_dim[k];
    return FCAMD_OK;
}
FCAMD_INLINE int fcamd_model_constraint(const fcamd_model* model, int* constraint) {
    fcamd_model_info i;
    const int st = fcamd_model_get_info(model, &i);
    if (st != FCAMD_OK) return st;
    if (!constraint) return FCAMD_ERR_BAD_ARG;
    *constraint = i.constraint;
    return FCAMD_OK;
}
FCAMD_INLINE int fcamd_model_dims(const fcamd_model* model, int* stress_strain_dim, int* geometric_dim) {
    fcamd_model_info i;
    const int st = fcamd_model_get_info(model, &i);
    if (st != FCAMD_OK) return st;
    if (!stress_strain_dim || !geometric_dim) return FCAMD_ERR_BAD_ARG;
    *stress_strain_dim = i.stress_strain_dim;
    *geometric_dim = i.geometric_dim;
    return FCAMD_OK;
}

/* the narrower device forms: out of place; in place; sparse trial history; submesh-indexed; fused 3D wrapper */
FCAMD_INLINE int fcamd_evaluate_device_from(fcamd_model* model, double t, double del_t, int64_t n,
                                            const double* grad_del_u, const double* stress_prev,
                                            double* stress, double* tangent,
                                            const double* const* history_prev, double* const* history, int n_hist) {
    fcamd_eval_args x = FCAMD_ZERO_INIT;
    x.grad_del_u = grad_del_u, x.stress_prev = stress_prev, x.stress = stress, x.tangent = tangent;
    x.history_prev = history_prev, x.history = history, x.n_hist = n_hist;
    return fcamd_evaluate_device_ex(model, t, del_t, n, &x);
}
FCAMD_INLINE int fcamd_evaluate_device(fcamd_model* model, double t, double del_t, int64_t n,
                                       const double* grad_del_u, double* stress, double* tangent,
                                       double* const* history, int n_hist) {
    return fcamd_evaluate_device_from(model, t, del_t, n, grad_del_u, stress, stress, tangent,
                                      (const double* const*)history, history, n_hist);
}
FCAMD_INLINE int fcamd_evaluate_device_from_sparse(fcamd_model* model, double t, double del_t, int64_t n,
                                                   const double* grad_del_u, const double* stress_prev,
                                                   double* stress, double* tangent,
                                                   const double* const* history_prev, double* const* history,
                                                   int n_hist, uint64_t* history_mask) {
    fcamd_eval_args x = FCAMD_ZERO_INIT;
    if (n > 0 && !history_mask) return FCAMD_ERR_BAD_ARG;
    x.grad_del_u = grad_del_u, x.stress_prev = stress_prev, x.stress = stress, x.tangent = tangent;
    x.history_prev = history_prev, x.history = history, x.n_hist = n_hist, x.history_mask = history_mask;
    return fcamd_evaluate_device_ex(model, t, del_t, n, &x);
}
FCAMD_INLINE int fcamd_evaluate_device_indexed(fcamd_model* model, double t, double del_t, int64_t n,
                                               const double* grad_del_u, const double* stress_prev_parent,
                                               double* stress_parent, double* tangent_parent,
                                               const int32_t* parent_rows,
                                               const double* const* history_prev, double* const* history, int n_hist) {
    fcamd_eval_args x = FCAMD_ZERO_INIT;
    if (n > 0 && !parent_rows) return FCAMD_ERR_BAD_ARG;
    x.grad_del_u = grad_del_u, x.stress_prev = stress_prev_parent, x.stress = stress_parent, x.tangent = tangent_parent;
    x.history_prev = history_prev, x.history = history, x.n_hist = n_hist, x.parent_rows = parent_rows;
    return fcamd_evaluate_device_ex(model, t, del_t, n, &x);
}
FCAMD_INLINE int fcamd_evaluate_device_wrapped(fcamd_model* model, int wrapper_constraint, double t, double del_t,
                                               int64_t n, const double* grad_lo, double* stress_lo,
                                               double* tangent_lo, double* stress_3d, double* const* history, int n_hist) {
    fcamd_eval_args x = FCAMD_ZERO_INIT;
    x.grad_del_u = grad_lo, x.stress_prev = stress_lo, x.stress = stress_lo, x.tangent = tangent_lo;
    x.history_prev = (const double* const*)history, x.history = history, x.n_hist = n_hist;
    x.wrapper_constraint = wrapper_constraint ? wrapper_constraint : -1, x.stress_3d = stress_3d;
    return fcamd_evaluate_device_ex(model, t, del_t, n, &x);
}

/* copies, one direction per name */
FCAMD_INLINE int fcamd_copy_to_device(fcamd_context* ctx, void* dst_device, const void* src_host, size_t bytes) {
    return fcamd_copy(ctx, dst_device, src_host, bytes, FCAMD_COPY_TO_DEVICE);
}
FCAMD_INLINE int fcamd_copy_to_host(fcamd_context* ctx, void* dst_host, const void* src_device, size_t bytes) {
    return fcamd_copy(ctx, dst_host, src_device, bytes, FCAMD_COPY_TO_HOST);
}
FCAMD_INLINE int fcamd_copy_device(fcamd_context* ctx, void* dst_device, const void* src_device, size_t bytes) {
    return fcamd_copy(ctx, dst_device, src_device, bytes, FCAMD_COPY_DEVICE);
}

/* options by name */
FCAMD_INLINE int fcamd_context_set_grid(fcamd_context* ctx, int n_workgroups) { return fcamd_context_set_option(ctx, "grid", n_workgroups); }
FCAMD_INLINE int fcamd_context_set_timing(fcamd_context* ctx, int enabled) { return fcamd_context_set_option(ctx, "timing", enabled); }
FCAMD_INLINE int fcamd_context_trim(fcamd_context* ctx) { return fcamd_context_set_option(ctx, "trim", 1); }
FCAMD_INLINE int fcamd_enable_peer_access(fcamd_context* ctx, int peer_device) { return fcamd_context_set_option(ctx, "peer_access", peer_device); }
FCAMD_INLINE int fcamd_context_last_host_mode(fcamd_context* ctx, int* mode) {
    long long v = 0;
    const int st = fcamd_context_get_option(ctx, "last_host_mode", &v);
    if (st != FCAMD_OK) return st;
    if (!mode) return FCAMD_ERR_BAD_ARG;
    *mode = (int)v;
    return FCAMD_OK;
}
/* kernel time of the model's last entry (context option "timing"); FCAMD_ERR_BAD_ARG when it was not timed */
FCAMD_INLINE int fcamd_model_last_kernel_ms(fcamd_model* model, float* ms) {
    fcamd_stats s;
    const int st = fcamd_model_last_stats(model, &s);
    if (st != FCAMD_OK) return st;
    if (!ms || s.kernel_ms < 0.0) return FCAMD_ERR_BAD_ARG;
    *ms = (float)s.kernel_ms;
    return FCAMD_OK;
}

#ifdef __cplusplus
}
#endif
#endif /* FCAMD_H */
