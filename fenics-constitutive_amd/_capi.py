"""ctypes binding of ``libfcamd.so`` (C ABI in ``include/fcamd.h``).

This is the only place that touches the shared library.  There is no CPU fallback: when
the library cannot be loaded, or no HIP device is present, every entry raises.
"""

from __future__ import annotations

import ctypes as C
import os
import sys
import threading

import numpy as np

from . import _build

# status codes (include/fcamd.h)
OK, ERR_SIZE, ERR_NULL_HISTORY, ERR_DEL_T, ERR_NONCONVERGED, ERR_HIP, ERR_BAD_ARG, ERR_ALIGN, ERR_UNSUPPORTED, ERR_DOMAIN = range(10)

# model ids (include/fcamd.h)
(LINEAR_ELASTICITY, VON_MISES_3D, SPRING_MAXWELL, SPRING_KELVIN, COMFE_LINEAR_ELASTICITY, COMFE_MISES_PLASTICITY,
 COMFE_DRUCKER_PRAGER, COMFE_DRUCKER_PRAGER_HYPERBOLIC) = range(1, 9)

MAX_HISTORY = 2
COUNTER_SLOTS, COUNTER_WORDS = 64, 256  # FCAMD_COUNTER_SLOTS / FCAMD_COUNTER_WORDS

# fcamd_eval_args.flags / fcamd_evaluate_resident flags (include/fcamd.h)
EVAL_SPARSE_TANGENT = 1
EVAL_SPLIT_HISTORY = 4
EVAL_PACKED_HISTORY = 8

# context option "last_host_mode": FCAMD_HOST_* flags (include/fcamd.h)
HOST_ZERO_COPY_IN, HOST_ZERO_COPY_OUT, HOST_TEMP_LOCK, HOST_BOUNCE = 1, 2, 4, 8
HOST_TANGENT_CPU = 16  # the tangent rows were written by host threads (context option "host_tangent_threads"), not sent over the link

# conversion kinds (include/fcamd.h)
(GRAD_1D_TO_3D, STRESS_1D_TO_3D, STRESS_3D_TO_1D, TANGENT_3D_TO_1D,
 GRAD_2D_TO_3D, STRESS_2D_TO_3D, STRESS_3D_TO_2D, TANGENT_3D_TO_2D) = range(1, 9)

#: every symbol include/fcamd.h (26: the measured core) and include/fcamd_multi.h (20: the multi-GPU forms) export (FCAMD_API; checked by tests/test_host_logic.py::test_library_exports_every_declared_symbol)
SYMBOLS = [
    "fcamd_context_create", "fcamd_context_destroy", "fcamd_context_set_stream", "fcamd_context_synchronize",
    "fcamd_model_create", "fcamd_model_destroy", "fcamd_model_get_info",
    "fcamd_evaluate_host", "fcamd_evaluate_device_ex", "fcamd_evaluate_batch", "fcamd_evaluate_resident",
    "fcamd_strain_from_grad_u_device", "fcamd_convert_device", "fcamd_map_rows_device", "fcamd_model_last_stats",
    "fcamd_register_host_buffer", "fcamd_unregister_host_buffer", "fcamd_host_device_pointer", "fcamd_copy",
    "fcamd_shard_bounds", "fcamd_gather_chunk_plan", "fcamd_ipc_export", "fcamd_ipc_open", "fcamd_ipc_close",
    "fcamd_allgather_direct", "fcamd_allgather_direct_wait",
    "fcamd_multi_create", "fcamd_multi_destroy", "fcamd_multi_plan", "fcamd_multi_evaluate_host",
    "fcamd_multi_register_host_buffer", "fcamd_multi_set_option", "fcamd_multi_get_option",
    "fcamd_multi_state_create", "fcamd_multi_state_destroy", "fcamd_multi_state_set", "fcamd_multi_state_get",
    "fcamd_multi_state_evaluate", "fcamd_multi_state_commit",
    "fcamd_device_alloc_set", "fcamd_device_free", "fcamd_context_set_option", "fcamd_context_get_option",
    "fcamd_device_count", "fcamd_last_error", "fcamd_version",
]

IPC_HANDLE_BYTES = 64
MULTI_MAX_DEVICES, MULTI_MIN_POINTS = 64, 8192  # FCAMD_MULTI_MAX_DEVICES / FCAMD_MULTI_MIN_POINTS
GATHER_PULL = 1
ALLOC_SEQUENTIAL, ALLOC_INTERLEAVED, ALLOC_IPC = 0, 1, 2
COPY_TO_DEVICE, COPY_TO_HOST, COPY_DEVICE = 1, 2, 3


class EvalArgs(C.Structure):
    """``fcamd_eval_args``."""

    _fields_ = [("grad_del_u", C.c_void_p), ("stress_prev", C.c_void_p), ("stress", C.c_void_p),
                ("tangent", C.c_void_p), ("history_prev", C.POINTER(C.c_void_p)), ("history", C.POINTER(C.c_void_p)),
                ("n_hist", C.c_int), ("parent_rows", C.c_void_p), ("history_mask", C.c_void_p), ("flags", C.c_int), ("stress2", C.c_void_p),
                ("counters", C.c_void_p), ("packed_mask_prev", C.c_void_p), ("packed_mask", C.c_void_p),
                ("wrapper_constraint", C.c_int), ("stress_3d", C.c_void_p)]


class ModelInfo(C.Structure):
    """``fcamd_model_info``."""

    _fields_ = [("model_id", C.c_int), ("constraint", C.c_int), ("stress_strain_dim", C.c_int), ("geometric_dim", C.c_int),
                ("n_history", C.c_int), ("history_name", C.c_char_p * MAX_HISTORY), ("history_dim", C.c_int * MAX_HISTORY)]


class Stats(C.Structure):
    _fields_ = [
        ("n_nonconverged", C.c_uint64),
        ("n_plastic", C.c_uint64),
        ("n_newton_iters", C.c_uint64),
        ("n_domain", C.c_uint64),
        ("kernel_ms", C.c_double),
    ]


_lib = None
_lock = threading.Lock()


def library_path() -> str:
    return _build.LIB


def _share_hip_runtime_with_torch() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own ``libamdhip64.so``
    (soname ``libamdhip64.so.7``, the same as the system ROCm one).  If libfcamd.so were loaded
    first it would bind the system runtime, torch would later load its bundled copy, and the
    second runtime to initialise finds no device.  Importing torch first makes the dynamic
    loader satisfy our ``NEEDED libamdhip64.so.7`` with the copy torch already mapped, so
    tensors, streams and our launches share one runtime.  Without torch installed the system
    runtime is used (NumPy host path only)."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    if importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401


def load(build_if_missing: bool = True) -> C.CDLL:
    """Load (building first if needed) libfcamd.so.  Raises if that is impossible."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = os.environ.get("FCAMD_LIBRARY")  # override: A/B of two builds (tools/)
        if not path:
            path = _build.LIB
            # Rebuild when the sources changed since the library was built (content hash, _build._stale):
            # tests and bench.py must never run against stale kernels.  A no-op when the hash matches; on a
            # box without hipcc an up-to-date prebuilt library is used as it is.
            if build_if_missing and (_build._stale() if os.path.isdir(_build.CSRC) else not os.path.exists(path)):
                _build.build_library()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing; run __graft_entry__.build()")
        _share_hip_runtime_with_torch()
        lib = C.CDLL(path)
        dp = C.POINTER(C.c_double)
        vp = C.c_void_p
        lib.fcamd_context_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
        lib.fcamd_context_destroy.argtypes = [vp]
        lib.fcamd_context_set_stream.argtypes = [vp, vp]
        lib.fcamd_context_synchronize.argtypes = [vp]
        lib.fcamd_model_create.argtypes = [vp, C.c_int, C.c_int, dp, C.c_int, C.POINTER(vp)]
        lib.fcamd_model_destroy.argtypes = [vp]
        lib.fcamd_model_get_info.argtypes = [vp, C.POINTER(ModelInfo)]
        i64p = C.POINTER(C.c_int64)
        lib.fcamd_shard_bounds.argtypes = [C.c_int64, C.c_int, C.c_int, i64p, i64p, i64p]
        lib.fcamd_gather_chunk_plan.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_size_t, C.c_int, i64p, i64p]
        lib.fcamd_ipc_export.argtypes = [vp, vp, C.c_char_p, C.POINTER(C.c_size_t)]
        lib.fcamd_ipc_open.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(vp)]
        lib.fcamd_ipc_close.argtypes = [vp, vp, C.c_size_t]
        lib.fcamd_allgather_direct.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(C.c_int), C.c_size_t,
                                               C.c_size_t, C.c_size_t, C.c_int]
        lib.fcamd_allgather_direct_wait.argtypes = [vp, C.c_int]
        lib.fcamd_device_alloc_set.argtypes = [vp, C.c_int, C.POINTER(C.c_size_t), C.c_size_t, C.c_int, C.POINTER(vp)]
        lib.fcamd_device_free.argtypes = [vp, vp]
        lib.fcamd_context_set_option.argtypes = [vp, C.c_char_p, C.c_longlong]
        lib.fcamd_context_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_longlong)]
        lib.fcamd_evaluate_host.argtypes = [vp, C.c_double, C.c_double, C.c_int64, vp, vp, vp, C.POINTER(vp), C.c_int, C.POINTER(Stats)]
        lib.fcamd_evaluate_device_ex.argtypes = [vp, C.c_double, C.c_double, C.c_int64, C.POINTER(EvalArgs)]
        lib.fcamd_evaluate_batch.argtypes = [C.c_int, C.POINTER(vp), C.POINTER(C.c_int64), C.POINTER(EvalArgs), C.c_double, C.c_double]
        lib.fcamd_evaluate_resident.argtypes = [vp, C.c_double, C.c_double, C.c_int64, C.POINTER(EvalArgs), vp, vp, C.POINTER(Stats)]
        lib.fcamd_strain_from_grad_u_device.argtypes = [vp, C.c_int64, vp, vp, C.c_int]
        lib.fcamd_convert_device.argtypes = [vp, C.c_int, C.c_int64, vp, vp]
        lib.fcamd_map_rows_device.argtypes = [vp, C.c_int64, C.c_int, vp, vp, vp, vp]
        lib.fcamd_model_last_stats.argtypes = [vp, C.POINTER(Stats)]
        lib.fcamd_register_host_buffer.argtypes = [vp, vp, C.c_size_t]
        lib.fcamd_unregister_host_buffer.argtypes = [vp, vp]
        lib.fcamd_host_device_pointer.argtypes = [vp, vp, C.c_size_t, C.POINTER(vp)]
        lib.fcamd_copy.argtypes = [vp, vp, vp, C.c_size_t, C.c_int]
        ip = C.POINTER(C.c_int)
        lib.fcamd_multi_create.argtypes = [ip, C.c_int, C.c_int, C.c_int, dp, C.c_int, C.POINTER(vp)]
        lib.fcamd_multi_destroy.argtypes = [vp]
        lib.fcamd_multi_plan.argtypes = [vp, C.c_int64, C.c_int, ip, i64p, i64p]
        lib.fcamd_multi_evaluate_host.argtypes = [vp, C.c_double, C.c_double, C.c_int64, vp, vp, vp, C.POINTER(vp), C.c_int, C.POINTER(Stats)]
        lib.fcamd_multi_register_host_buffer.argtypes = [vp, vp, C.c_size_t]
        lib.fcamd_multi_set_option.argtypes = [vp, C.c_char_p, C.c_longlong]
        lib.fcamd_multi_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_longlong)]
        lib.fcamd_multi_state_create.argtypes = [vp, C.c_int64, C.c_int, C.POINTER(vp)]
        lib.fcamd_multi_state_destroy.argtypes = [vp]
        lib.fcamd_multi_state_set.argtypes = [vp, vp, C.POINTER(vp), C.c_int]
        lib.fcamd_multi_state_get.argtypes = [vp, C.c_int, vp, C.POINTER(vp), C.c_int]
        lib.fcamd_multi_state_evaluate.argtypes = [vp, C.c_double, C.c_double, vp, vp, vp, C.c_int, C.POINTER(Stats)]
        lib.fcamd_multi_state_commit.argtypes = [vp]
        lib.fcamd_device_count.argtypes = [C.POINTER(C.c_int)]
        lib.fcamd_last_error.restype = C.c_char_p
        for name in SYMBOLS:
            f = getattr(lib, name)
            if name != "fcamd_last_error":
                f.restype = C.c_int
        _lib = lib
        return lib


#: the header's fcamd_status_string (a static inline there: no symbol to call)
STATUS_STRINGS = {
    OK: "ok", ERR_SIZE: "Stress, strain, and tangent lengths do not match", ERR_NULL_HISTORY: "history must not be None",
    ERR_DEL_T: "Time step must be defined and positive.",
    ERR_NONCONVERGED: "Newton-Raphson method did not converge for plastic multiplier.", ERR_HIP: "HIP runtime error",
    ERR_BAD_ARG: "bad argument", ERR_ALIGN: "device arrays must be 16-byte aligned",
    ERR_UNSUPPORTED: "constraint / layout not implemented", ERR_DOMAIN: "non-differentiable tip of Drucker-Prager surface reached",
}


def status_string(status: int) -> str:
    return STATUS_STRINGS.get(int(status), "unknown status")


def check(status: int) -> None:
    """Map a status code to the exception type the reference raises in the same situation
    (SURVEY.md 8b "Error conventions")."""
    if status == OK:
        return
    lib = load()
    detail = (lib.fcamd_last_error() or b"").decode() or status_string(status)
    if status in (ERR_SIZE, ERR_DEL_T):
        raise AssertionError(detail)
    if status == ERR_NULL_HISTORY:
        raise ValueError(detail)
    if status in (ERR_NONCONVERGED, ERR_DOMAIN):
        raise RuntimeError(detail)
    if status == ERR_UNSUPPORTED:
        raise NotImplementedError(detail)
    if status in (ERR_BAD_ARG, ERR_ALIGN):
        raise ValueError(detail)
    raise RuntimeError(f"libfcamd: {detail}")


# ---------------------------------------------------------------------------------------
# contexts: one per (device, thread), held in thread-local storage; the launch stream is re-bound per call
# ---------------------------------------------------------------------------------------
class Context:
    def __init__(self, device: int = 0):
        lib = load()
        h = C.c_void_p()
        check(lib.fcamd_context_create(int(device), None, C.byref(h)))
        self.handle = h
        self.device = int(device)
        self._lib = lib

    def set_stream(self, stream_ptr: int | None) -> None:
        check(self._lib.fcamd_context_set_stream(self.handle, C.c_void_p(stream_ptr or 0)))

    def synchronize(self) -> None:
        check(self._lib.fcamd_context_synchronize(self.handle))

    def set_grid(self, n_workgroups: int) -> None:
        self.set_option("grid", int(n_workgroups))

    def set_timing(self, enabled: bool) -> None:
        self.set_option("timing", int(bool(enabled)))

    def set_option(self, name: str, value: int) -> None:
        """Launch / data-path knob (``fcamd_context_set_option``; the FCAMD_* environment variables are only
        the defaults, read once when the context is created)."""
        check(self._lib.fcamd_context_set_option(self.handle, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_longlong()
        check(self._lib.fcamd_context_get_option(self.handle, name.encode(), C.byref(v)))
        return int(v.value)

    def trim(self) -> None:
        """Release the staging buffers of the pageable host path."""
        self.set_option("trim", 1)

    def register_host_buffer(self, arr: np.ndarray) -> None:
        check(self._lib.fcamd_register_host_buffer(self.handle, C.c_void_p(arr.ctypes.data), arr.nbytes))

    def unregister_host_buffer(self, arr: np.ndarray) -> None:
        check(self._lib.fcamd_unregister_host_buffer(self.handle, C.c_void_p(arr.ctypes.data)))

    def device_pointer(self, arr: np.ndarray) -> int:
        """Address at which device launches see the (page-locked, registered) NumPy array; ValueError if it
        is not inside a registered range."""
        d = C.c_void_p()
        check(self._lib.fcamd_host_device_pointer(self.handle, C.c_void_p(arr.ctypes.data), arr.nbytes, C.byref(d)))
        return int(d.value)

    def copy_to_device(self, dst_device_ptr: int, src: "np.ndarray") -> None:
        """``fcamd_copy_to_device``: synchronous, ordered after the context stream, never through the HIP
        runtime's pageable-copy path (include/fcamd.h)."""
        check(self._lib.fcamd_copy(self.handle, C.c_void_p(dst_device_ptr), C.c_void_p(src.ctypes.data), src.nbytes, COPY_TO_DEVICE))

    def copy_device(self, dst_device_ptr: int, src_device_ptr: int, nbytes: int) -> None:
        """``fcamd_copy_device``: asynchronous device-to-device copy on the context stream (non-temporal, 16 B per lane)."""
        check(self._lib.fcamd_copy(self.handle, C.c_void_p(dst_device_ptr), C.c_void_p(src_device_ptr), int(nbytes), COPY_DEVICE))

    def copy_to_host(self, dst: "np.ndarray", src_device_ptr: int) -> None:
        check(self._lib.fcamd_copy(self.handle, C.c_void_p(dst.ctypes.data), C.c_void_p(src_device_ptr), dst.nbytes, COPY_TO_HOST))

    def last_host_mode(self) -> int:
        """Data path of the last host-entry call (HOST_* flags): bit 0 = inputs, bit 1 = results moved by the kernel
        itself (zero copy on page-locked caller arrays), bit 2 = pageable caller arrays were page-locked for the call,
        bit 3 = moved by the CPU through the context's page-locked scratch; 0 = staged through device buffers."""
        return self.get_option("last_host_mode")

    # -- multi-GPU ----------------------------------------------------------------------------
    def ipc_alloc(self, nbytes: int) -> int:
        """``fcamd_device_alloc_set(FCAMD_ALLOC_IPC)``: device buffer whose size hipIpcOpenMemHandle can map (see include/fcamd.h)."""
        size, out = (C.c_size_t * 1)(int(nbytes)), (C.c_void_p * 1)()
        check(self._lib.fcamd_device_alloc_set(self.handle, 1, size, 0, ALLOC_IPC, out))
        return int(out[0])

    def ipc_free(self, device_ptr: int) -> None:
        check(self._lib.fcamd_device_free(self.handle, C.c_void_p(device_ptr)))

    def ipc_export(self, device_ptr: int) -> tuple[bytes, int]:
        """(64-byte handle of the allocation ``device_ptr`` lies in, offset of the pointer inside it)."""
        buf = C.create_string_buffer(IPC_HANDLE_BYTES)
        off = C.c_size_t()
        check(self._lib.fcamd_ipc_export(self.handle, C.c_void_p(device_ptr), buf, C.byref(off)))
        return buf.raw, int(off.value)

    def ipc_open(self, handle: bytes, offset: int) -> int:
        out = C.c_void_p()
        check(self._lib.fcamd_ipc_open(self.handle, handle, int(offset), C.byref(out)))
        return int(out.value)

    def ipc_close(self, device_ptr: int, offset: int) -> None:
        check(self._lib.fcamd_ipc_close(self.handle, C.c_void_p(device_ptr), int(offset)))

    def enable_peer_access(self, peer_device: int) -> None:
        self.set_option("peer_access", int(peer_device))

    def allgather_direct(self, world: int, rank: int, gathered_ptrs, slot_bytes: int, offset_bytes: int = 0,
                         nbytes: int | None = None, devices=None, pull: bool = False) -> None:
        """``fcamd_allgather_direct``: world-1 peer copies on world-1 streams (asynchronous)."""
        arr = (C.c_void_p * world)(*[C.c_void_p(int(p)) for p in gathered_ptrs])
        dev = None if devices is None else (C.c_int * world)(*[int(d) for d in devices])
        check(self._lib.fcamd_allgather_direct(self.handle, int(world), int(rank), arr, dev, int(slot_bytes),
                                               int(offset_bytes), int(slot_bytes - offset_bytes if nbytes is None else nbytes),
                                               GATHER_PULL if pull else 0))

    def allgather_direct_wait(self, host_sync: bool = True) -> None:
        check(self._lib.fcamd_allgather_direct_wait(self.handle, int(bool(host_sync))))

    # -- device memory ------------------------------------------------------------------------
    def alloc_set(self, nbytes: list[int], granule: int = 0, interleaved: bool = True) -> list[int]:
        """``fcamd_device_alloc_set``: base addresses of a working set placed through the VMM API."""
        k = len(nbytes)
        sizes = (C.c_size_t * k)(*[int(b) for b in nbytes])
        out = (C.c_void_p * k)()
        check(self._lib.fcamd_device_alloc_set(self.handle, k, sizes, int(granule),
                                               ALLOC_INTERLEAVED if interleaved else ALLOC_SEQUENTIAL, out))
        return [int(p) for p in out]

    def free(self, ptr: int) -> None:
        check(self._lib.fcamd_device_free(self.handle, C.c_void_p(ptr)))

    def close(self) -> None:
        if self.handle:
            self._lib.fcamd_context_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        # The context of a thread dies with the thread's local storage (get_context) once the model handles
        # and page-locked registrations that refer to it are gone: streams and staging buffers are released
        # instead of accumulating with every short-lived worker thread.
        if sys is None or sys.is_finalizing():  # process exit releases everything; the HIP runtime may already be going down
            return
        try:
            self.close()
        except Exception:
            pass


def shard_bounds(n: int, world: int, rank: int) -> tuple[int, int]:
    """``fcamd_shard_bounds``: [lo, hi) of rank's contiguous, 64-aligned slice of [0, n)."""
    lo, hi = C.c_int64(), C.c_int64()
    check(load().fcamd_shard_bounds(int(n), int(world), int(rank), C.byref(lo), C.byref(hi), None))
    return int(lo.value), int(hi.value)


def shard_slot_points(n: int, world: int) -> int:
    lo, hi, per = C.c_int64(), C.c_int64(), C.c_int64()
    check(load().fcamd_shard_bounds(int(n), int(world), 0, C.byref(lo), C.byref(hi), C.byref(per)))
    return int(per.value)


def gather_chunk_plan(slot_points: int, world: int, values_per_point: int, budget_bytes: int, n_buffers: int = 2) -> tuple[int, int]:
    """``fcamd_gather_chunk_plan``: (chunk_points, n_chunks); AssertionError if the budget holds no tile."""
    c, k = C.c_int64(), C.c_int64()
    check(load().fcamd_gather_chunk_plan(int(slot_points), int(world), int(values_per_point), int(budget_bytes),
                                         int(n_buffers), C.byref(c), C.byref(k)))
    return int(c.value), int(k.value)


def device_count() -> int:
    """visible HIP devices (``fcamd_device_count``; 0 without a GPU)"""
    n = C.c_int()
    return int(n.value) if load().fcamd_device_count(C.byref(n)) == OK else 0


def default_device() -> int:
    """GPU used by the NumPy (host) path of this process: ``FCAMD_DEVICE`` if set; else a device the application has
    selected itself (``torch.cuda.set_device(k)``, k != 0); else the node-local rank of the usual launchers MODULO the
    number of visible GPUs -- dolfinx under MPI runs one rank per GPU, or several ranks per GPU when there are more ranks
    than GPUs; else torch's current device / 0."""
    import sys

    if "FCAMD_DEVICE" in os.environ:
        return int(os.environ["FCAMD_DEVICE"])
    torch = sys.modules.get("torch")
    cur = int(torch.cuda.current_device()) if torch is not None and torch.cuda.is_available() else None
    if cur:  # selected explicitly: the application's choice wins over the launcher's numbering
        return cur
    for var in ("LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "SLURM_LOCALID"):
        if var in os.environ:
            rank = int(os.environ[var])
            count = int(torch.cuda.device_count()) if cur is not None else device_count()
            return rank % count if count > 0 else rank
    return cur or 0


# Thread-local: a context (4 chunk streams, up to ~1.1 GB of staging buffers, the registry of page-locked
# ranges) belongs to the thread that created it and is released when that thread ends -- a dict keyed by
# threading.get_ident() would keep it for ever and hand it to a later thread that recycles the ident.
_tls = threading.local()


class PreparedBatch:
    """the argument arrays of one ``fcamd_evaluate_batch``, kept so that the same call (same arrays, same ``t`` / ``del_t``) can be
    issued again without rebuilding them: ``launch()``"""

    def __init__(self, calls, t, del_t):
        k = len(calls)
        self.count, self.t, self.del_t = k, C.c_double(t), C.c_double(del_t)
        self.ctx = calls[0][0].ctx
        self._lib = calls[0][0]._lib
        self._keep = calls  # the models (handles) and the per-call pointer arrays the structs point into
        self.models = (C.c_void_p * k)(*[m.handle for m, _, _, _ in calls])
        self.ns = (C.c_int64 * k)(*[n for _, n, _, _ in calls])
        self.args = (EvalArgs * k)(*[x for _, _, x, _ in calls])

    def launch(self):
        check(self._lib.fcamd_evaluate_batch(self.count, self.models, self.ns, self.args, self.t, self.del_t))


class LaunchCache:
    """Replays the ``PreparedBatch`` of a group of device calls for as long as the group's signature -- every pointer, ``del_t`` -- is
    the one it was built for (the Newton iterations of an increment: same arrays, new gradient values); otherwise the calls are
    made through ``enqueue()`` inside a ``batched_launches()`` block and kept.  ``t`` is not part of a signature: no law of the
    reference uses it and the C entries ignore it."""

    def __init__(self):
        self._kept = {}

    def run(self, slot, signature, enqueue, stream_ptr):
        if getattr(_tls, "batch", None) is not None:  # inside a caller's own batched_launches(): the calls join that batch
            enqueue()
            return
        signature = (threading.get_ident(), signature)  # (model handles and contexts belong to the thread that made the calls)
        kept = self._kept.get(slot)
        if kept is not None and kept[0] == signature:
            kept[1].ctx.set_stream(stream_ptr)
            kept[1].launch()
            return
        with batched_launches() as b:
            enqueue()
        self._kept[slot] = (signature, b.prepared[0]) if len(b.prepared) == 1 else None

    def clear(self):
        self._kept.clear()


def _flush_recorded() -> None:
    """an evaluate that is NOT recorded (the fused wrapper form, the host and resident entries) is about to run inside a
    ``batched_launches()`` block: the calls recorded so far leave first, so that the block keeps the order the caller wrote"""
    batch = getattr(_tls, "batch", None)
    if batch is not None:
        batch.flush()


class batched_launches:
    """``with batched_launches():`` -- the device calls made inside the block (``Model.evaluate_device_ex``, i.e. every
    ``DeviceLaw.evaluate_from`` / ``evaluate_indexed`` on device tensors) are recorded and leave together as ONE
    ``fcamd_evaluate_batch`` when the block ends: the laws of one ``form()`` (solver/_solver.py:143-144) in one trip through the
    binding, the small ones as ONE launch of the batch kernel (``fcamd_evaluate_batch``).  A call with another context, ``t`` or
    ``del_t`` than the recorded ones flushes what has been recorded first, and so does every evaluate that is not recorded (the
    fused wrapper form, the host and resident entries): the block keeps the order in which the calls were written.  The launches happen when the block ENDS: every array passed to a
    recorded call must stay alive and unchanged until then (a temporary tensor freed inside the block may be handed out again
    before the kernel reads it).  Not re-entrant; per thread."""

    def __init__(self):
        self.calls = []  # (model, n, args, keep-alive)
        self.key = None
        self.prepared = []  # what left: one PreparedBatch per flush (a caller whose arrays do not move may launch it again)

    def __enter__(self):
        assert getattr(_tls, "batch", None) is None, "batched_launches is not re-entrant"
        _tls.batch = self
        return self

    def add(self, model, t, del_t, n, x, keep):
        key = (id(model.ctx), t, del_t)
        if self.key is not None and key != self.key:
            self.flush()
        self.key = key
        self.calls.append((model, n, x, keep))

    def flush(self):
        calls, key, self.calls, self.key = self.calls, self.key, [], None
        if not calls:
            return
        prepared = PreparedBatch(calls, key[1], key[2])
        self.prepared.append(prepared)
        prepared.launch()

    def __exit__(self, exc_type, exc, tb):
        _tls.batch = None
        if exc_type is None:
            self.flush()
        return False


def get_context(device: int = 0) -> Context:
    ctxs = _tls.__dict__.setdefault("contexts", {})
    ctx = ctxs.get(int(device))
    if ctx is None:
        ctx = ctxs[int(device)] = Context(device)
    return ctx


def default_devices():
    """Devices of the single-process multi-GPU host path (``fcamd_multi``): ``FCAMD_DEVICES`` = a comma-separated list
    of device ordinals ("0,1,2,3") or "all"; unset / empty: None (the one-device path on ``default_device()``)."""
    spec = os.environ.get("FCAMD_DEVICES", "").strip()
    if not spec:
        return None
    if spec.lower() == "all":
        import torch

        return list(range(torch.cuda.device_count()))
    return [int(x) for x in spec.split(",") if x.strip()]


def _ptr_array(ptrs):
    if not ptrs:
        return None, 0
    return (C.c_void_p * len(ptrs))(*[C.c_void_p(int(p)) for p in ptrs]), len(ptrs)


class Multi:
    """``fcamd_multi``: one law on several GPUs of this process -- one context, model handle and worker thread per
    device; ``evaluate_host`` is ``fcamd_evaluate_host`` with every device working on its own slice of the caller's
    arrays over its own PCIe link (include/fcamd.h, "one process, several GPUs")."""

    def __init__(self, devices, model_id: int, constraint: int, params):
        self._lib = load()
        self.devices = [int(d) for d in devices]
        assert 1 <= len(self.devices) <= MULTI_MAX_DEVICES
        dv = (C.c_int * len(self.devices))(*self.devices)
        p = np.ascontiguousarray(params, dtype=np.float64)
        h = C.c_void_p()
        check(self._lib.fcamd_multi_create(dv, len(self.devices), int(model_id), int(constraint),
                                           p.ctypes.data_as(C.POINTER(C.c_double)), p.size, C.byref(h)))
        self.handle = h

    def plan(self, n: int) -> int:
        """number of devices a call over ``n`` points uses"""
        used = C.c_int()
        check(self._lib.fcamd_multi_plan(self.handle, int(n), 0, C.byref(used), None, None))
        return int(used.value)

    def bounds(self, n: int, k: int) -> tuple[int, int]:
        lo, hi = C.c_int64(), C.c_int64()
        check(self._lib.fcamd_multi_plan(self.handle, int(n), int(k), None, C.byref(lo), C.byref(hi)))
        return int(lo.value), int(hi.value)

    def evaluate_host(self, t, del_t, n, grad_ptr, stress_ptr, tangent_ptr, hist_ptrs) -> Stats:
        arr, nh = _ptr_array(hist_ptrs)
        st = Stats()
        check(self._lib.fcamd_multi_evaluate_host(self.handle, float(t), float(del_t), int(n), C.c_void_p(grad_ptr),
                                                  C.c_void_p(stress_ptr), C.c_void_p(tangent_ptr or 0), arr, nh, C.byref(st)))
        return st

    def register_host_buffer(self, arr: np.ndarray) -> None:
        check(self._lib.fcamd_multi_register_host_buffer(self.handle, C.c_void_p(arr.ctypes.data), arr.nbytes))

    def unregister_host_buffer(self, arr: np.ndarray) -> None:
        check(self._lib.fcamd_multi_register_host_buffer(self.handle, C.c_void_p(arr.ctypes.data), 0))  # bytes = 0: unregister

    def last_host_mode(self) -> tuple[int, int]:
        """(HOST_* flags OR-ed over the devices of the last call, number of devices it used)"""
        return self.get_option("last_host_mode"), self.get_option("last_n_used")

    def get_option(self, name: str) -> int:
        v = C.c_longlong()
        check(self._lib.fcamd_multi_get_option(self.handle, name.encode(), C.byref(v)))
        return int(v.value)

    def set_option(self, name: str, value: int) -> None:
        check(self._lib.fcamd_multi_set_option(self.handle, name.encode(), int(value)))

    def close(self) -> None:
        if self.handle:
            self._lib.fcamd_multi_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        if sys is None or sys.is_finalizing():  # process exit releases everything; the HIP runtime may already be going down
            return
        try:
            self.close()
        except Exception:
            pass


class MultiState:
    """``fcamd_multi_state``: committed + trial stress / history of n points, sliced over the devices of a ``Multi``."""

    def __init__(self, multi: Multi, n: int, flags: int = 0):
        self.multi, self._lib, self.n = multi, multi._lib, int(n)
        h = C.c_void_p()
        check(self._lib.fcamd_multi_state_create(multi.handle, int(n), int(flags), C.byref(h)))
        self.handle = h

    def set(self, stress_ptr, hist_ptrs) -> None:
        arr, nh = _ptr_array(hist_ptrs)
        check(self._lib.fcamd_multi_state_set(self.handle, C.c_void_p(stress_ptr or 0), arr, nh))

    def get(self, trial: bool, stress_ptr, hist_ptrs) -> None:
        arr, nh = _ptr_array(hist_ptrs)
        check(self._lib.fcamd_multi_state_get(self.handle, int(bool(trial)), C.c_void_p(stress_ptr or 0), arr, nh))

    def evaluate(self, t, del_t, grad_ptr, stress_ptr, tangent_ptr, flags: int = 0) -> Stats:
        st = Stats()
        check(self._lib.fcamd_multi_state_evaluate(self.handle, float(t), float(del_t), C.c_void_p(grad_ptr),
                                                   C.c_void_p(stress_ptr or 0), C.c_void_p(tangent_ptr or 0), int(flags), C.byref(st)))
        return st

    def commit(self) -> None:
        check(self._lib.fcamd_multi_state_commit(self.handle))

    def close(self) -> None:
        if self.handle and self.multi.handle:
            self._lib.fcamd_multi_state_destroy(self.handle)
        self.handle = C.c_void_p()

    def __del__(self):
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


class Model:
    """Handle of one constitutive law on one context."""

    def __init__(self, ctx: Context, model_id: int, constraint: int, params):
        self.ctx = ctx
        self._lib = ctx._lib
        p = np.ascontiguousarray(params, dtype=np.float64)
        h = C.c_void_p()
        check(self._lib.fcamd_model_create(ctx.handle, int(model_id), int(constraint),
                                           p.ctypes.data_as(C.POINTER(C.c_double)), p.size, C.byref(h)))
        self.handle = h
        info = self.info()
        self.history_fields: list[tuple[str, int]] = [(info.history_name[k].decode(), int(info.history_dim[k]))
                                                      for k in range(info.n_history)]

    def info(self) -> ModelInfo:
        """``fcamd_model_get_info``: constraint, array widths per point, history fields."""
        info = ModelInfo()
        check(self._lib.fcamd_model_get_info(self.handle, C.byref(info)))
        return info

    @property
    def constraint(self) -> int:
        """the StressStrainConstraint value of the handle"""
        return int(self.info().constraint)

    @property
    def dims(self) -> tuple[int, int]:
        """(stress_strain_dim, geometric_dim)"""
        i = self.info()
        return int(i.stress_strain_dim), int(i.geometric_dim)

    def _ptr_array(self, ptrs):
        if not ptrs:
            return None, 0
        arr = (C.c_void_p * len(ptrs))(*[C.c_void_p(int(p)) for p in ptrs])
        return arr, len(ptrs)

    def evaluate_host(self, t, del_t, n, grad_ptr, stress_ptr, tangent_ptr, hist_ptrs) -> Stats:
        arr, nh = self._ptr_array(hist_ptrs)
        st = Stats()
        _flush_recorded()
        status = self._lib.fcamd_evaluate_host(self.handle, float(t), float(del_t), int(n),
                                               C.c_void_p(grad_ptr), C.c_void_p(stress_ptr),
                                               C.c_void_p(tangent_ptr or 0), arr, nh, C.byref(st))
        check(status)
        return st

    def evaluate_device(self, t, del_t, n, grad_ptr, stress_ptr, tangent_ptr, hist_ptrs,
                        stress_prev_ptr=None, hist_prev_ptrs=None) -> None:
        """in place, or -- with ``stress_prev_ptr`` / ``hist_prev_ptrs`` -- committed -> trial (the header's
        fcamd_evaluate_device / fcamd_evaluate_device_from shorthands)"""
        self.evaluate_device_ex(t, del_t, n, grad_ptr, stress_ptr if stress_prev_ptr is None else stress_prev_ptr, stress_ptr,
                                tangent_ptr, hist_ptrs if hist_prev_ptrs is None else hist_prev_ptrs, hist_ptrs)

    def evaluate_device_ex(self, t, del_t, n, grad_ptr, stress_prev_ptr, stress_ptr, tangent_ptr, hist_prev_ptrs,
                           hist_ptrs, rows_ptr=None, mask_ptr=None, flags: int = 0, stress2_ptr=None,
                           counters_ptr=None, packed_mask_ptrs=None, wrapper_constraint: int = 0, stress3d_ptr=None) -> None:
        """``fcamd_evaluate_device_ex``: THE device entry, every form of the call in one argument struct.  Inside a
        ``batched_launches()`` block the call is recorded and leaves with the block's other calls as ONE
        ``fcamd_evaluate_batch``."""
        arr, nh = self._ptr_array(hist_ptrs)
        parr, _ = self._ptr_array(hist_prev_ptrs)
        pm_prev, pm = packed_mask_ptrs or (None, None)
        x = EvalArgs(grad_ptr, stress_prev_ptr, stress_ptr, tangent_ptr or None, parr, arr, nh, rows_ptr or None,
                     mask_ptr or None, int(flags), stress2_ptr or None, counters_ptr or None, pm_prev or None, pm or None,
                     int(wrapper_constraint), stress3d_ptr or None)
        batch = getattr(_tls, "batch", None)
        if batch is not None and not wrapper_constraint:
            batch.add(self, float(t), float(del_t), int(n), x, (arr, parr))
            return
        _flush_recorded()  # (the fused wrapper form is not batched: what was recorded before it runs before it)
        check(self._lib.fcamd_evaluate_device_ex(self.handle, float(t), float(del_t), int(n), C.byref(x)))

    def evaluate_device_wrapped(self, wrapper_constraint, t, del_t, n, grad_ptr, stress_ptr, tangent_ptr, stress3d_ptr,
                                hist_ptrs) -> None:
        """the fused 3D -> 1D/2D wrapper form (``wrapper_constraint`` + ``stress_3d`` of fcamd_eval_args), in place"""
        if n > 0 and not stress3d_ptr:
            raise ValueError("stress_3d is NULL")
        arr, nh = self._ptr_array(hist_ptrs)
        x = EvalArgs(grad_ptr, stress_ptr, stress_ptr, tangent_ptr or None, arr, arr, nh, None, None, 0, None, None, None, None,
                     int(wrapper_constraint) or -1, stress3d_ptr or None)
        _flush_recorded()
        check(self._lib.fcamd_evaluate_device_ex(self.handle, float(t), float(del_t), int(n), C.byref(x)))

    def evaluate_resident(self, t, del_t, n, grad_host_ptr, stress_prev_ptr, stress_ptr, hist_prev_ptrs, hist_ptrs,
                          mask_ptr, stress_host_ptr, tangent_host_ptr, flags: int = 0, packed_mask_ptrs=None) -> Stats:
        arr, nh = self._ptr_array(hist_ptrs)
        parr, _ = self._ptr_array(hist_prev_ptrs)
        pm_prev, pm = packed_mask_ptrs or (None, None)
        x = EvalArgs(grad_host_ptr, stress_prev_ptr, stress_ptr, None, parr, arr, nh, None, mask_ptr or None, int(flags), None, None,
                     pm_prev or None, pm or None, 0, None)
        st = Stats()
        _flush_recorded()
        status = self._lib.fcamd_evaluate_resident(self.handle, float(t), float(del_t), int(n), C.byref(x),
                                                   C.c_void_p(stress_host_ptr or 0), C.c_void_p(tangent_host_ptr or 0), C.byref(st))
        check(status)
        return st

    def evaluate_device_from_sparse(self, t, del_t, n, grad_ptr, stress_prev_ptr, stress_ptr, tangent_ptr,
                                    hist_prev_ptrs, hist_ptrs, mask_ptr) -> None:
        if n > 0 and not mask_ptr:
            raise ValueError("history_mask is NULL")
        self.evaluate_device_ex(t, del_t, n, grad_ptr, stress_prev_ptr, stress_ptr, tangent_ptr, hist_prev_ptrs, hist_ptrs,
                                mask_ptr=mask_ptr)

    def evaluate_device_indexed(self, t, del_t, n, grad_ptr, stress_prev_parent_ptr, stress_parent_ptr,
                                tangent_parent_ptr, rows_ptr, hist_prev_ptrs, hist_ptrs) -> None:
        if n > 0 and not rows_ptr:
            raise ValueError("parent_rows is NULL")
        self.evaluate_device_ex(t, del_t, n, grad_ptr, stress_prev_parent_ptr, stress_parent_ptr, tangent_parent_ptr,
                                hist_prev_ptrs, hist_ptrs, rows_ptr=rows_ptr)

    def last_stats(self) -> Stats:
        st = Stats()
        check(self._lib.fcamd_model_last_stats(self.handle, C.byref(st)))
        return st

    def last_kernel_ms(self) -> float:
        """time of the model's last entry (context option "timing"): ``fcamd_stats.kernel_ms`` of ``fcamd_model_last_stats``"""
        ms = float(self.last_stats().kernel_ms)
        if ms < 0.0:
            raise ValueError("timing was not enabled for the last launch")
        return ms

    def close(self) -> None:
        if self.handle:
            self._lib.fcamd_model_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):  # best effort
        try:
            self.close()
        except Exception:
            pass
