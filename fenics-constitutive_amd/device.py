"""Array plumbing between the ndarray interface and the C ABI.

Two kinds of arrays are accepted everywhere the reference takes ``np.ndarray``:

* ``numpy.ndarray`` (float64, C-contiguous)  -> host path, ``fcamd_evaluate_host``:
  the library stages the arrays to the GPU, runs the kernel and writes the results back in
  place.  This is the drop-in path for dolfinx ``Function.x.array`` views.
* ``torch.Tensor`` on a ROCm device (float64, contiguous) -> device path,
  ``fcamd_evaluate_device`` on torch's current stream, zero copies.

PyTorch is used for device memory and streams only.  No CPU fallback exists: without the
built library and a GPU, evaluation raises.
"""

from __future__ import annotations

import threading

import numpy as np

from . import _capi
from .interfaces import IncrSmallStrainModel, StressStrainConstraint


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


def _check_numpy(name: str, a: np.ndarray) -> np.ndarray:
    if not isinstance(a, np.ndarray):
        raise TypeError(f"{name} must be a numpy.ndarray or a torch CUDA tensor, got {type(a).__name__}")
    if a.dtype != np.float64:
        raise TypeError(f"{name} must be float64, got {a.dtype}")
    if not a.flags.c_contiguous:
        raise TypeError(f"{name} must be C-contiguous")
    return a


def _check_torch(name: str, a):
    import torch

    if a.dtype != torch.float64:
        raise TypeError(f"{name} must be float64, got {a.dtype}")
    if not a.is_cuda:
        raise TypeError(f"{name} must live on the GPU (or be a numpy array)")
    if not a.is_contiguous():
        raise TypeError(f"{name} must be contiguous")
    return a


def _current_stream_ptr(device_index: int) -> int:
    import torch

    return int(torch.cuda.current_stream(device_index).cuda_stream)


#: FCAMD_EVAL_SPLIT_HISTORY: the two arrays that replace the 7-double ``history`` rows of the comfe-rs plasticity laws
SPLIT_HISTORY_FIELDS = (("scalar", 1), ("rows", 6))
SPLIT_HISTORY_LAWS = ("MisesPlasticityLinearHardening3D", "DruckerPrager3D", "DruckerPragerHyperbolic3D")


def split_history_rows(history7):
    """``{"history": [7 n]}`` rows (device tensor) -> ``{"scalar": [n], "rows": [6 n]}`` (new contiguous tensors)"""
    v = history7.view(-1, 7)
    return {"scalar": v[:, 0].contiguous(), "rows": v[:, 1:].contiguous().view(-1)}


def join_history_rows(split):
    """the inverse: the reference's ``history`` array assembled from the split layout (new tensor)"""
    import torch

    return torch.cat([split["scalar"].view(-1, 1), split["rows"].view(-1, 6)], dim=1).reshape(-1)


def _tile_bits(mask, n):
    """[tiles, 64] 0/1 matrix of the bits of one int64 word per 64-point tile"""
    import torch

    shifts = torch.arange(64, device=mask.device, dtype=torch.int64)
    return (mask[: (n + 63) // 64, None] >> shifts[None, :]) & 1


def pack_rows(rows):
    """The packed layout of a plastic-strain array (FCAMD_EVAL_PACKED_HISTORY, include/fcamd.h): ``rows`` is the reference's
    ``[6 n]`` array (device tensor); returns ``(packed [6 n], ever [tiles] int64)`` -- per 64-point tile the rows that are
    not all +0.0 (bitwise) at the head of the tile's slot in ascending point order, and the tile's EVER mask.  The rest of
    a slot is zero-filled here (undefined by contract)."""
    import torch

    n = rows.numel() // 6
    tiles = (n + 63) // 64
    r = rows.view(n, 6)
    nz = (r.view(torch.int64) != 0).any(dim=1)  # bitwise: a -0.0 component keeps its row
    bits = torch.zeros(tiles * 64, dtype=torch.int64, device=rows.device)
    bits[:n] = nz
    bits = bits.view(tiles, 64)
    shifts = torch.arange(64, device=rows.device, dtype=torch.int64)
    ever = (bits << shifts[None, :]).sum(dim=1)  # distinct bits: the sum is the OR (bit 63 wraps to the sign bit)
    dst = (torch.arange(tiles, device=rows.device)[:, None] * 64 + torch.cumsum(bits, dim=1) - 1).reshape(-1)[:n]
    packed = torch.zeros_like(rows)
    packed.view(n, 6)[dst[nz]] = r[nz]
    return packed, ever


def unpack_rows(packed, ever, n=None):
    """The inverse of ``pack_rows``: the reference's ``[6 n]`` array (new tensor) from the packed layout."""
    import torch

    n = packed.numel() // 6 if n is None else n
    bits = _tile_bits(ever, n)
    src = (torch.arange(bits.shape[0], device=packed.device)[:, None] * 64 + torch.cumsum(bits, dim=1) - 1).reshape(-1)[:n]
    sel = bits.reshape(-1)[:n].bool()
    rows = torch.zeros(6 * n, dtype=packed.dtype, device=packed.device)
    rows.view(n, 6)[sel] = packed.view(-1, 6)[src[sel]]
    return rows


#: Below this many points ONE ndarray ``evaluate`` call is bound by its fixed cost -- the launch and the PCIe round trips of a
#: kernel that reads and writes host memory, 45-60 us on MI355X -- and a NumPy evaluation of the same law on the host is
#: faster (measured per law by bench.py, ``cpu_baseline.small_call_crossover``; table in INTEGRATION.md).  A dolfinx rank
#: often holds 1e3-1e5 quadrature points per law (solver/_lawonsubmesh.py:86-94): the first such call of a law warns, once.
#: Measured on an MI355X box (EPYC 9575F host, round 4): LinearElasticityModel 2980, SpringMaxwellModel 1969, SpringKelvinModel 708,
#: VonMises3D < 64 points (against a vectorised NumPy restatement; the reference itself evaluates VonMises3D point by point in
#: Python, 30-70 us per POINT -- there the engine wins from the first point on).  ``FCAMD_SMALL_CALL_WARNING=0`` silences the warning.
SMALL_CALL_POINTS = {"LinearElasticityModel": 3000, "LinearElasticity3D": 3000, "SpringMaxwellModel": 2000, "SpringKelvinModel": 700,
                     "VonMises3D": 64}
_small_call_warned: set = set()


def _warn_small_call(law, n: int) -> None:
    import os
    import warnings

    name = type(law).__name__
    limit = SMALL_CALL_POINTS.get(name)
    if limit is None or n >= limit or n == 0 or name in _small_call_warned or os.environ.get("FCAMD_SMALL_CALL_WARNING") == "0":
        return
    _small_call_warned.add(name)
    warnings.warn(f"{name}.evaluate on {n} points: below ~{limit} points one call is bound by its fixed cost on the GPU (launch + PCIe round "
                  "trips, 45-60 us) and a NumPy evaluation of this law on the host is faster; batch several laws / cells into one "
                  "call, keep the state resident (ResidentState), or stay on the CPU for this size (INTEGRATION.md, 'small calls').  "
                  "Shown once per law; FCAMD_SMALL_CALL_WARNING=0 silences it.", RuntimeWarning, stacklevel=3)


class DeviceLaw(IncrSmallStrainModel):
    """Base of all GPU-backed laws: owns the C model handle (created lazily, per device)
    and implements ``evaluate`` on top of the C ABI with the reference's validation."""

    #: set by subclasses
    _model_id: int = 0
    #: Opt-in: page-lock (and map) the caller's NumPy arrays the first time they are passed; once every
    #: array of a call is registered the kernel runs directly on them (zero copy: 93 -> 140 Mpts/s at
    #: 1e7 points, 145 -> 40 us per call at 1e3 points).  Meant for the dolfinx loop, which hands over the same
    #: ``Function.x.array`` views every Newton iteration (solver/_lawonsubmesh.py:87-94).  The law
    #: keeps a reference to every pinned array until ``unpin_arrays()``: page-locked memory must not
    #: be freed while registered (a later array at the same address would DMA through stale pages).
    auto_pin = False

    def __init__(self, parameter_vector, constraint: StressStrainConstraint = None):
        self._constraint = constraint if constraint is not None else StressStrainConstraint.FULL
        self._parameter_vector = [float(p) for p in parameter_vector]
        # C model handles, one per (thread, device), in thread-local storage: they are destroyed -- and with
        # the last of them the thread's context -- when the thread that created them ends
        self._tls = threading.local()
        self.n_handles_created = 0
        self.last_stats = None
        # single-process multi-GPU host path: None = one device (default_device()), else the device ordinals the
        # ndarray path spreads every call over (use_devices; environment default FCAMD_DEVICES)
        self._devices = _capi.default_devices()
        self._multi_handle = None
        self._multi_lock = threading.Lock()  # two threads' first calls must not build two handles (the handle itself serialises calls)

    # -- interface properties --------------------------------------------------------------
    @property
    def constraint(self) -> StressStrainConstraint:
        return self._constraint

    # -- C handle ---------------------------------------------------------------------------
    def _handle(self, device: int = 0) -> _capi.Model:
        # one C handle per (device, thread): contexts are per thread (staging buffers, streams, the
        # registry of page-locked ranges), so a law object shared by several threads stays thread-compatible
        handles = self._tls.__dict__.setdefault("handles", {})
        h = handles.get(device)
        if h is None:
            ctx = _capi.get_context(device)
            h = handles[device] = _capi.Model(ctx, self._model_id, self._constraint.value, self._parameter_vector)
            self.n_handles_created += 1
        return h

    # -- several GPUs, one process ----------------------------------------------------------------
    def use_devices(self, devices) -> "DeviceLaw":
        """NumPy-array ``evaluate`` calls of this law run on these GPUs at once (list of device ordinals; ``None``:
        back to one device): every call is cut into contiguous slices, each device evaluates its slice of the
        caller's arrays in place over its own PCIe link (``fcamd_multi_evaluate_host``) -- the single-assembler mode
        of a dolfinx process that drives a whole node; results are bit-identical to one device.  The environment
        variable ``FCAMD_DEVICES`` ("0,1,2,3" / "all") sets the default for every law.  Returns ``self``."""
        devices = None if devices is None else [int(d) for d in devices]
        if devices != self._devices:
            self.unpin_arrays()
            if self._multi_handle is not None:
                self._multi_handle.close()
                self._multi_handle = None
            self._devices = devices
        return self

    @property
    def devices(self):
        return None if self._devices is None else list(self._devices)

    def _multi(self) -> "_capi.Multi":
        with self._multi_lock:
            if self._multi_handle is None:
                self._multi_handle = _capi.Multi(self._devices, self._model_id, self._constraint.value, self._parameter_vector)
            return self._multi_handle

    def _history_arrays(self, history):
        """Order the caller's history dict by the law's field order."""
        fields = self._history_fields()
        if not fields:
            return []
        if history is None:
            raise ValueError("history must not be None")
        return [history[name] for name, _ in fields]

    def _history_fields(self) -> list[tuple[str, int]]:
        hd = self.history_dim
        return [] if hd is None else list(hd.items())

    # -- the hot call --------------------------------------------------------------------------
    def evaluate(self, t, del_t, grad_del_u, stress, tangent, history, check: bool = False) -> None:
        """``IncrSmallStrainModel.evaluate`` (interfaces.py:82-101): overwrite ``stress``,
        ``tangent`` and every history array in place.

        NumPy arrays: synchronous, raises the reference's exceptions itself.  Device tensors: the launch is
        asynchronous on torch's current stream; ``check=True`` synchronises and raises the reference's
        ``RuntimeError`` on Newton non-convergence right here, as the reference does inside ``evaluate``
        (mises_plasticity_isotropic_hardening.py:141-143) -- otherwise call ``device_stats()`` (or let
        ``ResidentState.update()`` do it) before the results are committed."""
        hist = self._history_arrays(history)
        gd2 = self.geometric_dim**2
        sd = self.stress_strain_dim
        n = _size(grad_del_u) // gd2
        # size assertion of the reference (linear_elasticity_model.py:36-40, interfaces.rs:402-415)
        assert n == _size(stress) // sd and (tangent is None or n == _size(tangent) // (sd * sd)), (
            "Stress, strain, and tangent lengths do not match"
        )
        assert _size(grad_del_u) == n * gd2 and _size(stress) == n * sd, "Input arrays are not of the correct length"
        for (name, dim), h in zip(self._history_fields(), hist):
            assert _size(h) == n * dim, f"history '{name}' has the wrong length"
        if _is_torch(grad_del_u):
            self._evaluate_device(t, del_t, n, grad_del_u, stress, tangent, hist)
            if check:
                self.device_stats(grad_del_u.device.index or 0)
        else:
            _warn_small_call(self, n)
            self._evaluate_host(t, del_t, n, grad_del_u, stress, tangent, hist)

    def _evaluate_host(self, t, del_t, n, grad, stress, tangent, hist) -> None:
        _check_numpy("grad_del_u", grad)
        _check_numpy("stress", stress)
        if tangent is not None:
            _check_numpy("tangent", tangent)
        for (name, _), h in zip(self._history_fields(), hist):
            _check_numpy(f"history['{name}']", h)
        multi = self._devices is not None
        m = self._multi() if multi else self._handle(_capi.default_device())
        if self.auto_pin:
            self._pin(m if multi else m.ctx, [grad, stress] + ([] if tangent is None else [tangent]) + list(hist))
        self.last_stats = m.evaluate_host(
            t, del_t, n, grad.ctypes.data, stress.ctypes.data,
            None if tangent is None else tangent.ctypes.data, [h.ctypes.data for h in hist],
        )

    def _pin(self, ctx, arrays) -> None:
        pinned = self.__dict__.setdefault("_pinned", {})  # (ptr, nbytes) -> (array kept alive, its context) | None
        for a in arrays:
            key = (a.ctypes.data, a.nbytes)
            if key in pinned or a.nbytes < (1 << 16):
                continue
            try:
                ctx.register_host_buffer(a)
                pinned[key] = (a, ctx)
            except (RuntimeError, ValueError):
                pinned[key] = None  # e.g. overlaps another registration: the calls page-lock it (or find it locked) themselves

    def pin_host_arrays(self, *arrays) -> None:
        """Page-lock caller-owned NumPy arrays that will be passed to ``evaluate`` repeatedly (the
        dolfinx ``Function.x.array`` views of one problem): the host path then DMAs directly.  The law
        keeps the arrays alive until ``unpin_arrays()``."""
        target = self._multi() if self._devices is not None else self._handle(_capi.default_device()).ctx
        self._pin(target, [_check_numpy("array", a) for a in arrays])

    def unpin_arrays(self) -> None:
        """Undo ``auto_pin`` registrations and drop the references that kept the arrays alive."""
        pinned = self.__dict__.pop("_pinned", {})
        for entry in pinned.values():
            if entry is not None:
                a, ctx = entry  # the context (thread) that registered it, whichever thread runs this
                ctx.unregister_host_buffer(a)

    def __del__(self):
        # The law holds the only references that keep auto-pinned arrays alive; once it is gone they may
        # be freed, and a registration that outlives its memory would make later accesses at the same
        # address go through stale pages.  Unregister first (best effort at interpreter shutdown).
        try:
            self.unpin_arrays()
            if self._multi_handle is not None:
                self._multi_handle.close()
        except Exception:
            pass

    def _evaluate_device(self, t, del_t, n, grad, stress, tangent, hist,
                         stress_prev=None, hist_prev=None) -> None:
        _check_torch("grad_del_u", grad)
        _check_torch("stress", stress)
        if tangent is not None:
            _check_torch("tangent", tangent)
        for (name, _), h in zip(self._history_fields(), hist):
            _check_torch(f"history['{name}']", h)
        dev = grad.device.index or 0
        m = self._handle(dev)
        m.ctx.set_stream(_current_stream_ptr(dev))
        m.evaluate_device(
            t, del_t, n, grad.data_ptr(), stress.data_ptr(),
            None if tangent is None else tangent.data_ptr(), [h.data_ptr() for h in hist],
            None if stress_prev is None else _check_torch("stress_prev", stress_prev).data_ptr(),
            None if hist_prev is None else [_check_torch("history_prev", h).data_ptr() for h in hist_prev],
        )

    def evaluate_from(self, t, del_t, grad_del_u, stress_prev, stress, tangent, history_prev, history,
                      history_mask=None, sparse_tangent: bool = False, counters=None,
                      split_history: bool = False, packed_masks=None) -> None:
        """Out-of-place device evaluate: read the committed state (``stress_prev``,
        ``history_prev``), write the trial state (``stress``, ``history``).  Fuses the two
        copies the reference makes before every call (solver/_lawonsubmesh.py:58-61,
        solver/_history.py:64-79).  Device tensors only.  ``sparse_tangent`` (with ``history_mask``):
        ``tangent`` holds the tangent of the previous evaluate with this mask; only the rows of plastic /
        formerly plastic points are rewritten (FCAMD_EVAL_SPARSE_TANGENT, include/fcamd.h).  ``counters``:
        caller-owned int64 device tensor of ``_capi.COUNTER_WORDS`` words that receives this launch's
        statistics instead of the law's own counters (``fcamd_eval_args.counters``; read it with
        ``read_counters``).  ``split_history`` (the comfe-rs plasticity
        laws): the histories are dicts ``{"scalar": n, "rows": 6 n}`` instead of the reference's ``{"history": 7 n}``
        (FCAMD_EVAL_SPLIT_HISTORY: ``SPLIT_HISTORY_FIELDS``).  ``packed_masks = (ever_prev, ever)`` (with ``history_mask``;
        int64 device tensors, one word per 64-point tile): the plastic-strain arrays of ``history_prev`` / ``history``
        (``eps_n`` resp. the split ``rows``) are in the packed layout of ``pack_rows`` (FCAMD_EVAL_PACKED_HISTORY)."""
        if split_history:
            hist = [history[k] for k, _ in SPLIT_HISTORY_FIELDS]
            hprev = [history_prev[k] for k, _ in SPLIT_HISTORY_FIELDS]
        else:
            hist = self._history_arrays(history)
            hprev = self._history_arrays(history_prev)
        gd2, sd = self.geometric_dim**2, self.stress_strain_dim
        n = _size(grad_del_u) // gd2
        assert n == _size(stress) // sd == _size(stress_prev) // sd and (tangent is None or n == _size(tangent) // (sd * sd))
        if history_mask is None and counters is None and not split_history:
            self._evaluate_device(t, del_t, n, grad_del_u, stress, tangent, hist, stress_prev, hprev)
            return
        # sparse trial history (fcamd_evaluate_device_from_sparse): see ResidentState
        import torch

        if history_mask is not None:
            assert history_mask.dtype == torch.int64 and history_mask.is_cuda and history_mask.numel() >= (n + 63) // 64
        for x in (grad_del_u, stress, stress_prev):
            _check_torch("array", x)
        dev = grad_del_u.device.index or 0
        m = self._handle(dev)
        m.ctx.set_stream(_current_stream_ptr(dev))
        if (sparse_tangent and tangent is not None) or counters is not None or split_history or packed_masks is not None:
            flags = _capi.EVAL_SPARSE_TANGENT if (sparse_tangent and tangent is not None and history_mask is not None) else 0
            pm = None
            if packed_masks is not None:
                assert history_mask is not None, "packed history: with history_mask"
                for mk in packed_masks:
                    assert mk.dtype == torch.int64 and mk.is_cuda and mk.numel() >= (n + 63) // 64
                flags |= _capi.EVAL_PACKED_HISTORY
                pm = (packed_masks[0].data_ptr(), packed_masks[1].data_ptr())
            if split_history:
                flags |= _capi.EVAL_SPLIT_HISTORY
            m.evaluate_device_ex(
                t, del_t, n, grad_del_u.data_ptr(), stress_prev.data_ptr(), stress.data_ptr(),
                None if tangent is None else _check_torch("tangent", tangent).data_ptr(),
                [h.data_ptr() for h in hprev], [h.data_ptr() for h in hist],
                None, None if history_mask is None else history_mask.data_ptr(), flags,
                counters_ptr=_counters_ptr(counters), packed_mask_ptrs=pm)
            return
        m.evaluate_device_from_sparse(
            t, del_t, n, grad_del_u.data_ptr(), stress_prev.data_ptr(), stress.data_ptr(),
            None if tangent is None else tangent.data_ptr(), [h.data_ptr() for h in hprev],
            [h.data_ptr() for h in hist], history_mask.data_ptr())

    def evaluate_indexed(self, t, del_t, grad_del_u, stress_prev_parent, stress_parent, tangent_parent,
                         parent_rows, history_prev, history, history_mask=None, sparse_tangent: bool = False,
                         counters=None, packed_masks=None) -> None:
        """Multi-material form: this law owns ``n = len(parent_rows)`` points whose stress/tangent
        rows live in PARENT arrays at ``parent_rows`` (int32 device tensor).  Reads the committed
        stress from ``stress_prev_parent`` rows, writes stress and tangent into the parent rows;
        ``grad_del_u`` and the history are local to the law.  Replaces map_to_sub + evaluate +
        map_to_parent of ``LawOnSubMesh`` (solver/_lawonsubmesh.py:58-95) by one launch."""
        import torch

        hist = self._history_arrays(history)
        hprev = hist if history_prev is None else self._history_arrays(history_prev)
        n = parent_rows.numel()
        assert _size(grad_del_u) == 9 * n, "grad_del_u has the wrong length"
        assert parent_rows.dtype == torch.int32 and parent_rows.is_cuda and parent_rows.is_contiguous()
        for x in (grad_del_u, stress_prev_parent, stress_parent):
            _check_torch("array", x)
        assert _size(stress_prev_parent) == _size(stress_parent)
        assert tangent_parent is None or _size(tangent_parent) == 6 * _size(stress_parent)
        for (name, dim), h in zip(self._history_fields(), hist):
            assert _size(h) == n * dim, f"history '{name}' has the wrong length"
        dev = grad_del_u.device.index or 0
        m = self._handle(dev)
        m.ctx.set_stream(_current_stream_ptr(dev))
        tan_ptr = None if tangent_parent is None else _check_torch("tangent", tangent_parent).data_ptr()
        if history_mask is None and counters is None and packed_masks is None:
            m.evaluate_device_indexed(
                t, del_t, n, grad_del_u.data_ptr(), stress_prev_parent.data_ptr(), stress_parent.data_ptr(),
                tan_ptr, parent_rows.data_ptr(), [h.data_ptr() for h in hprev], [h.data_ptr() for h in hist])
            return
        # sparse trial history on a submesh (fcamd_evaluate_device_ex): mask and history are local to the law
        if history_mask is not None:
            assert history_mask.dtype == torch.int64 and history_mask.is_cuda and history_mask.numel() >= (n + 63) // 64
        flags = _capi.EVAL_SPARSE_TANGENT if (sparse_tangent and tan_ptr is not None and history_mask is not None) else 0
        pm = None
        if packed_masks is not None:  # the law's (local) plastic-strain arrays in the packed layout (VonMises3D; ``pack_rows``)
            assert history_mask is not None, "packed history: with history_mask"
            flags |= _capi.EVAL_PACKED_HISTORY
            pm = (packed_masks[0].data_ptr(), packed_masks[1].data_ptr())
        m.evaluate_device_ex(
            t, del_t, n, grad_del_u.data_ptr(), stress_prev_parent.data_ptr(), stress_parent.data_ptr(), tan_ptr,
            [h.data_ptr() for h in hprev], [h.data_ptr() for h in hist], parent_rows.data_ptr(),
            None if history_mask is None else history_mask.data_ptr(), flags,
            counters_ptr=_counters_ptr(counters), packed_mask_ptrs=pm)

    def empty_tangent(self, grad_del_u, stress, history, del_t: float = 1.0, tries: int = 6):
        """A tangent array for ``evaluate(t, del_t, grad_del_u, stress, tangent, history)`` on device tensors, PLACED: on MI355X
        the kernel time follows where the driver puts the written arrays (above all the tangent) -- identical data in different
        hipMalloc allocations run 0.65-0.82 of the roofline, reproducibly per allocation.  Up to ``tries``
        candidate allocations are timed with this law's real launch on the caller's arrays (committed -> scratch trial arrays:
        ``stress`` and ``history`` are NOT modified) and the fastest is returned with what was measured:
        ``(tangent, {"candidate_ms": [...], "chosen": k})``.  What ``ResidentState(placement="auto")`` does for its own arrays,
        for a caller who keeps the reference's in-place protocol on tensors of their own; costs 4 launches per candidate, once."""
        import torch

        from .placement import fastest_allocation

        _check_torch("grad_del_u", grad_del_u), _check_torch("stress", stress)
        sd = self.stress_strain_dim
        n = _size(stress) // sd
        scratch_s = torch.empty_like(stress)
        scratch_h = None if history is None else {k: torch.empty_like(v) for k, v in history.items()}

        def probe(tan):
            self.evaluate_from(0.0, del_t, grad_del_u, stress, scratch_s, tan, history, scratch_h)

        tangent, info = fastest_allocation(sd * sd * n, probe, tries=tries, device=stress.device)
        del scratch_s, scratch_h
        torch.cuda.empty_cache()
        return tangent, info

    def raise_for_stats(self, st) -> None:
        """The reference's errors for the counters of a finished launch: the Drucker-Prager tip
        (drucker_prager_classic.rs:82) and Newton non-convergence (general.rs:186 resp.
        mises_plasticity_isotropic_hardening.py:141-143; same messages as the host entries)."""
        if getattr(st, "n_domain", 0):
            raise RuntimeError("non-differentiable tip of Drucker-Prager surface reached")
        if st.n_nonconverged:
            raise RuntimeError("Plasticity3D: Newton-Raphson did not converge." if self._model_id >= _capi.COMFE_DRUCKER_PRAGER
                               else "Newton-Raphson method did not converge for plastic multiplier.")

    def device_stats(self, device: int = 0):
        """Synchronise and return the counters of the last device-path launch; raises
        RuntimeError like the reference if a Newton iteration did not converge."""
        st = self._handle(device).last_stats()
        self.last_stats = st
        self.raise_for_stats(st)
        return st


def _counters_ptr(counters):
    if counters is None:
        return None
    import torch

    assert counters.dtype == torch.int64 and counters.is_cuda and counters.is_contiguous() and \
        counters.numel() >= _capi.COUNTER_WORDS, "counters: contiguous int64 device tensor of COUNTER_WORDS words"
    return counters.data_ptr()


def new_counters(device):
    """Caller-owned counter buffer for ``evaluate_from`` / ``evaluate_indexed`` (``fcamd_eval_args.counters``)."""
    import torch

    return torch.zeros(_capi.COUNTER_WORDS, dtype=torch.int64, device=device)


def read_counters(counters) -> "_capi.Stats":
    """Synchronise with the launches that wrote ``counters`` (on the current stream) and sum the slots."""
    c = counters[: _capi.COUNTER_WORDS].cpu().view(_capi.COUNTER_SLOTS, 4).sum(dim=0).tolist()
    return _capi.Stats(int(c[0]), int(c[1]), int(c[2]), int(c[3]))


def _size(a) -> int:
    if isinstance(a, np.ndarray):
        return int(a.size)
    if _is_torch(a):
        return int(a.numel())
    return int(np.asarray(a).size)


def strain_from_grad_u_full(grad_u):
    """FULL-constraint ``strain_from_grad_u`` on the GPU (utils.py:187-208)."""
    import ctypes as C

    lib = _capi.load()
    if _is_torch(grad_u):
        import torch

        _check_torch("grad_u", grad_u)
        n = grad_u.numel() // 9
        out = torch.empty(6 * n, dtype=torch.float64, device=grad_u.device)
        dev = grad_u.device.index or 0
        ctx = _capi.get_context(dev)
        ctx.set_stream(_current_stream_ptr(dev))
        _capi.check(lib.fcamd_strain_from_grad_u_device(ctx.handle, n, C.c_void_p(grad_u.data_ptr()),
                                                        C.c_void_p(out.data_ptr()), 0))
        return out
    import torch

    g = np.ascontiguousarray(np.asarray(grad_u, dtype=np.float64)).reshape(-1)
    if not torch.cuda.is_available():
        raise RuntimeError("strain_from_grad_u(FULL) runs on the GPU and no HIP device is available")
    from .hostio import to_device, to_host

    d = to_device(g, torch.device("cuda", _capi.default_device()))
    return to_host(strain_from_grad_u_full(d))


def strain_from_grad_u_lowdim(grad_u, constraint):
    """1-D / 2-D ``strain_from_grad_u`` (utils.py:153-186) on the GPU with the kernels that already
    exist: expand the gradient to 3-D (``fcamd_convert_device``), run the FULL strain kernel, keep the
    mapped Mandel components -- [g0] resp. [g0, g3, 0, f (g1 + g2)], the reference's formulas."""
    import ctypes as C

    import torch

    lib = _capi.load()
    uniaxial = constraint.name.startswith("UNIAXIAL")
    gd2, sd = (1, 1) if uniaxial else (4, 4)
    host = not _is_torch(grad_u)
    if host:
        if not torch.cuda.is_available():
            raise RuntimeError("strain_from_grad_u runs on the GPU and no HIP device is available")
        from .hostio import to_device

        g = to_device(np.asarray(grad_u, dtype=np.float64).reshape(-1), torch.device("cuda", _capi.default_device()))
    else:
        g = _check_torch("grad_u", grad_u).reshape(-1)
    n = g.numel() // gd2
    dev = g.device.index or 0
    ctx = _capi.get_context(dev)
    ctx.set_stream(_current_stream_ptr(dev))
    g3 = torch.zeros(9 * n, dtype=torch.float64, device=g.device)
    e3 = torch.empty(6 * n, dtype=torch.float64, device=g.device)
    out = torch.empty(sd * n, dtype=torch.float64, device=g.device)
    up, down = (_capi.GRAD_1D_TO_3D, _capi.STRESS_3D_TO_1D) if uniaxial else (_capi.GRAD_2D_TO_3D, _capi.STRESS_3D_TO_2D)
    _capi.check(lib.fcamd_convert_device(ctx.handle, up, n, C.c_void_p(g.data_ptr()), C.c_void_p(g3.data_ptr())))
    _capi.check(lib.fcamd_strain_from_grad_u_device(ctx.handle, n, C.c_void_p(g3.data_ptr()), C.c_void_p(e3.data_ptr()), 0))
    _capi.check(lib.fcamd_convert_device(ctx.handle, down, n, C.c_void_p(e3.data_ptr()), C.c_void_p(out.data_ptr())))
    if host:
        from .hostio import to_host

        return to_host(out)
    return out
