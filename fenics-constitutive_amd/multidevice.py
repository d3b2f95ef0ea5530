"""One process, several GPUs: device-resident increment state for a single host assembler.

north_star's single-process mode: ONE dolfinx process assembles -- its gradient, stress and tangent arrays are host
memory (views of ``Function.x.array``, ``solver/_lawonsubmesh.py:87-94``) -- and the GPUs of the node evaluate.
``MultiDeviceResidentState`` is ``ResidentState`` for that mode: the committed and the trial copy of stress and history
of the n quadrature points live sliced over the devices (contiguous, tile-aligned slices, the rule of
``fcamd_shard_bounds``), every Newton iteration each device reads ITS slice of the gradient from and writes ITS slice
of stress and tangent to the caller's arrays over its own PCIe link, all devices at once
(``fcamd_multi_state_evaluate``; include/fcamd.h "one process, several GPUs").  There is no gather: nothing crosses
xGMI, the history never leaves its device, the commit is a pointer swap per device.

The ndarray ``evaluate`` of every law has the same spread without resident state: ``law.use_devices([0, 1, ...])``
(``fcamd_multi_evaluate_host``), or ``FCAMD_DEVICES=all`` in the environment of an unchanged dolfinx script.

``MultiDeviceResidentState`` does not involve PyTorch (its device arrays are owned by the C library);
``MultiDeviceProblemState`` -- the fused multi-material host flow over several GPUs -- is built from one
``ResidentProblemState`` (torch tensors) per device.
"""

from __future__ import annotations

import numpy as np

from . import _capi
from .device import SPLIT_HISTORY_LAWS, DeviceLaw, _check_numpy

__all__ = ["MultiDeviceResidentState", "MultiDeviceProblemState"]

_PLASTICITY = ("VonMises3D", "MisesPlasticityLinearHardening3D", "DruckerPrager3D", "DruckerPragerHyperbolic3D")
_CONST_TANGENT = ("LinearElasticityModel", "LinearElasticity3D", "SpringMaxwellModel", "SpringKelvinModel")


class MultiDeviceResidentState:
    """Committed + trial state of ``n`` points of ``law`` on the GPUs ``devices`` (default: the law's
    ``use_devices`` list, else ``FCAMD_DEVICES``, else every visible device).

    Protocol = the reference's increment loop (``solver/_solver.py:130-159``): ``evaluate_into`` any number of times
    per increment (Newton iterations; the committed state is never modified), ``update()`` commits.  The shortcuts
    of ``ResidentState.evaluate_into`` apply per device: the point-independent tangent of linear elasticity / SLS is
    written into the caller's array once per ``del_t``; for the plasticity laws only the tangent rows of plastic /
    formerly plastic points are rewritten from the second call into the same array on (``sparse_tangent``); the
    comfe-rs laws keep their history as [scalar, eps_p rows] on the device (``split_history``)."""

    def __init__(self, law: DeviceLaw, n: int, devices=None, stress0=None, history0=None, split_history: bool = True,
                 sparse_tangent: bool = True, reuse_constant_tangent: bool = True):
        if devices is None:
            devices = law.devices or _capi.default_devices()
        if devices is None:
            import torch

            devices = list(range(torch.cuda.device_count()))
        assert len(devices) >= 1, "no GPU to run on"
        self.law, self.n, self.devices = law, int(n), [int(d) for d in devices]
        self._gd2, self._sd = law.geometric_dim**2, law.stress_strain_dim
        name = type(law).__name__
        self._fields = [] if law.history_dim is None else list(law.history_dim.items())
        self._split = bool(split_history) and name in SPLIT_HISTORY_LAWS
        self._multi = _capi.Multi(self.devices, law._model_id, law.constraint.value, law._parameter_vector)
        self._state = _capi.MultiState(self._multi, self.n, _capi.EVAL_SPLIT_HISTORY if self._split else 0)
        self._const_tangent = bool(reuse_constant_tangent) and name in _CONST_TANGENT
        self._sparse_tangent = bool(sparse_tangent) and name in _PLASTICITY
        self._tangent_target = None    # (address, bytes) of the host array that holds the previous evaluate's tangent
        self._host_tangent_key = None  # (address, bytes, del_t) of the host array that holds the constant tangent
        self._host_tangent_ref = None
        self._pinned = []
        self._evaluated = False
        self._failed = None
        if stress0 is not None or history0 is not None:
            self.set_state(stress0, history0)

    # -- state -----------------------------------------------------------------------------------------
    def _hist_ptrs(self, history, writable: bool):
        if not self._fields:
            return []
        if history is None:
            raise ValueError("history must not be None")
        out = []
        for name, dim in self._fields:
            a = _check_numpy(f"history['{name}']", history[name])
            assert a.size == dim * self.n, f"history '{name}' has the wrong length"
            assert not writable or a.flags.writeable
            out.append(a.ctypes.data)
        return out

    def set_state(self, stress=None, history=None) -> None:
        """Committed state <- NumPy arrays (``None``: zeros); trial history = committed, every shortcut reset."""
        if stress is not None:
            _check_numpy("stress", stress)
            assert stress.size == self._sd * self.n, "stress has the wrong length"
        self._state.set(None if stress is None else stress.ctypes.data, None if history is None else self._hist_ptrs(history, False))
        self._tangent_target = self._host_tangent_key = None
        self._evaluated = False
        self._failed = None

    def download(self, stress: np.ndarray | None = None, history: dict | None = None, committed: bool = False) -> None:
        """Trial (default) or committed state -> the caller's NumPy arrays, in the law's reference layout."""
        if stress is not None:
            _check_numpy("stress", stress)
            assert stress.size == self._sd * self.n and stress.flags.writeable
        self._state.get(not committed, None if stress is None else stress.ctypes.data,
                        None if history is None else self._hist_ptrs(history, True))

    def _new_history(self):
        return None if not self._fields else {k: np.empty(d * self.n) for k, d in self._fields}

    @property
    def stress(self) -> np.ndarray:
        out = np.empty(self._sd * self.n)
        self.download(stress=out)
        return out

    @property
    def stress_committed(self) -> np.ndarray:
        out = np.empty(self._sd * self.n)
        self.download(stress=out, committed=True)
        return out

    @property
    def history(self):
        h = self._new_history()
        if h is not None:
            self.download(history=h)
        return h

    @property
    def history_committed(self):
        h = self._new_history()
        if h is not None:
            self.download(history=h, committed=True)
        return h

    def slices(self) -> list[tuple[int, int]]:
        """[lo, hi) of every device's points (``fcamd_shard_bounds(n, len(devices), k)``)."""
        return [_capi.shard_bounds(self.n, len(self.devices), k) for k in range(len(self.devices))]

    # -- page-locked caller arrays ---------------------------------------------------------------------
    def pin_host_arrays(self, *arrays) -> None:
        """Page-lock caller arrays that are passed call after call (the dolfinx ``Function.x.array`` views), once,
        for all devices; kept alive until ``unpin_arrays()``."""
        for a in arrays:
            _check_numpy("array", a)
            if any(b is a for b in self._pinned):
                continue
            self._multi.register_host_buffer(a)
            self._pinned.append(a)

    def unpin_arrays(self) -> None:
        pinned, self._pinned = self._pinned, []
        for a in pinned:
            self._multi.unregister_host_buffer(a)

    # -- the Newton-iteration call -----------------------------------------------------------------------
    def evaluate_into(self, t: float, del_t: float, grad_del_u: np.ndarray, stress: np.ndarray | None = None,
                      tangent: np.ndarray | None = None):
        """Trial state <- law(committed state, grad_del_u); trial stress / tangent into the caller's arrays.
        Synchronous; raises the reference's exceptions like the ndarray ``evaluate``."""
        _check_numpy("grad_del_u", grad_del_u)
        assert grad_del_u.size == self._gd2 * self.n, "grad_del_u has the wrong length"
        if stress is not None:
            _check_numpy("stress", stress)
            assert stress.size == self._sd * self.n, "stress has the wrong length"
        if tangent is not None:
            _check_numpy("tangent", tangent)
            assert tangent.size == self._sd * self._sd * self.n, "tangent has the wrong length"
        self._evaluated = True
        key = None
        if tangent is not None and self._const_tangent:
            key = (tangent.ctypes.data, tangent.nbytes, float(del_t) if type(self.law).__name__.startswith("Spring") else 0.0)
            if self._host_tangent_key == key:
                tangent = None  # the caller's array holds exactly what this call would write
            else:
                self._host_tangent_key = None
        target = None if tangent is None else (tangent.ctypes.data, tangent.nbytes)
        flags = _capi.EVAL_SPARSE_TANGENT if (self._sparse_tangent and target is not None and self._tangent_target == target) else 0
        self._tangent_target = None
        try:
            st = self._state.evaluate(t, del_t, grad_del_u.ctypes.data, None if stress is None else stress.ctypes.data,
                                      None if tangent is None else tangent.ctypes.data, flags)
        except Exception as e:
            self._failed = e
            raise
        self._failed = None
        self._tangent_target = target
        if key is not None and tangent is not None:
            self._host_tangent_key = key
        if tangent is not None:
            self._host_tangent_ref = tangent  # identified by address above: keep the memory from being handed out again
        self.law.last_stats = st
        return st

    def update(self) -> None:
        """Commit the trial state (``IncrSmallStrainProblem.update``, solver/_solver.py:149-159): a pointer swap per
        device.  A failed evaluate (Newton non-convergence, the Drucker-Prager tip) is never committed."""
        if not self._evaluated:
            raise RuntimeError("update() before any evaluate() of this increment")
        if self._failed is not None:
            raise RuntimeError(f"the last evaluate failed, nothing to commit: {self._failed}")
        self._state.commit()
        self._evaluated = False

    def last_host_mode(self) -> tuple[int, int]:
        return self._multi.last_host_mode()

    def close(self) -> None:
        self.unpin_arrays()
        self._state.close()
        self._multi.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiDeviceProblemState:
    """Several laws on one mesh, one assembling process, several GPUs: ``ResidentProblemState`` (the fused multi-material
    host flow: every law's kernel reads its local gradient from and writes its rows of the problem's GLOBAL stress /
    tangent host arrays itself -- no ``map_to_sub`` / ``map_to_parent``, solver/maps.py:82-123) with every law's points cut
    into contiguous slices, one per device.  Device d keeps the history of ITS slices and a parent-sized committed / trial
    stress pair of which it only ever touches the rows of its slices (2 x 48 bytes per PARENT point and device: the indexed
    kernel addresses all stress arrays of a launch by parent row) and launches its laws on its own stream; one Python
    thread enqueues the launches of all devices (they are asynchronous: the devices work concurrently, each over its own
    PCIe link) and synchronises once.

    ``laws``: a list of ``(law, rows)`` as for ``ResidentProblemState`` (``rows`` = parent rows of the law's local points;
    one law with ``rows=None`` covers all points).  The global host arrays must be page-locked for every device:
    ``pin_host_arrays`` does it (one page lock, every device context enters the range).  Results are bit-identical to
    the one-device state (``tests/test_gpu_multidevice.py``)."""

    def __init__(self, laws, n_points: int, devices, del_t: float = 1.0, **kw):
        from .problem import ResidentProblemState

        if isinstance(laws, DeviceLaw):
            laws = [(laws, None)]
        self.devices = [int(d) for d in devices]
        assert self.devices, "no GPU to run on"
        self.n, world = int(n_points), len(self.devices)
        self._law_objs = [law for law, _ in laws]
        self._rows = [np.arange(self.n, dtype=np.int32) if rows is None else np.ascontiguousarray(rows, dtype=np.int32) for _, rows in laws]
        # slice [lo, hi) of every law's LOCAL points per device (tile-aligned: the rule of fcamd_shard_bounds)
        self._bounds = [[_capi.shard_bounds(r.size, world, d) for r in self._rows] for d in range(world)]
        kw.setdefault("placement", "torch")  # the tangent lives on the host: nothing to place
        self.states = [ResidentProblemState([(law, self._rows[k][lo:hi]) for k, (law, (lo, hi)) in enumerate(zip(self._law_objs, self._bounds[d]))],
                                            self.n, del_t=del_t, device=f"cuda:{dev}", **kw)
                       for d, dev in enumerate(self.devices)]
        self._pinned = []

    # -- the reference's names (solver/_solver.py:165-219), forwarded to every device's state -----------------
    @property
    def _time(self):
        return self.states[0]._time

    @_time.setter
    def _time(self, v):
        for s in self.states:
            s._time = v

    @property
    def _del_t(self):
        return self.states[0]._del_t

    @_del_t.setter
    def _del_t(self, v):
        for s in self.states:
            s._del_t = v

    def set_state(self, stress=None, history=None) -> None:
        """Committed state: parent stress (6 n, host) and one history dict per law (the law's LOCAL arrays)."""
        for d, s in enumerate(self.states):
            h = None
            if history is not None:
                h = []
                for k, hk in enumerate(history):
                    lo, hi = self._bounds[d][k]
                    dims = self._law_objs[k].history_dim
                    h.append(None if hk is None else {key: np.ascontiguousarray(hk[key][dims[key] * lo: dims[key] * hi]) for key in hk})
            s.set_state(stress, h)

    def pin_host_arrays(self, *arrays) -> None:
        """Page-lock the problem's global stress / tangent arrays and the laws' gradient arrays once, for every device."""
        seen = set()
        for dev in self.devices:
            if dev in seen:
                continue
            seen.add(dev)
            ctx = _capi.get_context(dev)
            for a in arrays:
                _check_numpy("array", a)
                ctx.register_host_buffer(a)  # the first context takes the page lock, the others enter the range
                self._pinned.append((ctx, a))

    def unpin_arrays(self) -> None:
        pinned, self._pinned = self._pinned, []
        for ctx, a in reversed(pinned):  # the owner of the page lock (first to register) last
            ctx.unregister_host_buffer(a)

    def evaluate_law_into(self, k: int, grad_del_u: np.ndarray, stress_parent: np.ndarray, tangent_parent, sync: bool = True) -> None:
        """Law ``k`` on every device: each device's slice of the law's local gradient in, its rows of the global arrays
        out.  ``sync=False`` returns after the launches (several laws in flight); the last call of a Newton iteration
        synchronises (``check``)."""
        _check_numpy("grad_del_u", grad_del_u)
        assert grad_del_u.size == 9 * self._rows[k].size, "grad_del_u has the wrong length"
        for d, s in enumerate(self.states):
            lo, hi = self._bounds[d][k]
            if hi > lo:
                s.evaluate_law_into(k, grad_del_u[9 * lo: 9 * hi], stress_parent, tangent_parent, sync=False)
        if sync:
            self.check()

    def check(self) -> None:
        """Synchronise every device and look at every law's counters; raises the reference's exceptions."""
        first = None
        for s in self.states:
            try:
                s.check()
            except RuntimeError as e:  # every device is synchronised before the first error is passed on
                first = first or e
        if first is not None:
            raise first

    def update(self) -> None:
        """Commit (``IncrSmallStrainProblem.update``): refused as a whole if any device's trial state is not fit."""
        self.check()
        for s in self.states:
            if s._evaluated:
                s.update()
            else:  # a device whose slices are all empty never evaluates: keep its clock in step
                s._time += s._del_t

    def download_history(self, k: int, history: dict, committed: bool = True) -> None:
        """Law ``k``'s history (its LOCAL arrays) -> the caller's NumPy arrays."""
        from .hostio import download

        dims = self._law_objs[k].history_dim
        for d, s in enumerate(self.states):
            lo, hi = self._bounds[d][k]
            if hi == lo:
                continue
            src = s.history_of(k, committed=committed)
            for key, arr in history.items():
                download(arr[dims[key] * lo: dims[key] * hi], src[key])

    def close(self) -> None:
        self.unpin_arrays()
        self.states = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
