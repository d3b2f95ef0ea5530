"""Attach device-resident state to an (unchanged) reference ``IncrSmallStrainProblem``.

``use_resident_state(problem)`` is the "patch of ``LawOnSubMesh.evaluate`` / ``update_history``"
that INTEGRATION.md describes, applied at run time to the problem's own objects: for every
``LawOnSubMesh`` (``solver/_lawonsubmesh.py:47-110``) whose law is one of this package's
GPU-backed laws it

* creates a ``ResidentState`` from the committed state the problem holds (the law's rows of
  ``problem.stress.previous`` through ``submesh_map.map_to_sub``; ``history.history_0``),
* replaces ``evaluate`` by: incremental gradient (unchanged dolfinx call) ->
  ``ResidentState.evaluate_into`` (gradient up, trial stress + tangent straight into the law's
  ``stress`` / ``local_tangent`` arrays) -> ``map_to_parent`` (unchanged),
* replaces ``update_history`` by the pointer-swap commit (optionally mirroring the committed
  history into the host ``Function``s once per increment, for post-processing code that reads them).

The host-side copies of every Newton iteration (``local_stress``: ``map_to_sub``;
``History.reset_trial_state``) and the history copy of every commit disappear; PCIe carries 72 + 336
bytes per point and iteration instead of 176 + 392.  Only attribute access is used, no dolfinx
import: anything that looks like the reference's dataclasses works (``tests/test_gpu_integration.py``
drives it with stand-ins).
"""

from __future__ import annotations

import types

import numpy as np

from .device import DeviceLaw
from .hostio import assign
from .problem import ResidentProblemState
from .resident import ResidentState

__all__ = ["use_resident_state", "use_resident_problem_state"]


def use_resident_state(problem, sync_history: bool = True, pin: bool = True, direct_global: bool = True, devices=None) -> list:
    """Returns the created states (one per GPU-backed law of ``problem._law_on_submeshs``).

    ``devices`` (list of device ordinals): the single-process multi-GPU mode -- every law's state is a
    ``MultiDeviceResidentState`` sliced over these GPUs, each of which reads its slice of the gradient from and writes
    its slice of stress and tangent to the problem's host arrays over its own PCIe link (no gather).

    ``direct_global``: a law whose submesh map is the reference's ``IdentityMap`` (one material on the
    whole mesh: ``map_to_parent`` is ``parent.x.array[:] = sub.x.array[:]``, solver/maps.py:29-47) writes
    its stress and tangent straight into the problem's global arrays -- the host copy of 336 B per point
    and Newton iteration disappears as well (the law's local ``stress`` / ``local_tangent`` Functions are
    then no longer updated)."""
    states = []
    for los in problem._law_on_submeshs:
        law = los.law
        if not isinstance(law, DeviceLaw):
            continue
        sd = law.stress_strain_dim
        n = los.stress.x.array.size // sd
        stress0 = los.local_stress(problem.stress).copy()  # committed stress of this law's cells
        hist0 = None if los.history is None else {k: f.x.array for k, f in los.history.history_0.items()}
        if devices is not None:
            from .multidevice import MultiDeviceResidentState

            state = MultiDeviceResidentState(law, n, devices=devices, stress0=stress0,
                                             history0=None if hist0 is None else {k: np.ascontiguousarray(v) for k, v in hist0.items()})
        else:
            state = ResidentState(law, n, stress0=stress0, history0=hist0)
        direct = bool(direct_global) and type(los.submesh_map).__name__ == "IdentityMap"
        if pin:
            pinner = state if devices is not None else law  # one page lock for all devices of a multi-device state
            if direct:
                pinner.pin_host_arrays(los.displacement_gradient_fn.x.array, problem.stress.current.x.array,
                                       problem.tangent.x.array)
            else:
                pinner.pin_host_arrays(los.displacement_gradient_fn.x.array, los.stress.x.array, los.local_tangent.x.array)

        def evaluate(self, sim_time, incr_disp, global_stress, global_tangent, _state=state, _direct=direct):
            incr_disp.evaluate_local_incremental_gradient(self.cells, self.displacement_gradient_fn)
            if _direct:
                _state.evaluate_into(sim_time.current, sim_time.dt, self.displacement_gradient_fn.x.array,
                                     global_stress.current.x.array, global_tangent.x.array)
                global_stress.current.x.scatter_forward()  # what IdentityMap.map_to_parent does after its copy
                global_tangent.x.scatter_forward()
                return
            _state.evaluate_into(sim_time.current, sim_time.dt, self.displacement_gradient_fn.x.array,
                                 self.stress.x.array, self.local_tangent.x.array)
            self.map_to_parent(global_stress, global_tangent)

        def update_history(self, _state=state, _sync=sync_history, _multi=devices is not None):
            _state.update()
            if _sync and self.history is not None:
                if _multi:  # every device's slice straight into the problem's history_0 arrays
                    _state.download(history={key: fn.x.array for key, fn in self.history.history_0.items()}, committed=True)
                else:
                    committed = _state.history_committed
                    for key, fn in self.history.history_0.items():
                        assign(fn.x.array, committed[key])
                for key, fn in self.history.history_0.items():
                    self.history.history_1[key].x.array[:] = fn.x.array

        los.evaluate = types.MethodType(evaluate, los)
        los.update_history = types.MethodType(update_history, los)
        los.resident_state = state
        states.append(state)
    return states


def _parent_rows(los, n_local: int):
    """Parent row of every local quadrature point of a law: ``parent_array[map.parent] = sub_array[map.sub]``
    (solver/maps.py:98-100) read as local point ``sub[i]`` -> parent row ``parent[i]``; ``None`` for the
    reference's ``IdentityMap``."""
    m = los.submesh_map
    if type(m).__name__ == "IdentityMap":
        return None
    rows = np.empty(n_local, dtype=np.int64)
    rows[np.asarray(m.sub)] = np.asarray(m.parent)
    return rows


def use_resident_problem_state(problem, sync_history: bool = True, pin: bool = True, devices=None):
    """Multi-material form of ``use_resident_state``: ONE ``ResidentProblemState`` for all GPU-backed (FULL
    3-D) laws of ``problem`` -- the committed / trial stress of the whole mesh and every law's history on the
    GPU -- and every ``LawOnSubMesh.evaluate`` replaced by: incremental gradient (unchanged dolfinx call) ->
    ``ResidentProblemState.evaluate_law_into``, whose kernel reads the law's gradient from the host array and
    writes the law's rows of the problem's GLOBAL ``stress.current`` / ``tangent`` arrays itself.  Besides
    the state copies this removes both submesh maps of every Newton iteration (``local_stress``:
    ``map_to_sub``; ``map_to_parent``: fancy-indexed host copies of 48 + 336 bytes per point,
    solver/maps.py:82-123) and the per-law local stress / tangent arrays.  The laws of one iteration are
    launched back to back and synchronised once, after the last one.  ``update_history`` of the first law
    commits the shared state (pointer swap); laws that are not GPU-backed keep the reference's path.

    ``devices`` (list of device ordinals): the same flow with ONE process driving several GPUs -- every law's points are
    cut into one contiguous slice per device (``multidevice.MultiDeviceProblemState``), each device writes its rows of
    the global host arrays over its own PCIe link."""
    gpu = [(i, los) for i, los in enumerate(problem._law_on_submeshs) if isinstance(los.law, DeviceLaw)]
    assert gpu, "no GPU-backed law in this problem"
    assert all(los.law.constraint.name == "FULL" for _, los in gpu), "use_resident_problem_state: FULL 3-D laws only"
    n = problem.stress.current.x.array.size // 6
    laws = []
    for _, los in gpu:
        n_k = los.displacement_gradient_fn.x.array.size // 9
        laws.append((los.law, _parent_rows(los, n_k)))
    single_identity = len(laws) == 1 and laws[0][1] is None
    if devices is not None:
        from .multidevice import MultiDeviceProblemState

        state = MultiDeviceProblemState(laws, n, devices, del_t=problem.sim_time.dt)
    else:
        state = ResidentProblemState(laws[0][0] if single_identity else laws, n, del_t=problem.sim_time.dt)
    state.set_state(problem.stress.previous.x.array,
                    [None if los.history is None else {k: f.x.array for k, f in los.history.history_0.items()}
                     for _, los in gpu])
    if pin and devices is not None:  # one page lock, entered by every device's context
        state.pin_host_arrays(problem.stress.current.x.array, problem.tangent.x.array,
                              *[los.displacement_gradient_fn.x.array for _, los in gpu])
    elif pin:
        first = gpu[0][1].law
        first.pin_host_arrays(problem.stress.current.x.array, problem.tangent.x.array)
        for _, los in gpu:
            los.law.pin_host_arrays(los.displacement_gradient_fn.x.array)
    last = len(gpu) - 1

    for k, (_, los) in enumerate(gpu):
        def evaluate(self, sim_time, incr_disp, global_stress, global_tangent, _k=k):
            incr_disp.evaluate_local_incremental_gradient(self.cells, self.displacement_gradient_fn)
            state._time, state._del_t = sim_time.current, sim_time.dt
            state.evaluate_law_into(_k, self.displacement_gradient_fn.x.array, global_stress.current.x.array,
                                    global_tangent.x.array, sync=(_k == last))
            if _k == last:
                global_stress.current.x.scatter_forward()  # as the reference's map_to_parent does per law
                global_tangent.x.scatter_forward()

        def update_history(self, _k=k):
            if _k == 0:
                state.update()
            if sync_history and self.history is not None:
                if devices is not None:  # every device's slice straight into the problem's history_0 arrays
                    state.download_history(_k, {key: fn.x.array for key, fn in self.history.history_0.items()})
                else:
                    committed = state.history_of(_k, committed=True)
                    for key, fn in self.history.history_0.items():
                        assign(fn.x.array, committed[key])
                for key, fn in self.history.history_0.items():
                    self.history.history_1[key].x.array[:] = fn.x.array

        los.evaluate = types.MethodType(evaluate, los)
        los.update_history = types.MethodType(update_history, los)
    problem.resident_problem_state = state
    return state
