"""Device-resident counterpart of the state that ``IncrSmallStrainProblem`` keeps around the hot
path (SURVEY 8f-1 + 8f-2 together): several laws on disjoint sets of quadrature points of one mesh.

What the reference does on the host for every Newton iteration (``solver/_solver.py:130-147``,
``solver/_lawonsubmesh.py:72-95``): for each law, gather the committed stress of its cells from the
global array (``SubSpaceMap.map_to_sub``), reset its trial history, evaluate, scatter stress and
tangent back into the global arrays (``map_to_parent``, ``solver/maps.py:82-123``); and on
``update()`` (``solver/_solver.py:149-159``) copy trial -> committed for the stress and every
history.  Here the global ("parent") stress pair and tangent and every law's history pair live on
the GPU; one fused launch per law does gather + evaluate + scatter
(``fcamd_evaluate_device_indexed``), and the commit swaps pointers.

Names follow the reference's backward-compatibility properties (``stress_0``, ``stress_1``,
``_history_0``, ``_history_1``, ``_time``, ``_del_t``; ``solver/_solver.py:165-219``).
"""

from __future__ import annotations


import numpy as np

from .device import DeviceLaw, _is_torch
from .hostio import assign, to_device, to_host, upload


def _store(dst, src) -> None:
    """``dst`` (device tensor) <- ``src`` (device tensor or NumPy array)"""
    if _is_torch(src):
        dst.copy_(src)
    else:
        upload(dst, np.ascontiguousarray(src, dtype=np.float64))

__all__ = ["ResidentProblemState", "rows_of_cells"]


def rows_of_cells(cells: np.ndarray, points_per_cell: int) -> np.ndarray:
    """Quadrature-point rows of a set of cells: the dofs of the reference's quadrature spaces are
    numbered cell by cell (``solver/maps.py:43-79`` builds exactly this map)."""
    cells = np.asarray(cells, dtype=np.int64)
    return (cells[:, None] * points_per_cell + np.arange(points_per_cell)[None, :]).reshape(-1).astype(np.int32)


class _LawState:
    def __init__(self, law, rows, n, f, device, sparse_history, packed_history=True):
        import torch

        self.law, self.n = law, n
        self.rows = None if rows is None else to_device(rows, device, np.int32)
        hd = law.history_dim
        self.hist = None if hd is None else [{k: torch.zeros(d * n, **f) for k, d in hd.items()} for _ in range(2)]
        self.grad = None  # staging buffer for NumPy gradients
        # LE / SLS: the law's tangent rows are point-independent and change only with del_t; the
        # parent tangent array is owned by the problem state, so they are written once per del_t
        # (see ResidentState)
        self.const_tangent = type(law).__name__ in ("LinearElasticityModel", "LinearElasticity3D",
                                                    "SpringMaxwellModel", "SpringKelvinModel")
        self.tangent_key = None
        # plasticity laws: sparse trial history (see ResidentState); the mask survives the pointer swap
        self.mask = None
        if sparse_history and type(law).__name__ in ("VonMises3D", "MisesPlasticityLinearHardening3D",
                                                     "DruckerPrager3D", "DruckerPragerHyperbolic3D"):
            self.mask = torch.zeros((n + 63) // 64, dtype=torch.int64, device=device)
        # packed plastic-strain history (see ResidentState; FCAMD_EVAL_PACKED_HISTORY): VonMises3D's eps_n -- both copies packed per
        # tile, one EVER mask per copy; ``history_view`` hands the reference's layout out (a copy)
        self.packed = bool(packed_history) and self.mask is not None and type(law).__name__ == "VonMises3D"
        self.ever = [torch.zeros_like(self.mask), torch.zeros_like(self.mask)] if self.packed else None
        # sparse tangent (see ResidentState): the array that received this law's previous tangent --
        # "dev" (the state's device array) or the address of the host assembler's parent array
        self.tangent_target = None
        self.host_tangent_key = None  # constant-tangent laws: (host address, bytes, del_t) the host array is valid for
        # the state's own counters of this law's launches (see ResidentState): read before every commit
        from .device import new_counters

        self.counters = new_counters(device) if self.mask is not None or type(law).__name__ in (
            "VonMises3D", "MisesPlasticityLinearHardening3D", "DruckerPrager3D", "DruckerPragerHyperbolic3D") else None
        self.stats_pending = False
        self.failed = None  # the error of THIS law's last evaluate (cleared only when this law is evaluated again)

    def packed_masks(self, c):
        """(EVER mask of the committed copy ``c``, of the trial copy) for the launches, or None"""
        return (self.ever[c], self.ever[1 - c]) if self.packed else None

    def history_view(self, copy):
        """this law's history of one copy in the reference's layout (the packed plastic-strain array unpacked: a new tensor)"""
        if self.hist is None:
            return None
        if not self.packed:
            return self.hist[copy]
        from .device import unpack_rows

        h = self.hist[copy]
        return {**h, "eps_n": unpack_rows(h["eps_n"], self.ever[copy], self.n)}

    def store_history(self, c, h):
        """committed copy ``c`` <- ``h`` (reference layout; NumPy or device), the trial copy set equal to it (sparse-history contract)"""
        from .device import pack_rows

        for k in self.hist[c]:
            _store(self.hist[c][k], h[k])
            if self.packed and k == "eps_n":
                packed, ever = pack_rows(self.hist[c][k].clone())
                self.hist[c][k].copy_(packed)
                self.ever[c].copy_(ever)
                self.ever[1 - c].copy_(ever)
            self.hist[1 - c][k].copy_(self.hist[c][k])


class ResidentProblemState:
    """``laws``: one ``DeviceLaw`` (covers all ``n_points``) or a list of ``(law, rows)`` with
    ``rows`` the quadrature-point rows of that law in the parent arrays (disjoint; see
    ``rows_of_cells``).  All laws are FULL 3-D, as the fused indexed kernel requires."""

    #: device-assembler mode: tune the placement of the parent tangent array on the first ``evaluate`` when it
    #: is at least this large (see ResidentState)
    AUTO_TUNE_MIN_BYTES = 256 << 20

    def __init__(self, laws, n_points: int, del_t: float = 1.0, device=None, reuse_constant_tangent: bool = True,
                 sparse_history: bool = True, sparse_tangent: bool = True, placement: str = "auto", packed_history: bool = True,
                 batch_launches: bool = True):
        import torch

        from . import _capi

        self.device = torch.device("cuda", _capi.default_device()) if device is None else torch.device(device)
        self.n = int(n_points)
        f = dict(dtype=torch.float64, device=self.device)
        self._f = f
        if isinstance(laws, DeviceLaw):
            laws = [(laws, None)]
        self._laws = []
        covered = np.zeros(self.n, dtype=np.int64)
        for law, rows in laws:
            assert law.constraint.name == "FULL", "ResidentProblemState: FULL 3-D laws only"
            if rows is None:
                assert len(laws) == 1, "rows=None (all points) is for a single law"
                n_k = self.n
                covered += 1
            else:
                rows = np.asarray(rows)
                n_k = rows.size
                from .maps import warn_if_rows_not_ascending

                warn_if_rows_not_ascending(rows, "ResidentProblemState")
                assert rows.min(initial=0) >= 0 and rows.max(initial=-1) < self.n, "row out of range"
                np.add.at(covered, rows, 1)
            self._laws.append(_LawState(law, rows, n_k, f, self.device, sparse_history, packed_history))
        assert covered.max(initial=0) <= 1, "a quadrature point belongs to more than one law"
        self._stress = [torch.zeros(6 * self.n, **f), torch.zeros(6 * self.n, **f)]
        self._tangent = None  # parent tangent on the device: allocated on first use (device-assembler mode only)
        self._c = 0
        self._time, self._del_t = 0.0, float(del_t)
        self._evaluated = False
        self.reuse_constant_tangent = reuse_constant_tangent
        self.sparse_tangent = sparse_tangent
        #: the laws of one ``evaluate`` leave as ONE ``fcamd_evaluate_batch`` (False: one ``fcamd_evaluate_device_ex`` per law)
        self.batch_launches = batch_launches
        self._launch_cache = _capi.LaunchCache()  # committed copy -> the kept argument arrays of the call (_capi.PreparedBatch)
        # the error of a law's last evaluate, if it raised, lives with the law (_LawState.failed): nothing to commit
        # placement of the arrays the launches stream (see ResidentState): "auto" / "vmm" move the parent
        # stress pair, the parent tangent and every law's history pair into one interleaved VMM working set on
        # the first device-assembler evaluate; "tune" times candidate allocations of the tangent; "torch": none
        assert placement in ("auto", "vmm", "tune", "torch")
        self._placement_mode = placement
        self._placed = placement == "torch"
        self._vmm = None
        self.placement = None

    @property
    def tangent(self):
        """Parent tangent array on the device (rows of points that belong to no law stay zero)."""
        if self._tangent is None:
            import torch

            self._tangent = torch.zeros(36 * self.n, **self._f)
        return self._tangent

    @tangent.setter
    def tangent(self, value):
        self._tangent = value

    # reference-compatible views ----------------------------------------------------------------------
    @property
    def stress_0(self):
        """Committed stress (``IncrementalStress.previous``)."""
        return self._stress[self._c]

    @property
    def stress_1(self):
        """Trial stress (``IncrementalStress.current``)."""
        return self._stress[1 - self._c]

    def history_of(self, k: int, committed: bool = True):
        """History of law ``k`` alone in the reference's layout -- committed, or trial (before the first evaluate of an increment
        the trial state IS the committed one).  Live tensors, except VonMises3D's packed ``eps_n``, which is unpacked into a new
        tensor: ask for the law you need (``_history_0`` / ``_history_1`` unpack EVERY packed law on each access)."""
        ls = self._laws[k]
        if committed:
            return ls.history_view(self._c)
        return ls.history_view(1 - self._c if (self._evaluated or not ls.packed) else self._c)

    @property
    def _history_0(self):
        """committed history per law, reference layout (live tensors; VonMises3D's packed ``eps_n``: a copy) -- ``history_of(k)`` for one law"""
        return [self.history_of(k, True) for k in range(len(self._laws))]

    @property
    def _history_1(self):
        """trial history per law -- ``history_of(k, committed=False)`` for one law"""
        return [self.history_of(k, False) for k in range(len(self._laws))]

    def set_state(self, stress=None, history=None) -> None:
        """Initial committed state: parent stress (6 n) and a list of per-law history dicts."""
        import torch

        if stress is not None:
            _store(self.stress_0, stress)
        if history is not None:
            for ls, h in zip(self._laws, history):
                if ls.hist is None:
                    continue
                ls.store_history(self._c, h)  # trial == committed (sparse-history contract)
                if ls.mask is not None:
                    ls.mask.zero_()
                ls.tangent_target = None  # the mask no longer remembers which rows hold plastic tangents
        for ls in self._laws:
            ls.stats_pending = False
            ls.failed = None

    # the Newton-iteration call (IncrSmallStrainProblem.form, solver/_solver.py:130-147) ------------------
    def evaluate(self, grads) -> None:
        """``grads``: one local ``grad_del_u`` array per law (9 n_k; NumPy or device tensor), what
        ``evaluate_local_incremental_gradient`` fills for the law's cells."""
        import torch

        if not isinstance(grads, (list, tuple)):
            grads = [grads]
        assert len(grads) == len(self._laws), "one gradient array per law"
        if not self._placed:
            self._placed = True
            if 8 * 36 * self.n >= self.AUTO_TUNE_MIN_BYTES:
                self._place(grads)
        for ls in self._laws:
            ls.failed = None
        from . import _capi

        staged = []
        for ls, g in zip(self._laws, grads):
            if not _is_torch(g):
                if ls.grad is None:
                    ls.grad = torch.empty(9 * ls.n, **self._f)
                upload(ls.grad, np.ascontiguousarray(g, dtype=np.float64))  # synchronous: the caller may free or rewrite g on return
                g = ls.grad
            staged.append(g)
        # what every law's launch takes this time (the constant-tangent and sparse-tangent shortcuts are state of the law)
        plan = []
        for ls, g in zip(self._laws, staged):
            tangent, key = self.tangent, None
            if ls.const_tangent and self.reuse_constant_tangent:
                key = self._del_t if type(ls.law).__name__.startswith("Spring") else 0.0
                if ls.tangent_key == key:
                    tangent = None
                ls.tangent_key = None  # valid again only once the launch below has been enqueued
            st = self.sparse_tangent and ls.mask is not None and ls.tangent_target == "dev"
            ls.tangent_target = None
            plan.append((g, tangent, st, key))
        # The laws of this form() leave as ONE fcamd_evaluate_batch (they write disjoint rows of the shared stress / tangent arrays):
        # one trip through the binding, the small laws concurrently.  Between the Newton iterations of an increment nothing but the
        # gradients' VALUES changes, so the argument arrays of the call are kept and issued again while every pointer is the same.
        if self.batch_launches:
            from .device import _current_stream_ptr

            sig = (self._del_t, self.stress_0.data_ptr(), self.stress_1.data_ptr(),
                   tuple((g.data_ptr(), g.numel(), 0 if tan is None else tan.data_ptr(), st,
                          0 if ls.hist is None else next(iter(ls.hist[0].values())).data_ptr(),
                          0 if ls.hist is None else next(iter(ls.hist[1].values())).data_ptr(),
                          0 if ls.mask is None else ls.mask.data_ptr()) for ls, (g, tan, st, _) in zip(self._laws, plan)))
            self._launch_cache.run(self._c, sig, lambda: self._enqueue_laws(plan), _current_stream_ptr(self.device.index or 0))
        else:
            self._enqueue_laws(plan)
        for ls, (_, _, _, key) in zip(self._laws, plan):
            ls.tangent_key = key
            ls.tangent_target = "dev"
            ls.stats_pending = ls.counters is not None
        self._evaluated = True

    def _enqueue_laws(self, plan) -> None:
        for ls, (g, tangent, st, _) in zip(self._laws, plan):
            hp = None if ls.hist is None else ls.hist[self._c]
            hc = None if ls.hist is None else ls.hist[1 - self._c]
            if ls.rows is None:
                ls.law.evaluate_from(self._time, self._del_t, g, self.stress_0, self.stress_1, tangent, hp, hc,
                                     history_mask=ls.mask, sparse_tangent=st, counters=ls.counters, packed_masks=ls.packed_masks(self._c))
            else:
                ls.law.evaluate_indexed(self._time, self._del_t, g, self.stress_0, self.stress_1, tangent,
                                        ls.rows, hp, hc, history_mask=ls.mask, sparse_tangent=st, counters=ls.counters,
                                        packed_masks=ls.packed_masks(self._c))

    # the host assembler's Newton-iteration call, law by law (LawOnSubMesh.evaluate, solver/_lawonsubmesh.py:72-95)
    def evaluate_law_into(self, k: int, grad_del_u: np.ndarray, stress_parent: np.ndarray,
                          tangent_parent: np.ndarray | None, sync: bool = True) -> None:
        """Law ``k`` only: trial state <- law(committed state, ``grad_del_u``) with the law's LOCAL gradient
        as a NumPy array, and the law's rows of the host assembler's PARENT arrays ``stress_parent`` (6 n)
        / ``tangent_parent`` (36 n; optional) written by the kernel itself, over PCIe -- the reference's
        ``map_to_sub`` / ``map_to_parent`` copies (solver/maps.py:82-123) and its per-law local stress and
        tangent arrays have no counterpart.  The two parent arrays must be page-locked
        (``law.pin_host_arrays`` / ``Context.register_host_buffer``) and are taken to be left alone between
        calls, as in ``ResidentState.evaluate_into``: constant tangents are written once per ``del_t``,
        point-dependent ones row by row (sparse tangent).  A page-locked gradient array is read in place,
        a pageable one is uploaded first.  ``sync=False`` returns after the launch (several laws in flight;
        the last call of a Newton iteration must synchronise before the host reads the arrays)."""
        import torch

        from . import _capi
        from .device import _check_numpy, _current_stream_ptr

        ls = self._laws[k]
        _check_numpy("grad_del_u", grad_del_u), _check_numpy("stress_parent", stress_parent)
        assert grad_del_u.size == 9 * ls.n, "grad_del_u has the wrong length"
        assert stress_parent.size == 6 * self.n, "stress_parent has the wrong length"
        dev = self.device.index or 0
        m = ls.law._handle(dev)
        ctx = m.ctx
        ctx.set_stream(_current_stream_ptr(dev))
        if tangent_parent is not None:
            _check_numpy("tangent_parent", tangent_parent)
            assert tangent_parent.size == 36 * self.n, "tangent_parent has the wrong length"
        try:
            sptr = ctx.device_pointer(stress_parent)
            tptr = None if tangent_parent is None else ctx.device_pointer(tangent_parent)
        except ValueError:
            # parent arrays that are not page-locked (e.g. too small to be worth pinning): evaluate on the
            # device arrays and copy this law's rows down -- correct for any array, meant for small problems
            self._evaluate_law_staged(k, grad_del_u, stress_parent, tangent_parent)
            return
        try:
            gptr = ctx.device_pointer(grad_del_u)
        except ValueError:  # pageable gradient: upload
            if ls.grad is None:
                ls.grad = torch.empty(9 * ls.n, **self._f)
            upload(ls.grad, grad_del_u)
            gptr = ls.grad.data_ptr()
        flags, target = 0, None if tptr is None else ("host", tptr, tangent_parent.nbytes)
        key = None
        if tptr is not None and ls.const_tangent and self.reuse_constant_tangent:
            key = (tptr, tangent_parent.nbytes, self._del_t if type(ls.law).__name__.startswith("Spring") else 0.0)
            if ls.host_tangent_key == key:
                tptr = None  # the parent array already holds this law's rows
            ls.host_tangent_key = None  # valid again only once the launch below has been enqueued
        elif tptr is not None and self.sparse_tangent and ls.mask is not None and ls.tangent_target == target:
            flags = _capi.EVAL_SPARSE_TANGENT
        hp = [] if ls.hist is None else [ls.hist[self._c][name].data_ptr() for name, _ in m.history_fields]
        hc = [] if ls.hist is None else [ls.hist[1 - self._c][name].data_ptr() for name, _ in m.history_fields]
        ls.tangent_target = None
        ls.failed = None  # this law's record only: another law's failure of the same iteration stays on the books
        self._evaluated = True  # the trial state is touched even if the launch fails
        pm = ls.packed_masks(self._c)
        if pm is not None:
            flags |= _capi.EVAL_PACKED_HISTORY
        m.evaluate_device_ex(self._time, self._del_t, ls.n, gptr, self.stress_0.data_ptr(), self.stress_1.data_ptr(), tptr,
                             hp, hc, None if ls.rows is None else ls.rows.data_ptr(),
                             None if ls.mask is None else ls.mask.data_ptr(), flags, stress2_ptr=sptr,
                             counters_ptr=None if ls.counters is None else ls.counters.data_ptr(),
                             packed_mask_ptrs=None if pm is None else (pm[0].data_ptr(), pm[1].data_ptr()))
        ls.host_tangent_key = key
        ls.tangent_target = target
        ls.stats_pending = ls.counters is not None
        # the arrays of every law are identified by address above: keep them alive
        self.__dict__.setdefault("_host_refs", {})[k] = (stress_parent, tangent_parent)
        if sync:
            self.check()

    def _evaluate_law_staged(self, k, grad_del_u, stress_parent, tangent_parent) -> None:
        import torch

        ls = self._laws[k]
        if ls.grad is None:
            ls.grad = torch.empty(9 * ls.n, **self._f)
        upload(ls.grad, grad_del_u)
        hp = None if ls.hist is None else ls.hist[self._c]
        hc = None if ls.hist is None else ls.hist[1 - self._c]
        ls.tangent_key = ls.host_tangent_key = None
        ls.tangent_target = None
        ls.failed = None
        tan = None if tangent_parent is None else self.tangent
        ls.stats_pending = ls.counters is not None
        if ls.rows is None:
            ls.law.evaluate_from(self._time, self._del_t, ls.grad, self.stress_0, self.stress_1, tan, hp, hc,
                                 history_mask=ls.mask, counters=ls.counters, packed_masks=ls.packed_masks(self._c))
            assign(stress_parent, self.stress_1)
            if tan is not None:
                assign(tangent_parent, tan)
        else:
            ls.law.evaluate_indexed(self._time, self._del_t, ls.grad, self.stress_0, self.stress_1, tan, ls.rows, hp, hc,
                                    history_mask=ls.mask, counters=ls.counters, packed_masks=ls.packed_masks(self._c))
            rows = ls.rows.long()
            rows_h = to_host(rows)
            stress_parent.reshape(-1, 6)[rows_h] = to_host(self.stress_1.view(-1, 6)[rows])
            if tan is not None:
                tangent_parent.reshape(-1, 36)[rows_h] = to_host(tan.view(-1, 36)[rows])
        self._evaluated = True
        self.check()

    def _place(self, grads) -> None:
        """See ResidentState._place: "tune", "vmm", or ("auto") the faster of the two."""
        mode, best_ms = self._placement_mode, None
        if mode in ("auto", "tune") and not (self.reuse_constant_tangent and all(ls.const_tangent for ls in self._laws)):
            info = self.tune_placement(grads, tries=6)
            best_ms = min(info["candidate_ms"])
            self.placement = {"mode": "hipmalloc_tuned", **info}
        if mode in ("auto", "vmm"):
            saved = (self._stress, [ls.hist for ls in self._laws], self._tangent)

            def restore():
                self._stress, hists, self._tangent = saved
                for ls, h in zip(self._laws, hists):
                    ls.hist = h
                self._vmm = None

            try:
                self._move_to_vmm()
                if best_ms is not None:
                    vmm_ms = self._time_evaluate(grads)
                    self.placement.update({"vmm_ms": round(vmm_ms, 4), "hipmalloc_best_ms": round(best_ms, 4)})
                    if vmm_ms >= best_ms:  # the tuned hipMalloc arrays win: back to them
                        restore()
                        self.placement["mode"] = "hipmalloc_tuned"
            except Exception as e:  # no VMM support / no room for the move: keep what there is
                if mode == "vmm":
                    raise
                restore()
                self.placement = {**(self.placement or {"mode": "torch"}), "vmm_error": f"{type(e).__name__}: {e}"[:200]}
            for ls in self._laws:
                ls.tangent_key = ls.tangent_target = None  # whichever tangent array it is: written in full next

    def _time_evaluate(self, grads, launches: int = 3) -> float:
        import torch

        def full():
            for ls in self._laws:
                ls.tangent_key = ls.tangent_target = None
            self.evaluate(grads)

        full()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
        for x, y in ev:
            x.record()
            full()
            y.record()
        torch.cuda.synchronize(self.device)
        return min(x.elapsed_time(y) for x, y in ev)

    def _move_to_vmm(self) -> None:
        from . import _capi
        from .placement import VmmArraySet

        numels = {"tangent": 36 * self.n, "stress0": 6 * self.n, "stress1": 6 * self.n}
        for i, ls in enumerate(self._laws):
            if ls.hist is not None:
                for k, d in ls.law.history_dim.items():
                    numels[f"law{i}_h0_{k}"] = d * ls.n
                    numels[f"law{i}_h1_{k}"] = d * ls.n
        numels = {k: v for k, v in numels.items() if v > 0}
        vmm = VmmArraySet(_capi.get_context(self.device.index or 0), numels, interleaved=True, device=self.device)
        new_stress = [vmm["stress0"], vmm["stress1"]]
        for i in (0, 1):
            new_stress[i].copy_(self._stress[i])
        self._stress = new_stress
        for i, ls in enumerate(self._laws):
            if ls.hist is None or ls.n == 0:
                continue
            new_hist = [{k: vmm[f"law{i}_h{c}_{k}"] for k in ls.hist[c]} for c in (0, 1)]
            for c in (0, 1):
                for k in new_hist[c]:
                    new_hist[c][k].copy_(ls.hist[c][k])
            ls.hist = new_hist
        old = self._tangent
        self._tangent = vmm["tangent"]
        if old is not None:
            self._tangent.copy_(old)
        else:
            self._tangent.zero_()  # rows of points that belong to no law stay zero, as in a freshly built state
        self._vmm = vmm
        self.placement = {**(self.placement or {}), "mode": "vmm_interleaved", "arrays": len(numels),
                          "GB": round(8 * sum(numels.values()) / 1e9, 2)}

    def tune_placement(self, grads, tries: int = 4) -> dict:
        """Choose the placement of the parent tangent array (the dominant write stream of every law's
        launch) by timing one full ``evaluate(grads)`` on a few candidate allocations and keeping the
        fastest (``placement.fastest_allocation``).  Call once before the Newton loops;
        leaves a valid trial state for ``grads``."""
        from .placement import fastest_allocation

        self._placed = True

        def probe(tan):
            self.tangent = tan
            for ls in self._laws:
                ls.tangent_key = None  # constant tangents have to be written into the candidate
                ls.tangent_target = None  # ... and every row of the point-dependent ones
            self.evaluate(grads)

        first, self.tangent = self.tangent, None
        chosen, info = fastest_allocation(36 * self.n, probe, tries=tries, device=self.device, first=first)
        del first
        chosen.zero_()  # rows of points that belong to no law stay zero, as in a freshly built state
        probe(chosen)
        if self.placement is None:
            self.placement = {"mode": "hipmalloc_tuned", **info}
        return info

    def check(self) -> None:
        """Synchronise with the last launches and look at every law's counters (read once per evaluate);
        raises the reference's exceptions (Newton non-convergence, Drucker-Prager tip) per law."""
        import torch

        from .device import read_counters

        # Always a synchronisation point: the host assembler reads the arrays the launches wrote (and may free
        # or unpin them) right after check() -- also when no law of the problem has counters to read.
        torch.cuda.current_stream(self.device).synchronize()
        for ls in self._laws:
            if not ls.stats_pending:
                continue
            ls.law.last_stats = st = read_counters(ls.counters)
            ls.stats_pending = False
            try:
                ls.law.raise_for_stats(st)
            except RuntimeError as e:
                ls.failed = e
        # a law whose last evaluate failed keeps the trial state uncommittable until THAT law has been evaluated again --
        # whatever the other laws did in between (a caller that catches law k's error and carries on with law k + 1)
        if self._failed is not None:
            raise self._failed

    @property
    def _failed(self):
        return next((ls.failed for ls in self._laws if ls.failed is not None), None)

    # the commit (IncrSmallStrainProblem.update, solver/_solver.py:149-159) -------------------------------
    def update(self) -> None:
        if not self._evaluated:
            raise RuntimeError("update() before any evaluate() of this increment")
        self.check()  # a trial state with a non-converged point raises the reference's error instead of being committed
        self._c = 1 - self._c  # stress and every history: trial becomes committed
        self._time += self._del_t
        self._evaluated = False

    def download(self, stress: np.ndarray | None = None, tangent: np.ndarray | None = None) -> None:
        if stress is not None:
            assign(stress, self.stress_1)
        if tangent is not None:
            assign(tangent, self.tangent)
