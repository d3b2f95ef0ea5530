"""Device-resident increment state.

The reference keeps two copies of stress and history (previous = committed, current = trial)
and, around every ``evaluate`` of every Newton iteration, copies committed -> trial on the host
(``solver/_lawonsubmesh.py:58-61``, ``solver/_history.py:64-79``) and, on ``update()``, trial ->
committed (``solver/_history.py:68-88``, ``solver/_incrementalunknowns.py:70-72``).  Between the
iterations of one increment only ``grad_del_u`` changes.

``ResidentState`` holds both copies on the GPU:

* ``evaluate(t, del_t, grad)``: one out-of-place launch (committed -> trial,
  ``fcamd_evaluate_device_from``); only ``grad`` crosses PCIe when it is a NumPy array;
* ``update()``: the commit is a pointer swap -- no copy at all;
* ``stress`` / ``tangent`` / ``history`` expose the trial state as device tensors (for a device
  assembler);
* ``evaluate_into(t, del_t, grad, stress_out, tangent_out)`` is the host assembler's call
  (dolfinx): one chunk-pipelined pass (``fcamd_evaluate_resident``) uploads ``grad``, evaluates on
  the resident state and streams stress and tangent into the caller's NumPy arrays while later
  chunks are still in flight; no n-sized gradient or tangent array exists on the device.

Per Newton iteration this moves 72 B/pt up and (if the host assembles) 336 B/pt down instead of
176 + 392 B/pt, and removes every host-side state copy.
"""

from __future__ import annotations

import numpy as np

from .device import DeviceLaw, _current_stream_ptr, _is_torch
from .hostio import assign, to_device, upload

__all__ = ["ResidentState"]


class ResidentState:
    #: device-assembler mode: place the state's arrays on the first ``evaluate`` when the tangent is at
    #: least this large (below, launches are latency-bound and the placement does not show)
    AUTO_TUNE_MIN_BYTES = 256 << 20

    def __init__(self, law: DeviceLaw, n: int, device=None, stress0=None, history0=None, sparse_history: bool = True,
                 reuse_constant_tangent: bool = True, sparse_tangent: bool = True, placement: str = "auto",
                 split_history: bool = True, packed_history: bool = True):
        """``placement`` (device-assembler mode, large states): where the arrays the kernel streams
        live decides 10-28 % of its time on MI355X.  "vmm": on the first ``evaluate`` the state moves its arrays
        (both stress / history copies, tangent, gradient staging) into ONE working set whose 2 MiB physical
        handles are interleaved over the arrays (``placement.VmmArraySet``) -- reproducible to ~2 %, at the
        level of the better hipMalloc draws; "tune": time a few candidate allocations of the tangent with the
        state's own launch and keep the fastest (``tune_placement``); "torch": take what the allocator gives;
        "auto" (default): "vmm", falling back to "tune" if the virtual-memory API is not available."""
        import torch

        assert placement in ("auto", "vmm", "tune", "torch")
        # ``packed_history`` (VonMises3D and -- with ``split_history`` -- the comfe-rs plasticity laws, under the sparse
        # protocol): the plastic-strain array only accumulates (mises_plasticity_isotropic_hardening.py:161,
        # mises_plasticity.rs:112, general.rs:243) and is +0.0 wherever a point has never been plastic, so BOTH copies of it
        # are kept packed per 64-point tile -- the rows that are not all +0.0 as one contiguous run at the head of the
        # tile's slot, plus one EVER-mask word per tile (FCAMD_EVAL_PACKED_HISTORY, include/fcamd.h; ``device.pack_rows``).
        # A launch reads and writes contiguous runs instead of isolated 48-byte rows, and ``update()`` stays a pointer swap.
        # ``history`` / ``history_committed`` hand out the plastic-strain array in the reference's layout as a COPY (always,
        # from the first call on -- nothing changes meaning mid-run); initialise with ``set_state``.  (Round 3 had a
        # "delta" protocol here -- increments in the trial array plus a commit kernel per increment; the packed layout
        # reaches the same byte rate without the commit kernel and replaced it.)

        # ``split_history`` (the comfe-rs plasticity laws, with the sparse protocol): the reference keeps one
        # [scalar, eps_p(6)] row of 7 doubles per point (``history_dim = {"history": 7}``), so every point pays 56 bytes
        # of history reads for the scalar; the state keeps the scalars and the eps_p rows in two arrays instead
        # (FCAMD_EVAL_SPLIT_HISTORY: elastic points read 8 bytes and write nothing) and ``history`` /
        # ``history_committed`` assemble the reference's rows on demand (copies -- initialise with ``set_state``).

        self.law, self.n = law, int(n)
        from . import _capi
        from .device import SPLIT_HISTORY_FIELDS, SPLIT_HISTORY_LAWS

        self.device = torch.device("cuda", _capi.default_device()) if device is None else torch.device(device)
        gd2, sd = law.geometric_dim**2, law.stress_strain_dim
        self._gd2, self._sd = gd2, sd
        f = dict(dtype=torch.float64, device=self.device)
        self._f = f
        self._grad = self._tangent = None  # device-assembler mode only; allocated on first use
        self._stress = [torch.zeros(sd * n, **f), torch.zeros(sd * n, **f)]
        hd = law.history_dim
        self._split = bool(split_history) and bool(sparse_history) and type(law).__name__ in SPLIT_HISTORY_LAWS
        if self._split:
            hd = dict(SPLIT_HISTORY_FIELDS)
        self._hist = None if hd is None else [{k: torch.zeros(d * n, **f) for k, d in hd.items()} for _ in range(2)]
        self._c = 0  # index of the committed copy
        self._launch_cache = _capi.LaunchCache()  # committed copy -> the kept argument arrays of evaluate()'s call
        if stress0 is not None:
            self._stress[0].copy_(self._as_dev(stress0))
        if history0 is not None and self._hist is not None:
            h0 = self._internal_history(history0)
            for k in self._hist[0]:
                self._hist[0][k].copy_(h0[k])
                self._hist[1][k].copy_(self._hist[0][k])  # trial == committed (sparse-history contract)
        # Plasticity laws: sparse trial history (fcamd_evaluate_device_from_sparse).  Elastic points keep
        # their history, so only plastic / formerly plastic points are written; the mask (one word
        # per 64-point tile) survives the pointer swap of update().
        self._mask = None
        if sparse_history and type(law).__name__ in ("VonMises3D", "MisesPlasticityLinearHardening3D",
                                                     "DruckerPrager3D", "DruckerPragerHyperbolic3D"):
            self._mask = torch.zeros((self.n + 63) // 64, dtype=torch.int64, device=self.device)
        # The packed layout needs a plastic-strain array of its own: VonMises3D's eps_n, the split laws' eps_p rows.
        self._rows_key = "eps_n" if type(law).__name__ == "VonMises3D" else ("rows" if self._split else None)
        self._packed = bool(packed_history) and self._mask is not None and self._rows_key is not None
        # EVER masks of the two copies of the packed plastic-strain array (one word per tile), swapped with them
        self._ever = [torch.zeros_like(self._mask), torch.zeros_like(self._mask)] if self._packed else None
        if self._packed and history0 is not None:
            self._pack_into_both(self._hist[0][self._rows_key].clone())
        self._n_eval = 0          # evaluates of the increment in progress (Newton iterations)
        self._evaluated = False
        # Linear elasticity and the SLS laws have one tangent for all points, a function of the
        # parameters (and del_t) only -- the reference tiles it into the array on every call
        # (linear_elasticity_model.py:45, spring_maxwell_model.py:84-86, spring_kelvin_model.py:85-86).
        # The device tangent array is owned by this object, so once it holds the tangent of a given
        # del_t it is not rewritten: 288 of the 456 / 648 bytes per point disappear from every further
        # Newton iteration and increment.  (Writing into ``.tangent`` from outside voids this: pass
        # reuse_constant_tangent=False then.)
        self._const_tangent = reuse_constant_tangent and type(law).__name__ in (
            "LinearElasticityModel", "LinearElasticity3D", "SpringMaxwellModel", "SpringKelvinModel")
        self._tangent_key = None  # del_t for which the tangent array is valid
        # Plasticity laws, the same idea per point: the tangent of an elastic point is one constant, so once
        # an array holds the tangent of the previous evaluate only the rows of plastic / formerly plastic
        # points are rewritten (FCAMD_EVAL_SPARSE_TANGENT; rows of points that stay elastic keep 288 of
        # their 464 bytes off the bus).  ``_tangent_target`` names the array that received the previous
        # evaluate's tangent ("dev" or the address of the host assembler's array): only that one is current.
        self._sparse_tangent = bool(sparse_tangent) and self._mask is not None
        self._tangent_target = None
        self._host_tangent_key = None  # (address, bytes, del_t) of the host array that holds the constant tangent
        self._host_tangent_ref = None  # the last host tangent array (kept alive, see evaluate_into)
        # Non-convergence must not be committed (the reference raises inside evaluate,
        # mises_plasticity_isotropic_hardening.py:141-143; general.rs:186; the Drucker-Prager tip,
        # drucker_prager_classic.rs:82).  The device launches are asynchronous, so the state owns the counters
        # of its launches (fcamd_eval_args.counters: states sharing one law object cannot read each other's)
        # and update() reads them -- one 2 KB copy -- before it swaps the pointers.
        self._counts = type(law).__name__ in ("VonMises3D", "MisesPlasticityLinearHardening3D",
                                              "DruckerPrager3D", "DruckerPragerHyperbolic3D")
        from .device import new_counters

        self._counters = new_counters(self.device) if self._counts else None
        self._stats_pending = False  # a device launch whose counters have not been looked at yet
        self._failed = None          # the error of the last evaluate, if it raised: nothing to commit
        self._placement_mode = placement
        #: Incremented whenever the state REPLACES its device arrays (the placement step of the first large device-assembler
        #: ``evaluate`` -- or an explicit ``prepare()`` -- and ``tune_placement``): tensors obtained earlier from ``stress`` /
        #: ``tangent`` / ``grad`` / ``history`` then alias arrays the kernel no longer writes.  A device assembler that grabs
        #: the tensors once calls ``prepare()`` first (or compares ``generation``).
        self.generation = 0
        self._placed = placement == "torch"  # nothing to do
        self._vmm = None
        self.placement = None  # what the placement step did, once it has run

    def _internal_history(self, history) -> dict:
        """The caller's history dict (NumPy arrays or device tensors, the law's ``history_dim`` fields) in the state's
        own layout, as device tensors."""
        if not self._split:
            return {k: self._as_dev(v) for k, v in history.items()}
        from .device import split_history_rows

        return split_history_rows(self._as_dev(history["history"]))

    def _pack_into_both(self, rows) -> None:
        """both copies of the plastic-strain array <- ``rows`` (reference layout) in the packed layout, both EVER masks"""
        from .device import pack_rows

        packed, ever = pack_rows(rows)
        for i in (0, 1):
            self._hist[i][self._rows_key].copy_(packed)
            self._ever[i].copy_(ever)

    def _unpacked(self, copy: int) -> dict:
        """internal history dict of one copy with the plastic-strain array in the reference's layout (a new tensor)"""
        from .device import unpack_rows

        h = self._hist[copy]
        return {**h, self._rows_key: unpack_rows(h[self._rows_key], self._ever[copy], self.n)}

    def _external_history(self, internal) -> dict:
        if not self._split:
            return internal
        from .device import join_history_rows

        return {"history": join_history_rows(internal)}

    def _as_dev(self, a):
        import torch

        return a if _is_torch(a) else to_device(a, self.device, np.float64)

    @property
    def grad(self):
        if self._grad is None:
            import torch

            self._grad = torch.zeros(self._gd2 * self.n, **self._f)
        return self._grad

    @property
    def tangent(self):
        """The trial tangent array (allocated on first use).  With ``placement`` other than "torch" the first large
        ``evaluate`` may move the state's arrays: take the tensor after ``prepare()`` / the first evaluate, or watch
        ``generation``."""
        if self._tangent is None:
            import torch

            self._tangent = torch.zeros(self._sd * self._sd * self.n, **self._f)
        return self._tangent

    # committed (previous) and trial (current) views -------------------------------------------
    @property
    def stress_committed(self):
        """Committed stress (live tensor).  To (re)initialise the state use ``set_state``."""
        return self._stress[self._c]

    @property
    def stress(self):
        return self._stress[1 - self._c]

    @property
    def history_committed(self):
        """Committed history (live tensors; the plastic-strain array of a packed state -- ``packed_history`` -- is a copy in
        the reference's layout).  Invariant of the sparse protocol: the trial history equals
        the committed one wherever the mask is clear -- do not write initial / restart values into these
        tensors, ``set_state`` writes both copies."""
        if self._hist is None:
            return None
        return self._external_history(self._unpacked(self._c) if self._packed else self._hist[self._c])

    @property
    def history(self):
        """Trial history.  The plastic-strain array of a packed state is a copy in the reference's layout; every other tensor
        is the live one the kernel writes."""
        if self._hist is None:
            return None
        if self._packed:  # nothing evaluated in this increment yet: the trial state is the committed one
            return self._external_history(self._unpacked(1 - self._c if self._evaluated else self._c))
        return self._external_history(self._hist[1 - self._c])

    def set_state(self, stress=None, history=None) -> None:
        """(Re)initialise the committed state -- initial conditions, a restart -- from NumPy arrays or device
        tensors.  Both copies of the history are written: the sparse trial-history protocol needs trial ==
        committed wherever the mask is clear, so writing into ``history_committed`` alone would let ``update()``
        commit stale rows at points that stay elastic.  Every shortcut that refers to earlier evaluates (mask,
        tangent bookkeeping) is reset."""
        if stress is not None:
            self.stress_committed.copy_(self._as_dev(stress))
        if history is not None and self._hist is not None:
            h = self._internal_history(history)
            for k in self._hist[self._c]:
                if self._packed and k == self._rows_key:
                    self._pack_into_both(h[k])
                    continue
                self._hist[self._c][k].copy_(h[k])
                self._hist[1 - self._c][k].copy_(self._hist[self._c][k])
        if self._mask is not None:
            self._mask.zero_()
        self._tangent_target = self._tangent_key = self._host_tangent_key = None
        self._evaluated = False
        self._n_eval = 0
        self._stats_pending = False
        self._failed = None

    # the Newton-iteration call --------------------------------------------------------------------
    def _launch(self, t, del_t, g, tangent, sparse_tangent=False) -> None:
        self.law.evaluate_from(t, del_t, g, self.stress_committed, self.stress, tangent,
                               None if self._hist is None else self._hist[self._c],
                               None if self._hist is None else self._hist[1 - self._c],
                               history_mask=self._mask, sparse_tangent=sparse_tangent, counters=self._counters,
                               split_history=self._split,
                               packed_masks=(self._ever[self._c], self._ever[1 - self._c]) if self._packed else None)

    def evaluate(self, t: float, del_t: float, grad_del_u) -> None:
        """Trial state <- law(committed state, grad_del_u).  May be called any number of times per
        increment; the committed state is never modified.  Asynchronous; Newton non-convergence surfaces at
        ``check()`` / ``update()``."""
        g = grad_del_u
        staging = not _is_torch(g)
        if staging:
            upload(self.grad, np.ascontiguousarray(g, dtype=np.float64))  # synchronous: the caller may free or rewrite g on return
            g = self.grad
        assert g.numel() == self._gd2 * self.n, "grad_del_u has the wrong length"
        if not self._placed:
            self._placed = True
            if 8 * self._sd * self._sd * self.n >= self.AUTO_TUNE_MIN_BYTES:
                self._place(t, del_t, g, staging)
                if staging:
                    g = self.grad  # the staging buffer may have moved (its contents moved along)
        tangent = self.tangent
        key = None
        if self._const_tangent:
            key = float(del_t) if type(self.law).__name__.startswith("Spring") else 0.0
            if self._tangent_key == key:
                tangent = None  # already holds exactly what this launch would write
            self._tangent_key = None  # valid again only once the launch below has been enqueued
        sparse = self._sparse_tangent and self._tangent_target == "dev"
        self._tangent_target = None
        self._failed = None
        # the Newton iterations of an increment issue the very same call (same arrays, new gradient VALUES): its argument arrays are
        # kept and replayed; a small state leaves through the batch kernel (counters + ONE launch instead of counters, main kernel,
        # ragged tile: 52 -> 38 us per iteration at 1e4 points)
        sig = (float(del_t), g.data_ptr(), 0 if tangent is None else tangent.data_ptr(), sparse, self.stress_committed.data_ptr(),
               self.stress.data_ptr(), self.generation)
        self._launch_cache.run(self._c, sig, lambda: self._launch(t, del_t, g, tangent, sparse), _current_stream_ptr(self.device.index or 0))
        self._tangent_key = key
        self._tangent_target = "dev"
        self._evaluated = True
        self._n_eval += 1
        self._stats_pending = self._counts

    def prepare(self, t: float, del_t: float, grad_del_u) -> None:
        """Run the placement step NOW (it otherwise runs inside the first large device-assembler ``evaluate``) and leave a
        valid trial state for ``grad_del_u``: after it the tensors handed out by ``stress`` / ``tangent`` / ``grad`` and the
        live tensors of ``history`` stay the ones the kernel writes (``generation`` does not change any more unless
        ``tune_placement`` is called).  The plastic-strain array of ``history`` is NOT such a tensor under ``packed_history``
        (always a copy): read it through ``history`` when needed.  For device assemblers that take the pointers once and then loop."""
        self.evaluate(t, del_t, grad_del_u)

    def _place(self, t, del_t, g, staging: bool) -> None:
        """First large device-assembler evaluate: "tune" -- the fastest of a few hipMalloc candidates of the
        tangent; "vmm" -- the state's arrays in one interleaved VMM working set; "auto" -- both, timed with the
        state's own launch, the faster kept (the hipMalloc draws spread over 10-28 % but their best can beat the
        VMM set by a few per cent; the VMM set is within ~5 % of that best on every box seen so far)."""
        mode, best_ms = self._placement_mode, None
        if mode in ("auto", "tune") and not self._const_tangent:  # a tangent written once per del_t: nothing to tune
            info = self.tune_placement(t, del_t, g, tries=6)  # a third of the draws are slow ones: six, memory permitting
            best_ms = min(info["candidate_ms"])
            self.placement = {"mode": "hipmalloc_tuned", **info}
        if mode in ("auto", "vmm"):
            saved = (self._stress, self._hist, self._tangent, self._grad)
            try:
                self._move_to_vmm(staging)
                if best_ms is not None:
                    vmm_ms = self._time_launch(t, del_t, self._grad if staging else g)
                    self.placement.update({"vmm_ms": round(vmm_ms, 4), "hipmalloc_best_ms": round(best_ms, 4)})
                    if vmm_ms >= best_ms:  # the tuned hipMalloc arrays win: back to them
                        self._stress, self._hist, self._tangent, self._grad = saved
                        self._vmm = None
                        self.placement["mode"] = "hipmalloc_tuned"
            except Exception as e:  # no VMM support / no room for the move: keep what there is
                if mode == "vmm":
                    raise
                self._stress, self._hist, self._tangent, self._grad = saved
                self._vmm = None
                self.placement = {**(self.placement or {"mode": "torch"}), "vmm_error": f"{type(e).__name__}: {e}"[:200]}
            self._tangent_key = self._tangent_target = None  # whichever tangent array it is: written in full next
        self.generation += 1

    def _time_launch(self, t, del_t, g, launches: int = 3) -> float:
        import torch

        self._launch(t, del_t, g, self.tangent)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
        for a, b in ev:
            a.record()
            self._launch(t, del_t, g, self.tangent)
            b.record()
        torch.cuda.synchronize(self.device)
        return min(a.elapsed_time(b) for a, b in ev)

    def _move_to_vmm(self, staging: bool) -> None:
        from . import _capi
        from .placement import VmmArraySet

        n, sd = self.n, self._sd
        numels = {"tangent": sd * sd * n, "stress0": sd * n, "stress1": sd * n}
        if self._hist is not None:
            for k, v in self._hist[0].items():  # the state's own layout (split or the law's fields)
                numels[f"h0_{k}"] = v.numel()
                numels[f"h1_{k}"] = v.numel()
        if staging or self._grad is not None:
            numels["grad"] = self._gd2 * n
        ctx = _capi.get_context(self.device.index or 0)
        vmm = VmmArraySet(ctx, numels, interleaved=True, device=self.device)
        new_stress = [vmm["stress0"], vmm["stress1"]]
        for i in (0, 1):
            new_stress[i].copy_(self._stress[i])
        if self._hist is not None:
            new_hist = [{k: vmm[f"h{i}_{k}"] for k in self._hist[i]} for i in (0, 1)]
            for i in (0, 1):
                for k in new_hist[i]:
                    new_hist[i][k].copy_(self._hist[i][k])
            self._hist = new_hist
        self._stress = new_stress
        self._tangent = vmm["tangent"]
        if "grad" in numels:
            new_grad = vmm["grad"]
            if self._grad is not None:
                new_grad.copy_(self._grad)
            self._grad = new_grad
        self._vmm = vmm
        self.placement = {**(self.placement or {}), "mode": "vmm_interleaved", "arrays": list(numels),
                          "GB": round(8 * sum(numels.values()) / 1e9, 2)}

    def tune_placement(self, t: float, del_t: float, grad_del_u, tries: int = 4) -> dict:
        """Device-assembler mode: choose the placement of the tangent array (the dominant write
        stream) by timing this state's own evaluate on a few candidate allocations and keeping the
        fastest (``placement.fastest_allocation``; on MI355X the kernel time follows where the
        written arrays live, by up to 20 %).  The ``placement="tune"`` mode of the state (and the fallback
        of "auto") runs it on the first ``evaluate``; leaves a valid trial state for ``grad_del_u``.
        Returns the candidate timings."""
        from .placement import fastest_allocation

        g = grad_del_u
        if not _is_torch(g):
            upload(self.grad, np.ascontiguousarray(g, dtype=np.float64))
            g = self.grad
        assert g.numel() == self._gd2 * self.n, "grad_del_u has the wrong length"
        self._placed = True
        first, self._tangent = self._tangent, None
        self._tangent_target = None  # the candidates hold no previous tangent: every probe writes every row
        self._tangent, info = fastest_allocation(
            self._sd * self._sd * self.n, lambda tan: self._launch(t, del_t, g, tan),
            tries=tries, device=self.device, first=first)
        del first
        self.generation += 1
        self._tangent_key = None  # a constant tangent has to be written into the chosen array
        if self.placement is None:
            self.placement = {"mode": "hipmalloc_tuned", **info}
        self.evaluate(t, del_t, g)
        return info

    def evaluate_into(self, t: float, del_t: float, grad_del_u: np.ndarray, stress: np.ndarray | None = None,
                      tangent: np.ndarray | None = None):
        """The host assembler's Newton-iteration call: trial state <- law(committed state,
        grad_del_u) with ``grad_del_u`` a NumPy array, and the trial stress / tangent written into
        the caller's NumPy arrays, all in one chunk-pipelined pass.  Synchronous; raises the
        reference's exceptions (non-convergence) like the ndarray ``evaluate``.  The caller's tangent
        array is taken to be left alone between calls (the assembler only reads it): for the laws with a
        point-independent tangent (linear elasticity, SLS at a fixed ``del_t``) it is downloaded once and
        not again while ``del_t`` and the array stay the same (288 of the 336 bytes per point); for the
        plasticity laws with a page-locked array only the rows of plastic / formerly plastic points are
        written from the second call on (``sparse_tangent``).  Construct the state with
        ``reuse_constant_tangent=False`` / ``sparse_tangent=False`` if the array is modified in between.
        Page-lock the three arrays once (``law.pin_host_arrays`` / ``Context.register_host_buffer``): the
        kernel then reads the gradient from and writes the tangent to them directly."""
        from . import _capi
        from .device import _check_numpy

        _check_numpy("grad_del_u", grad_del_u)
        assert grad_del_u.size == self._gd2 * self.n, "grad_del_u has the wrong length"
        if stress is not None:
            _check_numpy("stress", stress)
            assert stress.size == self._sd * self.n, "stress has the wrong length"
        if tangent is not None:
            _check_numpy("tangent", tangent)
            assert tangent.size == self._sd * self._sd * self.n, "tangent has the wrong length"
        dev = self.device.index or 0
        m = self.law._handle(dev)
        import torch

        m.ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        from .device import SPLIT_HISTORY_FIELDS

        fields = SPLIT_HISTORY_FIELDS if self._split else m.history_fields
        hp = [] if self._hist is None else [self._hist[self._c][k].data_ptr() for k, _ in fields]
        hc = [] if self._hist is None else [self._hist[1 - self._c][k].data_ptr() for k, _ in fields]
        self._evaluated = True  # the trial state is touched even if the call raises
        self._n_eval += 1
        # sparse tangent: only into the very array that received the previous evaluate's tangent, and only
        # when the kernel writes it directly (page-locked array; the C side ignores the flag otherwise --
        # then every row is downloaded and the array is current as well)
        if tangent is not None and self._const_tangent:
            # LE / SLS: the caller's array already holds the tangent of this del_t (written by an earlier call
            # into the very same array): nothing to download, 288 of the 336 bytes per point stay off PCIe
            key = (tangent.ctypes.data, tangent.nbytes, float(del_t) if type(self.law).__name__.startswith("Spring") else 0.0)
            if self._host_tangent_key == key:
                tangent = None
            else:
                self._host_tangent_key = None  # set below, once the call has succeeded
        else:
            key = None
        target = None if tangent is None else ("host", tangent.ctypes.data, tangent.nbytes)
        flags = _capi.EVAL_SPARSE_TANGENT if (self._sparse_tangent and target is not None
                                              and self._tangent_target == target) else 0
        if self._split:
            flags |= _capi.EVAL_SPLIT_HISTORY
        if self._packed:
            flags |= _capi.EVAL_PACKED_HISTORY
        self._tangent_target = None
        self._stats_pending = False  # synchronous: the call itself reports
        try:
            self.law.last_stats = m.evaluate_resident(
                t, del_t, self.n, grad_del_u.ctypes.data, self.stress_committed.data_ptr(), self.stress.data_ptr(),
                hp, hc, None if self._mask is None else self._mask.data_ptr(),
                None if stress is None else stress.ctypes.data, None if tangent is None else tangent.ctypes.data, flags,
                packed_mask_ptrs=(self._ever[self._c].data_ptr(), self._ever[1 - self._c].data_ptr()) if self._packed else None)
        except Exception as e:
            self._failed = e  # the trial state is not fit to be committed: update() raises until a clean evaluate
            raise
        self._failed = None
        self._tangent_target = target
        if key is not None and tangent is not None:
            self._host_tangent_key = key
        if tangent is not None:
            # The two shortcuts above identify the caller's array by address.  Holding a reference keeps its
            # memory from being freed and handed out again for a different array at the same address.
            self._host_tangent_ref = tangent
        return self.law.last_stats

    def update(self) -> None:
        """Commit the trial state (``IncrSmallStrainProblem.update``, solver/_solver.py:149-159):
        a pointer swap -- after the counters of the last evaluate have been looked at: a trial state with
        a non-converged (or out-of-domain) point raises the reference's error instead of being committed."""
        if not self._evaluated:
            raise RuntimeError("update() before any evaluate() of this increment")
        if self._failed is not None:
            raise RuntimeError(f"the last evaluate failed, nothing to commit: {self._failed}")
        self.check()
        self._c = 1 - self._c
        self._evaluated = False
        self._n_eval = 0

    # host access ------------------------------------------------------------------------------------
    def download(self, stress: np.ndarray | None = None, tangent: np.ndarray | None = None,
                 history: dict | None = None) -> None:
        """Copy the trial stress / tangent / history into the caller's NumPy arrays in place."""
        if stress is not None:
            assign(stress, self.stress)
        if tangent is not None:
            assign(tangent, self.tangent)
        if history is not None and self._hist is not None:
            trial = self.history  # once: the plastic-strain array of a packed state is assembled on demand
            for k in history:
                assign(history[k], trial[k])

    def check(self):
        """Synchronise with this state's last device evaluate and return its counters; raises the
        reference's RuntimeError on Newton non-convergence / the Drucker-Prager tip.  Cheap when nothing
        is pending (the counters are read once per evaluate)."""
        if self._counters is None:  # a law without counters: check() is still the synchronisation point it always was
            import torch

            torch.cuda.current_stream(self.device).synchronize()
            return None
        if self._stats_pending:
            from .device import read_counters

            self._stats = read_counters(self._counters)
            self.law.last_stats = self._stats
            self._stats_pending = False
            try:
                self.law.raise_for_stats(self._stats)
            except RuntimeError as e:
                self._failed = e
                raise
        elif self._failed is not None:
            raise self._failed
        return getattr(self, "_stats", None)
