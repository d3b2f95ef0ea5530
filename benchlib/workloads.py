"""bench.py: the synthetic workloads, their arrays, protocols and placement."""

from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)
METRIC = "quadrature-point stress updates/sec (Mpts/s) + % HBM roofline, 1/2/4/8 GPU"  # BASELINE.json "metric"

VM_P = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}
RS_P = {"mu": 80769.0, "kappa": 175000.0, "y_0": 1200.0, "h": 200.0}
SLS_P = {"E0": 42.0, "E1": 10.0, "tau": 10.0, "nu": 0.2}
LE_P = {"E": 42.0, "nu": 0.3}
DP_P = {"mu": 80769.0, "kappa": 175000.0, "a": 100.0, "b": 0.05, "b_flow": 0.02}

# workload -> (law kind, strain scale spec, bytes/pt elastic, bytes/pt plastic)   [SURVEY.md 8d]
WORKLOADS = {
    "von_mises_mixed": ("von_mises_3d", "loguniform", 464, 568),
    # the same 22 % of plastic points, but in contiguous zones of 4096 points (what a mesh-ordered
    # plastic zone looks like) instead of a random mixture in every 64-point tile
    "von_mises_zoned": ("von_mises_3d", "zoned", 464, 568),
    "von_mises_plastic": ("von_mises_3d", 1e-2, 464, 568),
    "von_mises_elastic": ("von_mises_3d", 1e-4, 464, 568),
    "linear_elasticity": ("linear_elasticity", 1e-3, 456, 456),
    "spring_maxwell": ("spring_maxwell", 1e-3, 648, 648),
    "spring_kelvin": ("spring_kelvin", 1e-3, 648, 648),
    "comfe_mises_mixed": ("comfe_mises_plasticity", "loguniform", 464, 568),
    # general return mapping (Newton per plastic point, invariant coordinates)
    "drucker_prager_mixed": ("comfe_drucker_prager", "isochoric", 464, 568),
    "drucker_prager_zoned": ("comfe_drucker_prager", "isochoric_zoned", 464, 568),
    "comfe_mises_zoned": ("comfe_mises_plasticity", "zoned", 464, 568),
}
HEADLINE = "von_mises_mixed"
# the other single-GPU configurations of BASELINE.json, timed after the headline in the default run
EXTRA_CONFIGS = ["linear_elasticity", "von_mises_plastic", "von_mises_elastic", "spring_maxwell", "spring_kelvin"]
BASELINE_CONFIG = {"linear_elasticity": "configs[1]", "von_mises_mixed": "configs[2] (mixed)", "von_mises_plastic": "configs[2] (all-plastic)",
                   "von_mises_elastic": "configs[2] (all-elastic)", "spring_maxwell": "configs[3]", "spring_kelvin": "configs[3] (Kelvin twin)"}
PLASTICITY = ("von_mises_3d", "comfe_mises_plasticity", "comfe_drucker_prager")


def make_law(kind):
    import numpy as np

    import fenics_constitutive_amd as fc

    FULL = fc.StressStrainConstraint.FULL
    if kind == "von_mises_3d":
        return fc.VonMises3D(VM_P), VM_P
    if kind == "linear_elasticity":
        return fc.LinearElasticityModel(LE_P, FULL), LE_P
    if kind == "spring_maxwell":
        return fc.SpringMaxwellModel(SLS_P, FULL), SLS_P
    if kind == "spring_kelvin":
        return fc.SpringKelvinModel(SLS_P, FULL), SLS_P
    if kind == "comfe_mises_plasticity":
        return fc.MisesPlasticityLinearHardening3D({k: np.array([v]) for k, v in RS_P.items()}), RS_P
    if kind == "comfe_drucker_prager":
        return fc.DruckerPrager3D({k: np.array([v]) for k, v in DP_P.items()}), DP_P
    raise ValueError(kind)


def synth_inputs(kind, scale_spec, n, seed, device):
    """Synthetic state, generated on the device: returns
    (gradient generator, committed stress, committed history dict)."""
    import torch

    gen = torch.Generator(device=device)
    gen.manual_seed(seed)

    def grad_array():
        g = torch.randn(9 * n, dtype=torch.float64, device=device, generator=gen)
        if scale_spec == "loguniform":
            sc = torch.pow(10.0, torch.rand(n, dtype=torch.float64, device=device, generator=gen) * 2.0 - 4.0)
            g.view(n, 9).mul_(sc[:, None])
        elif scale_spec == "zoned":
            zone = 4096
            nz = (n + zone - 1) // zone
            pl = torch.rand(nz, dtype=torch.float64, device=device, generator=gen) < 0.22
            sc = torch.where(pl, 1e-2, 1e-4).to(torch.float64).repeat_interleave(zone)[:n]
            g.view(n, 9).mul_(sc[:, None])
        elif scale_spec in ("isochoric", "isochoric_zoned"):
            # Drucker-Prager: mostly isochoric increments, scale log-uniform in [1e-4, 5e-3] (keeps the
            # trial states away from the tip of the classic surface); zoned: 4096-point zones, 22 % of
            # them at 5e-3, the others at 1e-4
            if scale_spec == "isochoric":
                sc = torch.pow(10.0, torch.rand(n, dtype=torch.float64, device=device, generator=gen) * 1.7 - 4.0)
            else:
                zone = 4096
                pl = torch.rand((n + zone - 1) // zone, dtype=torch.float64, device=device, generator=gen) < 0.22
                sc = torch.where(pl, 5e-3, 1e-4).to(torch.float64).repeat_interleave(zone)[:n]
            gv = g.view(n, 9)
            gv.mul_(sc[:, None])
            tr = (gv[:, 0] + gv[:, 4] + gv[:, 8]) * (0.95 / 3.0)
            for c in (0, 4, 8):
                gv[:, c] -= tr
        else:
            g.mul_(float(scale_spec))
        return g

    stress = torch.zeros(6 * n, dtype=torch.float64, device=device)
    if kind == "von_mises_3d":
        hist = {"eps_n": torch.zeros(6 * n, dtype=torch.float64, device=device),
                "alpha": torch.rand(n, dtype=torch.float64, device=device, generator=gen) * 0.02}
    elif kind in ("spring_maxwell", "spring_kelvin"):
        hist = {"strain_visco": torch.zeros(6 * n, dtype=torch.float64, device=device),
                "strain": torch.zeros(6 * n, dtype=torch.float64, device=device)}
    elif kind == "comfe_drucker_prager":
        hist = {"history": torch.zeros(7 * n, dtype=torch.float64, device=device)}
        stress.view(n, 6)[:, :3] = -1000.0  # compressive prestress
    elif kind == "comfe_mises_plasticity":
        h = torch.zeros(7 * n, dtype=torch.float64, device=device)
        h.view(n, 7)[:, 0] = torch.rand(n, dtype=torch.float64, device=device, generator=gen) * 0.02
        hist = {"history": h}
    else:
        hist = None
        stress.normal_(generator=gen)  # cfg2: sigma_in ~ N(0,1) exercises the "+="
    return grad_array, stress, hist


class Workload:
    """One law on n synthetic device-resident points: the committed state, two Newton iterates of the
    gradient, the trial arrays, and the launch every timed step issues."""

    def __init__(self, name, n, seed, device, dev_index, history="packed", sparse_tangent=False, grid=0, split_history=True):
        import torch

        self.torch = torch
        self.name, self.n, self.device, self.dev_index = name, n, device, dev_index
        self.kind, scale_spec, self.b_el, self.b_pl = WORKLOADS[name]
        self.del_t = 2.0
        self.law, self.params = make_law(self.kind)
        self.launch_log = []  # [phase, evaluate launches]: lets tools/summarize_profile.py slice a kernel trace
        grad_array, self.stress_c, self.hist_c = synth_inputs(self.kind, scale_spec, n, seed, device)
        # one in-place warm step from the initial state gives a committed state "from a previous step"
        self.tangent = torch.empty(36 * n, dtype=torch.float64, device=device)
        g_warm = grad_array()
        self.law.evaluate(0.0, self.del_t, g_warm, self.stress_c, self.tangent, self.hist_c)
        self.launch_log.append(["warm_in_place", 1])
        del g_warm
        # Two Newton iterates of one increment, evaluated alternately: between the iterations of the
        # reference's Newton loop only grad_del_u changes (solver/_solver.py:130-147), and with it the
        # plastic set at its margin -- so the sparse protocol sees new and stale points as it does in use.
        self.grads = [grad_array()]
        self.grads.append(self.grads[0] * 1.03)
        # trial-state arrays: every timed step reads the committed state and writes the trial state
        # (same traffic as in place, stationary workload)
        self.stress_t = torch.empty_like(self.stress_c)
        self.hist_t = None if self.hist_c is None else {k: torch.empty_like(v) for k, v in self.hist_c.items()}
        if grid:
            self.law._handle(dev_index).ctx.set_grid(grid)
        self.plasticity = self.kind in PLASTICITY
        self.sparse = self.plasticity and history in ("sparse", "packed")
        self.sparse_tangent = bool(sparse_tangent and self.sparse)
        # comfe-rs plasticity laws under the sparse protocol: ResidentState keeps their [scalar, eps_p(6)] history rows as
        # two arrays (FCAMD_EVAL_SPLIT_HISTORY) -- an internal layout of the device-resident state, same results
        self.split = bool(split_history and self.sparse and self.kind in ("comfe_mises_plasticity", "comfe_drucker_prager"))
        self.rows_key = "eps_n" if self.kind == "von_mises_3d" else ("rows" if self.split else None)  # the array that only accumulates plastic strain
        if self.split:
            from fenics_constitutive_amd.device import split_history_rows

            self.hist_c = split_history_rows(self.hist_c["history"])
            self.hist_t = {k: torch.empty_like(v) for k, v in self.hist_c.items()}
        # Packed plastic-strain history (FCAMD_EVAL_PACKED_HISTORY; ResidentState's default layout of the array that only
        # accumulates): committed and trial copy hold the rows of the ever-plastic points of every tile as one contiguous run,
        # one EVER-mask word per tile next to each; the commit stays a pointer swap.  Same values, same launch, same bytes
        # asked of the interface -- only the rows move as full lines instead of isolated 48-byte pieces.
        self.packed = bool(self.sparse and history == "packed" and self.rows_key is not None)
        self.ever_c = self.ever_t = None
        self._plain = None  # unpacked twin of the history arrays for the legs that run another protocol (full / unpacked sparse)
        if self.packed:
            from fenics_constitutive_amd.device import pack_rows

            self.hist_c[self.rows_key], self.ever_c = pack_rows(self.hist_c[self.rows_key])
            self.ever_t = self.ever_c.clone()
        self.hmask = None
        if self.sparse:
            for k in self.hist_c:
                self.hist_t[k].copy_(self.hist_c[k])  # contract: trial == committed where the mask is clear
            self.hmask = torch.zeros((n + 63) // 64, dtype=torch.int64, device=device)
        self.placement = None
        self._vmm, self.vmm_info = None, None
        self.n_pl_ab, self.its_ab = [0, 0], [0, 0]

    def launch(self, i, tangent=None, full_history=False, sparse_tangent=None, m=None, unpacked=False):
        """`m`: evaluate the first m points of the arrays only (the strong-scaling leg of a weak-scaling run)"""
        tan = self.tangent if tangent is None else tangent
        packed = self.packed and not full_history and not unpacked
        hist_c, hist_t, hmask = self.hist_c, self.hist_t, self.hmask
        if self.packed and not packed:  # another protocol on this workload: it needs the plain layout of the same state
            hist_c, hist_t, hmask = self.plain_twin()
        pm = None
        if m is None:
            g, sc, st, hc, ht, mask = self.grads[i & 1], self.stress_c, self.stress_t, hist_c, hist_t, hmask
            if packed:
                pm = (self.ever_c, self.ever_t)
        else:
            dims = {"eps_n": 6, "alpha": 1, "strain_visco": 6, "strain": 6, "history": 7, "scalar": 1, "rows": 6}
            g, sc, st, tan = self.grads[i & 1][: 9 * m], self.stress_c[: 6 * m], self.stress_t[: 6 * m], tan[: 36 * m]
            hc = None if hist_c is None else {k: v[: dims[k] * m] for k, v in hist_c.items()}
            ht = None if hist_t is None else {k: v[: dims[k] * m] for k, v in hist_t.items()}
            mask = None if hmask is None else hmask[: (m + 63) // 64]
            if packed:
                pm = (self.ever_c[: (m + 63) // 64], self.ever_t[: (m + 63) // 64])
        self.law.evaluate_from(0.0, self.del_t, g, sc, st, tan, hc, ht,
                               history_mask=None if full_history else mask,
                               sparse_tangent=self.sparse_tangent if sparse_tangent is None else sparse_tangent,
                               split_history=self.split, packed_masks=pm)

    def plain_twin(self):
        """(committed history, trial history, mask) of this packed workload in the PLAIN layout -- built on first use, for the
        legs that time another protocol on the same state (full trial history, the sparse protocol on the reference's layout)"""
        if self._plain is None:
            from fenics_constitutive_amd.device import unpack_rows

            hc = dict(self.hist_c)
            hc[self.rows_key] = unpack_rows(self.hist_c[self.rows_key], self.ever_c, self.n)
            ht = {k: v.clone() for k, v in hc.items()}
            self._plain = (hc, ht, self.torch.zeros_like(self.hmask))
        return self._plain

    def drop_plain_twin(self):
        self._plain = None
        self.torch.cuda.empty_cache()

    def reference_history(self):
        """the committed history in the reference's layout (packed rows unpacked, the split layout joined back into 7-double rows)"""
        hist_c = self.plain_twin()[0] if self.packed else self.hist_c
        if not self.split:
            return hist_c
        from fenics_constitutive_amd.device import join_history_rows

        return {"history": join_history_rows(hist_c)}

    def tune_placement(self, tries):
        """hipMalloc placements of the tangent (the dominant write stream): a few candidate allocations, the
        real kernel timed on each, the fastest kept (ResidentState(placement="tune")).  Candidate 0 is the array
        that exists already, i.e. what a caller runs on who takes what the allocator gives."""
        if tries <= 1:
            return
        from fenics_constitutive_amd.placement import fastest_allocation

        self.tangent, self.placement = fastest_allocation(
            36 * self.n, lambda tan: self.launch(0, tangent=tan, sparse_tangent=False), tries=tries, device=self.device,
            first=self.tangent)
        self.launch_log.append(["placement_candidates", 4 * len(self.placement["candidate_ms"])])

    def _arrays(self):
        return {"tangent": self.tangent, "stress_c": self.stress_c, "stress_t": self.stress_t, "grads": self.grads,
                "hist_c": self.hist_c, "hist_t": self.hist_t}

    def _time_iterate0(self, launches=3):
        """min of `launches` event-timed launches of iterate 0 (after one warm launch), as fastest_allocation times a candidate"""
        torch = self.torch
        self.launch(0, sparse_tangent=False)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
        for a, b in ev:
            a.record()
            self.launch(0, sparse_tangent=False)
            b.record()
        torch.cuda.synchronize()
        return min(a.elapsed_time(b) for a, b in ev)

    def place_vmm(self, keep_if_faster_than=None):
        """Every array of the step in ONE working set whose 2 MiB physical handles are interleaved over the
        arrays (placement.VmmArraySet) -- what ResidentState(placement="auto" / "vmm") does with its arrays.
        `keep_if_faster_than` (ms): "auto" mode -- time iterate 0 on the set and go back to the hipMalloc arrays
        if they were faster."""
        from fenics_constitutive_amd.placement import VmmArraySet

        n = self.n
        numels = {"tangent": 36 * n, "stress_c": 6 * n, "stress_t": 6 * n, "grad0": 9 * n}
        two = self.grads[1] is not self.grads[0]
        if two:
            numels["grad1"] = 9 * n
        for k, v in (self.hist_c or {}).items():
            numels["hc_" + k] = v.numel()
            numels["ht_" + k] = v.numel()
        t0 = time.perf_counter()
        old = self._arrays()
        try:
            vmm = VmmArraySet(self.law._handle(self.dev_index).ctx, numels, interleaved=True, device=self.device)

            def moved(name, src):
                dst = vmm[name]
                dst.copy_(src)
                return dst

            self.tangent = vmm["tangent"]  # rewritten by every launch: nothing to copy
            self.stress_c, self.stress_t = moved("stress_c", old["stress_c"]), moved("stress_t", old["stress_t"])
            g0 = moved("grad0", old["grads"][0])
            self.grads = [g0, moved("grad1", old["grads"][1]) if two else g0]
            if old["hist_c"] is not None:
                self.hist_c = {k: moved("hc_" + k, v) for k, v in old["hist_c"].items()}
                self.hist_t = {k: moved("ht_" + k, v) for k, v in old["hist_t"].items()}
            self.torch.cuda.synchronize()
        except Exception as e:
            # no room for the second copy of the working set (e.g. under rocprofv3, which keeps released VMM memory
            # alive: round-2 probe vmm_leak_probe.py (git history)): "auto" stays on the tuned hipMalloc arrays, "vmm" has nothing to run on
            for k, v in old.items():
                setattr(self, k, v)
            if keep_if_faster_than is None:
                raise
            self.vmm_info = {"mode": "hipmalloc_tuned", "vmm_error": f"{type(e).__name__}: {e}"[:160]}
            self.torch.cuda.empty_cache()
            return
        info = {"mode": "vmm_interleaved", "arrays": len(numels), "GB": round(8 * sum(numels.values()) / 1e9, 2),
                "granule_MiB": 2, "build_s": round(time.perf_counter() - t0, 2)}
        if keep_if_faster_than is not None:
            info["vmm_ms"] = round(self._time_iterate0(), 4)
            self.launch_log.append(["vmm_candidate", 4])
            info["hipmalloc_best_ms"] = round(keep_if_faster_than, 4)
            if info["vmm_ms"] >= keep_if_faster_than:  # the tuned hipMalloc arrays win: back to them
                for k, v in old.items():
                    setattr(self, k, v)
                del vmm
                info["mode"] = "hipmalloc_tuned"
                self.vmm_info = info
                self.torch.cuda.empty_cache()
                return
        del old
        self.torch.cuda.empty_cache()
        self._vmm = vmm
        self.vmm_info = info

    def place(self, mode, tries):
        """first: what the allocator gives; tune: the fastest of `tries` hipMalloc candidates of the tangent; vmm:
        the interleaved VMM working set; auto (= ResidentState's default): the faster of the two."""
        if mode != "first":
            self.tune_placement(tries)  # in "vmm" mode for the record only: what the hipMalloc draws give
        if mode == "vmm":
            self.place_vmm()
        elif mode == "auto":
            best = min(self.placement["candidate_ms"]) if self.placement else self._time_iterate0()
            self.place_vmm(keep_if_faster_than=best)

    def count_plastic(self):
        """Plastic counts / Newton iterations of the two iterates (two more untimed launches)."""
        for i in (0, 1):
            self.launch(i)
            self.torch.cuda.synchronize()
            if self.plasticity:
                st = self.law.device_stats(self.dev_index)
                self.n_pl_ab[i], self.its_ab[i] = int(st.n_plastic), int(st.n_newton_iters)
        self.launch_log.append(["plastic_counts", 2])

    def alg_bytes(self, n_pl):
        """Algorithmic bytes of one launch: interface-mandated traffic."""
        return int(round((self.n - n_pl) * self.b_el + n_pl * self.b_pl))

    def mean_plastic(self, steps):
        n_b = steps // 2
        n_a = steps - n_b
        return (n_a * self.n_pl_ab[0] + n_b * self.n_pl_ab[1]) / steps, (n_a * self.its_ab[0] + n_b * self.its_ab[1]) / steps

    def warmup(self, w):
        for i in range(w):
            self.launch(i)
        self.launch_log.append(["warmup", w])

    def timed_events(self, steps, phase="timed", **kw):
        """`steps` launches bracketed one by one with events on the launch stream (the library launches on
        torch's current stream); returns the per-launch kernel times in ms after a synchronise."""
        torch = self.torch
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for i, (a, b) in enumerate(ev):
            a.record()
            self.launch(i, **kw)
            b.record()
        torch.cuda.synchronize()
        self.launch_log.append([phase, steps])
        return [a.elapsed_time(b) for a, b in ev]

    def mem_floor(self, launches=4):
        """The headline kernel's SYNTHETIC TWIN, measured (the packed sparse-protocol VonMises3D launch only): the launches of the timed
        region issued as evaluate_twin_kernel (context option "twin_masks", csrc/fcamd_kernels.hip) -- the same loads and stores at
        the same addresses on the same buffers, every tile's plastic ballot read from a recording of the real step, no constitutive
        arithmetic: what the memory system alone takes for the step's request stream.  Returns the average ms, or None."""
        if not (self.packed and self.sparse and self.kind == "von_mises_3d" and self.hmask is not None):
            return None
        torch = self.torch
        ctx = self.law._handle(self.device.index or 0).ctx
        recorded = []
        for i in (0, 1):  # the ballots of the two alternating iterates, as the real kernel leaves them in the mask array
            self.launch(i)
            torch.cuda.synchronize()
            recorded.append(self.hmask.clone())
        self.launch_log.append(["mem_floor_record", 2])
        self.recorded_ballots = recorded  # (in_place_line_bytes: the same step's plastic sets)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches + 2)]
        try:
            for i, (a, b) in enumerate(ev):
                ctx.set_option("twin_masks", recorded[i & 1].data_ptr())
                a.record()
                self.launch(i)
                b.record()
            torch.cuda.synchronize()
        finally:
            ctx.set_option("twin_masks", 0)
        self.launch_log.append(["mem_floor_twin", launches + 2])
        if not torch.equal(self.hmask, recorded[(launches + 1) & 1]):
            return None  # the twin did not reproduce the step's ballots: no figure
        ms = [a.elapsed_time(b) for a, b in ev][2:]
        self.launch(0), self.launch(1)  # real launches again: the trial state is meaningful where the protocol touches it
        self.launch_log.append(["mem_floor_resync", 2])
        return sum(ms) / len(ms)

    def in_place_line_bytes(self):
        """What the reference's in-place call MUST move at the memory system's granularity, from the step's recorded plastic sets
        (mem_floor): the dense streams as they are, the 48-byte plastic-strain rows of the plastic points as the distinct 128-byte lines
        their 64-byte granules lie in (reads; every read request of the L2 is one line) and as distinct 64-byte granules (writes), alpha
        read whole and written per touched tile.  Average of the two alternating iterates.  (if the measured bytes of
        the in-place call equal these, every line it fetches is needed -- the 3 % to the packed layout are the layout's.)"""
        rec = getattr(self, "recorded_ballots", None)
        if rec is None or self.kind != "von_mises_3d":
            return None
        t = self.torch
        shifts = t.arange(64, device=self.device, dtype=t.int64)
        reads = writes = 0.0
        for mask in rec:
            lines = granules = tiles = 0
            for lo in range(0, mask.numel(), 1 << 16):  # bounded temporaries
                part = mask[lo: lo + (1 << 16)]
                bits = ((part[:, None] >> shifts[None, :]) & 1).bool()
                tiles += int(bits.any(dim=1).sum())
                pts = bits.nonzero()
                p = (pts[:, 0] + lo) * 64 + pts[:, 1]  # plastic points of this part (ascending)
                g = t.cat([(48 * p) // 64, (48 * p + 47) // 64])  # the granules a row lies in (one or two)
                g = t.unique(g)
                granules += int(g.numel())
                lines += int(t.unique(g // 2).numel())
            n = self.n
            reads += n * (72 + 48 + 8) + 128.0 * lines
            writes += n * (48 + 288) + 64.0 * granules + 512.0 * tiles
        return {"read_bytes": reads / len(rec), "write_bytes": writes / len(rec)}

    def timed_in_place(self, steps, phase="in_place", sets=1):
        """The reference's own call, `law.evaluate(t, del_t, grad, stress, tangent, history)` IN PLACE on arrays in the interface's
        layout (models/interfaces.py:82-101) -- what a drop-in caller with device tensors launches.  The committed state is copied
        into the call's arrays before every launch, outside the event bracket (the call overwrites them); one warm launch, then
        `steps` event-timed ones.  Returns the kernel times in ms."""
        torch = self.torch
        ref = self.reference_history()
        results, keep = [], []
        for _ in range(max(1, sets)):  # the call's arrays are new allocations: like every row, the faster of two draws is reported
            s = torch.empty_like(self.stress_c)
            h = None if ref is None else {k: torch.empty_like(v) for k, v in ref.items()}
            keep.append((s, h))  # alive together, so that the second draw is other memory
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps + 1)]
            for i, (a, b) in enumerate(ev):
                s.copy_(self.stress_c)
                for k in (h or {}):
                    h[k].copy_(ref[k])
                a.record()
                self.law.evaluate(0.0, self.del_t, self.grads[i & 1], s, self.tangent, h)
                b.record()
            torch.cuda.synchronize()
            self.launch_log.append(["in_place_warm", 1])
            self.launch_log.append([phase, steps])
            results.append([a.elapsed_time(b) for a, b in ev[1:]])
        self.in_place_draws_ms = [round(sum(r) / len(r), 4) for r in results]
        return min(results, key=lambda r: sum(r))

    def ever_fraction(self):
        """share of the points whose plastic-strain row is in the committed EVER set (packed layout): what the packed runs hold
        before the timed steps -- the warm increment's plastic set, not an empty one"""
        if not self.packed or self.ever_c is None:
            return None
        t = self.torch
        shifts = t.arange(64, device=self.device, dtype=t.int64)
        total = 0
        for part in self.ever_c.split(1 << 18):  # bounded temporaries
            total += int(((part[:, None] >> shifts[None, :]) & 1).sum())
        return round(total / self.n, 4)

    def config_text(self):
        return (f"{self.name}: {self.kind} FULL-3D, {self.n} quadrature points per GPU, device-resident AoS, "
                f"committed->trial evaluate of two alternating Newton iterates"
                f"{', sparse trial history (ResidentState protocol)' if self.sparse else (', full trial history' if self.plasticity else '')}"
                f"{', plastic-strain rows of both state copies packed per tile (ResidentState default; commit = pointer swap)' if self.packed else ''}"
                f"{', history kept as [scalar, eps_p rows] in the state (split history)' if self.split else ''}"
                f"{', sparse tangent (rows of points that stay elastic are not rewritten)' if self.sparse_tangent else ''}")

    def free(self):
        for k in ("grads", "stress_c", "stress_t", "hist_c", "hist_t", "tangent", "hmask", "_vmm", "_plain", "ever_c", "ever_t"):
            setattr(self, k, None)  # a VMM working set is released with its last view
        self.torch.cuda.empty_cache()


def traffic_key(wl):
    """key of a workload's PMC measurement in profiles/traffic.json: the packed layout is the default of every law that has it"""
    unpacked = wl.sparse and wl.rows_key is not None and not wl.packed
    return wl.name + ("_full" if wl.plasticity and not wl.sparse else "") + ("_unpacked" if unpacked else "")


def placement_fracs(wl, alg0):
    """Roofline fraction of the first (untuned), median, worst and chosen tangent candidate (iterate 0)."""
    if not wl.placement:
        return {}
    ms = wl.placement["candidate_ms"]
    frac = lambda t: round(alg0 / (t * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)  # noqa: E731
    srt = sorted(ms)
    med = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    out = {"frac_first_allocation": frac(ms[0]), "frac_median_candidate": frac(med), "frac_worst_candidate": frac(srt[-1]),
           "frac_best_candidate": frac(srt[0])}
    if wl.vmm_info and "vmm_ms" in wl.vmm_info:
        out["frac_vmm_set"] = frac(wl.vmm_info["vmm_ms"])
    return out
