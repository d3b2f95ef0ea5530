#!/usr/bin/env python3
"""Benchmark of the hot path: quadrature-point stress updates per second.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--n POINTS]

One "step" = one ``evaluate`` pass of one constitutive law over n synthetic quadrature points
per GPU, device-resident (inputs already in HBM when the timed region starts), committed state in,
trial state out -- the call of the product's device-resident Newton loop (ResidentState.evaluate;
for the plasticity laws with the sparse trial-history protocol, --history full for the mask-less
form).  The steps alternate between two Newton iterates of the increment.  Headline workload:
BASELINE.json's target configuration, VonMises3D return mapping, 1e8 points, mixed elastic/plastic
(SURVEY.md 8d cfg3 "mixed").  Before the warm-up the placement of the tangent array is chosen out of a
few candidate allocations -- the line carries the chosen, the FIRST (what an untuned
caller gets) and the median candidate.

With no --workload (the driver's command) and N = 1 the same run then times every other single-GPU
configuration of BASELINE.json / SURVEY.md 8d with the same method -- LinearElasticity (cfg2),
VonMises3D all-plastic and all-elastic (cfg3 sub-cases), Maxwell (cfg4), Kelvin -- and reports them
under "configs".  For N > 1 (launched by torch.distributed.run, one rank per GPU) every rank evaluates
its own contiguous shard of n points (weak scaling, no data-path collective: SURVEY.md 8e); the
stress/tangent all-gather of the single-assembler mode (config 5) is timed separately on the whole
shard -- stress in one piece, tangent in chunks that fit next to the working set -- and reported under
"allgather", never inside `value`.

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").  `roofline.traffic` (HBM bytes per launch by the PMC counters) is
measured at the end of an N = 1 run by two child passes of this file under `rocprofv3 --pmc` (--no-live-traffic: the stored
figure of profiles/traffic.json instead).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
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


class LineGuard:
    """The ONE JSON line survives the death of the process that measured it.

    The timed steps are over long before the optional legs (all-gather variants, extra configurations, host path, CPU
    baselines) are; a leg that takes the process down on hardware nobody could rehearse (a fault in a peer mapping, the
    launcher terminating rank 0 because another rank died) must not cost the measured line.  A small child process
    (started BEFORE anything touches the GPU, no GPU use of its own, a session of its own) reads lines from a pipe and
    prints the LAST one it received when the pipe closes: rank 0 sends the line as it stands before every optional leg
    (marked "incomplete": the leg it was about to enter) and the complete line at the end.  Exactly one line reaches
    stdout either way; the exit code of the job still says that a leg failed."""

    CHILD = ("import sys\nlast = None\nfor line in sys.stdin:\n    if line.strip():\n        last = line\n"
             "if last is not None:\n    sys.stdout.write(last if last.endswith('\\n') else last + '\\n')\n    sys.stdout.flush()\n")

    def __init__(self):
        import subprocess

        self.proc = None
        # a plain interpreter: under rocprofv3 the parent's environment preloads the profiler, which would initialise the GPU
        # (and claim counters) in the guard as well
        env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS"))}
        try:
            self.proc = subprocess.Popen([sys.executable, "-c", self.CHILD], stdin=subprocess.PIPE, text=True, start_new_session=True, env=env)
        except OSError:
            pass  # no guard: the line is printed directly at the end

    def _send(self, line):
        self.proc.stdin.write(line + "\n")
        self.proc.stdin.flush()

    def provisional(self, out, leg):
        if self.proc is None:
            return
        try:
            self._send(json.dumps(dict(out, incomplete=f"the process ended in the optional leg '{leg}'; everything in this line was measured before it")))
        except (OSError, ValueError):
            self.proc = None

    def final(self, line):
        if self.proc is not None:
            try:
                self._send(line)
                self.proc.stdin.close()
                self.proc.wait(timeout=30)
                return
            except Exception:
                pass
        print(line, flush=True)


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
        self.grads.append(self.grads[0] if os.environ.get("BENCH_SINGLE_ITERATE") == "1" else self.grads[0] * 1.03)  # knob: A/B only
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


def run_config(name, n, seed, device, dev_index, steps, warmup, tries, history="packed", placement="auto", cpu=True):
    """One extra configuration, same method as the headline: placement, warm up, count, >= 5 event-timed launches."""
    wl = Workload(name, n, seed, device, dev_index, history=history)
    try:
        wl.place(placement, tries)
        wl.warmup(warmup)
        wl.count_plastic()
        ms = wl.timed_events(steps)
        n_pl, n_its = wl.mean_plastic(steps)
        alg = wl.alg_bytes(n_pl)
        avg = sum(ms) / len(ms)
        out = {"baseline_config": BASELINE_CONFIG.get(name), "workload": wl.config_text(), "launches": steps,
               "kernel_ms_avg": round(avg, 4), "kernel_ms_min": round(min(ms), 4),
               "Mpts_s": round(n / (avg * 1e-3) / 1e6, 1), "algorithmic_bytes_per_launch": alg,
               "achieved_GBs": round(alg / (avg * 1e-3) / 1e9, 1), "frac": round(alg / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "plastic_fraction": round(n_pl / n, 4),
               "mean_newton_iters": round(n_its / max(n_pl, 1), 3) if wl.kind in ("von_mises_3d", "comfe_drucker_prager") else None,
               "bytes_per_point": {"elastic": wl.b_el, "plastic": wl.b_pl}}
        out.update(placement_fracs(wl, wl.alg_bytes(wl.n_pl_ab[0])))
        if wl.placement:
            out["placement_candidate_ms"] = wl.placement["candidate_ms"]
        # PMC-measured HBM bytes of this configuration's launch (profiles/traffic.json: its own rocprofv3 passes, same kernels)
        key = traffic_key(wl)
        out["traffic"] = read_traffic(key, n)
        out["traffic_source"] = None if not out["traffic"] else f"profiles/traffic.json[{key}] (stored rocprofv3 --pmc measurement of these kernels)"
        out["traffic_over_algorithmic"] = None if not out["traffic"] else round(out["traffic"] / alg, 4)
        out["placement_mode"] = (wl.vmm_info or {}).get("mode", "hipmalloc_tuned" if wl.placement else "first")
        if wl.vmm_info and "vmm_ms" in wl.vmm_info:
            out["placement_vmm_ms"] = wl.vmm_info["vmm_ms"]
        out["launch_log"] = wl.launch_log
        if cpu:
            try:
                out["cpu"] = cpu_quick(wl)
            except Exception as e:  # informational
                out["cpu"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        return out
    finally:
        wl.free()


def cpu_quick(wl, budget_s=1.0, ns=1_000_000):
    """A short CPU figure for one configuration: the C port (oracle/oracle.c, serial loop, 1 thread) and the NumPy
    restatement of the reference's code path on the first `ns` points of the configuration's own arrays."""
    import numpy as np

    from fenics_constitutive_amd.hostio import to_host
    from oracle import c_oracle as CO
    from oracle import numpy_oracle as NO

    ns = min(ns, wl.n)
    dims = {"eps_n": 6, "alpha": 1, "strain_visco": 6, "strain": 6, "history": 7}
    g = to_host(wl.grads[0][: 9 * ns])
    s0 = to_host(wl.stress_c[: 6 * ns])
    h0 = None if wl.hist_c is None else {k: to_host(v[: dims[k] * ns]) for k, v in wl.reference_history().items()}
    tan = np.zeros(36 * ns)
    out = {}
    for label, fn, m in (("c_port_1_thread_Mpts_s", CO.MODELS[wl.kind], ns), ("numpy_port_Mpts_s", NO.MODELS[wl.kind], min(ns, 200_000))):
        def one_pass():
            s = s0[: 6 * m].copy()
            h = None if h0 is None else {k: v[: dims[k] * m].copy() for k, v in h0.items()}
            t0 = time.perf_counter()
            fn(wl.params, 0.0, wl.del_t, g[: 9 * m], s, tan[: 36 * m], h)
            return time.perf_counter() - t0

        one_pass()  # untimed: faults in the pages of the output arrays
        reps, tt = 0, 0.0
        while tt < budget_s and reps < 50:
            tt += one_pass()
            reps += 1
        out[label] = round(m * reps / tt / 1e6, 2)
    out["sample"] = f"first {ns} points of this configuration's arrays, ~{budget_s:.0f} s each"
    return out


def cpu_baseline(kind, params, grad, stress, hist, del_t, budget_s=8.0):
    """Time the C oracle ("port": serial per-point loop, 1 thread -- what the reference does per
    MPI rank) on a bounded sample of the same workload."""
    import numpy as np

    from fenics_constitutive_amd.hostio import to_host
    from oracle import c_oracle as CO

    ns = min(grad.numel() // 9, 2_000_000)
    g = to_host(grad[: 9 * ns])
    s0 = to_host(stress[: 6 * ns])
    dims = {"eps_n": 6, "alpha": 1, "strain_visco": 6, "strain": 6, "history": 7}
    h0 = None if hist is None else {k: to_host(v[: dims[k] * ns]) for k, v in hist.items()}
    tan = np.zeros(36 * ns)
    fn = CO.MODELS[kind]

    def one_pass():
        s = s0.copy()
        h = None if h0 is None else {k: v.copy() for k, v in h0.items()}
        t0 = time.perf_counter()
        fn(params, 0.0, del_t, g, s, tan, h)
        return time.perf_counter() - t0

    one_pass()  # untimed: faults in the pages of the output arrays
    reps, t_total = 0, 0.0
    while t_total < budget_s and reps < 500:
        t_total += one_pass()
        reps += 1
    out = {
        "value": round(ns * reps / t_total / 1e6, 3),
        "unit": "Mpts/s",
        "cores": 1,
        "kind": "port",
        "sample": f"oracle/oracle.c serial loop ({CO.build_flags()}), first {ns} points of the headline workload x {reps} passes ({t_total:.1f} s)",
    }
    # the reference's own NumPy code path, restated (oracle/numpy_oracle.py): a few seconds, for scale
    try:
        from oracle import numpy_oracle as NO

        def time_np(fn, m):
            s = s0[: 6 * m].copy()
            h = None if h0 is None else {k: v[: dims[k] * m].copy() for k, v in h0.items()}
            t0 = time.perf_counter()
            fn(params, 0.0, del_t, g[: 9 * m], s, tan[: 36 * m], h)
            return round(m / (time.perf_counter() - t0) / 1e6, 4)

        extra = {"numpy_port_Mpts_s": time_np(NO.MODELS[kind], min(ns, 100_000 if kind == "comfe_drucker_prager" else 500_000)),
                 "threads": "NumPy/OpenBLAS default"}
        if kind == "von_mises_3d":
            extra["python_per_point_loop_port_Mpts_s"] = time_np(NO.von_mises_3d_loop, min(ns, 20_000))
            # BASELINE config 3 compares with the comfe-rs CPU path: our C restatement of the serial
            # evaluate_model loop around MisesPlasticity3D (interfaces.rs:354-456, mises_plasticity.rs:58-126;
            # mu, kappa, y_0 as above, h = 200 as in tests/models/test_plasticity.py:26-31) on the same
            # gradients and stresses, 1 thread
            hr = np.zeros(7 * ns)
            hr.reshape(-1, 7)[:, 0] = h0["alpha"]
            rs_p = {"mu": params["p_mu"], "kappa": params["p_ka"], "y_0": params["p_y0"], "h": 200.0}
            tt, rr = 0.0, 0
            while tt < 1.5 and rr < 100:
                s, hh = s0.copy(), {"history": hr.copy()}
                t0 = time.perf_counter()
                CO.MODELS["comfe_mises_plasticity"](rs_p, 0.0, del_t, g, s, tan, hh)
                tt += time.perf_counter() - t0
                rr += 1
            extra["comfe_rs_mises_c_port_1_thread_Mpts_s"] = round(ns * rr / tt / 1e6, 2)
        # BASELINE configs[0]: LinearElasticityModel FULL-3D, 1e5 points, the reference's NumPy evaluate() on the CPU --
        # here its NumPy restatement (and the C port) on the SURVEY 8d cfg1 inputs (grad ~ N(0, 1e-3^2), sigma = 0, E = 42, nu = 0.3, seed 0)
        rng = np.random.default_rng(0)
        g0, t0_ = rng.normal(scale=1e-3, size=9 * 100_000), np.zeros(36 * 100_000)
        for label, f0 in (("config0_le_1e5_numpy_port_Mpts_s", NO.MODELS["linear_elasticity"]), ("config0_le_1e5_c_port_Mpts_s", CO.MODELS["linear_elasticity"])):
            best = None
            for _ in range(5):
                s_ = np.zeros(6 * 100_000)
                tq = time.perf_counter()
                f0(LE_P, 0.0, 1.0, g0, s_, t0_, None)
                dq = time.perf_counter() - tq
                best = dq if best is None else min(best, dq)
            extra[label] = round(0.1 / best, 2)
        # the same C loop on all host cores (OpenMP over points), for scale only
        nthr = min(CO.max_threads(), os.cpu_count() or 1)
        CO.set_num_threads(nthr)
        one_pass()
        tt, rr = 0.0, 0
        while tt < 1.5 and rr < 200:
            tt += one_pass()
            rr += 1
        CO.set_num_threads(1)
        extra["c_port_all_cores_Mpts_s"] = round(ns * rr / tt / 1e6, 1)
        extra["c_port_all_cores_threads"] = nthr
        out["extra"] = extra
    except Exception as e:  # the extra figures are informational only
        out["extra"] = {"error": str(e)}
    try:
        out["small_call_crossover"] = small_call_crossover()
    except Exception as e:  # informational
        out["small_call_crossover"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    return out


def small_call_crossover(sizes=(64, 256, 1024, 4096, 16384, 65536), reps=7):
    """Per law: the number of points below which ONE ndarray ``evaluate`` call on the GPU (launch + PCIe round trips: a floor
    of tens of microseconds) loses to the NumPy restatement of the reference's own code path on this box's host -- what a
    dolfinx rank with a few thousand quadrature points per law pays (solver/_lawonsubmesh.py:86-94).  Medians of `reps`
    calls per size; the crossover is interpolated between the two sizes where the order flips.  DeviceLaw.evaluate warns
    once below `device.SMALL_CALL_POINTS` (INTEGRATION.md states the measured table)."""
    import numpy as np

    import fenics_constitutive_amd as fc
    from fenics_constitutive_amd import device as fdev
    from oracle import numpy_oracle as NO

    FULL = fc.StressStrainConstraint.FULL
    rng = np.random.default_rng(5)
    cases = {"linear_elasticity": (fc.LinearElasticityModel(LE_P, FULL), LE_P, None, 1e-3),
             "von_mises_3d": (fc.VonMises3D(VM_P), VM_P, {"eps_n": 6, "alpha": 1}, 3e-3),
             "spring_maxwell": (fc.SpringMaxwellModel(SLS_P, FULL), SLS_P, {"strain_visco": 6, "strain": 6}, 1e-3),
             "spring_kelvin": (fc.SpringKelvinModel(SLS_P, FULL), SLS_P, {"strain_visco": 6, "strain": 6}, 1e-3)}
    out = {"sizes": list(sizes), "unit": "us per call (median)", "warn_below_points": dict(fdev.SMALL_CALL_POINTS)}
    import warnings

    for kind, (law, params, hd, scale) in cases.items():
        gpu_us, np_us = [], []
        for n in sizes:
            g = rng.normal(scale=scale, size=9 * n)
            s0 = rng.normal(size=6 * n)
            h0 = None if hd is None else {k: np.abs(rng.normal(scale=1e-3, size=d * n)) for k, d in hd.items()}
            t = np.zeros(36 * n)

            def run(fn, is_law):
                ts = []
                for _ in range(reps + 2):
                    s = s0.copy()
                    h = None if h0 is None else {k: v.copy() for k, v in h0.items()}
                    t0 = time.perf_counter()
                    if is_law:
                        fn.evaluate(0.0, 2.0, g, s, t, h)
                    else:
                        fn(params, 0.0, 2.0, g, s, t, h)
                    ts.append(time.perf_counter() - t0)
                return sorted(ts[2:])[reps // 2] * 1e6

            with warnings.catch_warnings():
                warnings.simplefilter("ignore")  # the very warning this table calibrates
                gpu_us.append(round(run(law, True), 1))
            np_us.append(round(run(NO.MODELS[kind], False), 1))
        cross = None
        for k in range(len(sizes)):
            if gpu_us[k] <= np_us[k]:
                if k == 0:
                    cross = sizes[0]
                else:  # linear interpolation of the difference between the two sizes
                    d0, d1 = gpu_us[k - 1] - np_us[k - 1], gpu_us[k] - np_us[k]
                    cross = int(sizes[k - 1] + (sizes[k] - sizes[k - 1]) * d0 / (d0 - d1)) if d0 != d1 else sizes[k]
                break
        out[kind] = {"gpu_call_us": gpu_us, "numpy_port_us": np_us, "crossover_points": cross}
    return out


def host_path_figures(devices=None, sizes=(1_000_000, 10_000_000), latency_sizes=(1_000, 10_000), reps=3, budget_s=12.0,
                      seed=99):
    """PCIe-inclusive figures of the HOST entries -- the call the reference times with Timer("constitutive-law-evaluation")
    (solver/_lawonsubmesh.py:86-94: evaluate on views of Function.x.array) -- VonMises3D, mixed elastic / plastic NumPy arrays:
      evaluate            the reference contract: law.evaluate(ndarrays) in place (fcamd_evaluate_host), 176 B/pt up, <= 392 down;
      resident            ResidentState.evaluate_into (fcamd_evaluate_resident): state on the device, 72 B/pt up, 336 down;
      resident_sparse     the same with the sparse tangent (the product default): only the tangent rows of plastic / formerly
                          plastic points cross PCIe from the second call on;
    each with pageable arrays (page-locked by the library for the duration of the call) and with arrays registered once.
    `devices` = list of device ordinals: the single-process multi-GPU form of the same calls (fcamd_multi: every device on its
    own slice over its own PCIe link); None: one device, the plain objects.  Never part of `value` of the default line."""
    import numpy as np

    import fenics_constitutive_amd as fc
    from fenics_constitutive_amd import _capi

    t_begin = time.perf_counter()
    multi = devices is not None
    n_max = max(sizes)
    rng = np.random.default_rng(seed)
    law = fc.VonMises3D(VM_P)
    if multi:
        law.use_devices(devices)
    g = rng.standard_normal(9 * n_max)
    g *= np.repeat(10.0 ** (rng.random(n_max) * 2.0 - 4.0), 9)
    s0 = np.zeros(6 * n_max)
    a0 = rng.random(n_max) * 0.02
    s, t = np.zeros(6 * n_max), np.zeros(36 * n_max)
    e, a = np.zeros(6 * n_max), a0.copy()
    out = {"law": "VonMises3D, grad scale log-uniform in [1e-4, 1e-2], alpha ~ U(0, 0.02)", "devices": devices or [_capi.default_device()],
           "bytes_per_point": {"evaluate_up": 176, "evaluate_down_max": 392, "resident_up": 72, "resident_down": 336}, "sizes": {}}

    def make_state(n):
        if multi:
            from fenics_constitutive_amd.multidevice import MultiDeviceResidentState

            return [MultiDeviceResidentState(fc.VonMises3D(VM_P), n, devices=devices, history0={"eps_n": e[: 6 * n], "alpha": a0[:n]},
                                             sparse_tangent=sp) for sp in (False, True)]
        from fenics_constitutive_amd.resident import ResidentState

        return [ResidentState(law, n, history0={"eps_n": e[: 6 * n], "alpha": a0[:n]}, sparse_tangent=sp, placement="torch") for sp in (False, True)]

    def best_of(fn, k, reset=None):
        best = None
        for _ in range(k):
            if reset:
                reset()
            t0 = time.perf_counter()
            fn()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best

    def figures(n, k, lat=False):
        gs, ss, ts, es, al = g[: 9 * n], s[: 6 * n], t[: 36 * n], e[: 6 * n], a[:n]
        full, sparse = make_state(n)

        def reset():
            ss[:] = 0.0
            es[:] = 0.0
            al[:] = a0[:n]

        row = {}
        legs = (("evaluate", lambda: law.evaluate(0.0, 1.0, gs, ss, ts, {"eps_n": es, "alpha": al}), reset, 568),
                ("resident", lambda: full.evaluate_into(0.0, 1.0, gs, ss, ts), None, 408),
                ("resident_sparse", lambda: sparse.evaluate_into(0.0, 1.0, gs, ss, ts), None, 408))
        for name, fn, rs_, bpp in legs:
            fn()  # warm: first touch, first page lock, (sparse) the full tangent
            dt = best_of(fn, k, rs_)
            if lat:
                row[name + "_us"] = round(dt * 1e6, 1)
            else:
                row[name] = {"ms": round(dt * 1e3, 3), "Mpts_s": round(n / dt / 1e6, 1), "interface_GBs": round(n * bpp / dt / 1e9, 2)}
        if not lat:
            row["plastic_fraction"] = round(law.last_stats.n_plastic / n, 4)
        for st in (full, sparse):
            if multi:
                st.close()
        return row

    pin_target = None
    try:
        for registered in (False, True):
            if registered:
                if multi:
                    pin_target = law._multi()
                else:
                    pin_target = law._handle(_capi.default_device()).ctx
                for x in (g, s, t, e, a):
                    pin_target.register_host_buffer(x)
            key = "registered" if registered else "pageable"
            for n in sizes:
                if time.perf_counter() - t_begin > budget_s and n != min(sizes):
                    out["sizes"].setdefault(str(n), {})[key] = "skipped: time budget"
                    continue
                out["sizes"].setdefault(str(n), {})[key] = figures(n, reps)
            for n in latency_sizes:
                out.setdefault("per_call_us", {}).setdefault(str(n), {})[key] = figures(n, 30, lat=True)
        # what the link gives a plain copy between the registered tangent array and device memory
        import torch

        from fenics_constitutive_amd.hostio import download, upload

        dev = torch.device("cuda", (devices or [_capi.default_device()])[0])
        m = min(n_max, 4_000_000)
        buf = torch.empty(36 * m, dtype=torch.float64, device=dev)
        upload(buf, t[: 36 * m])
        h2d = best_of(lambda: upload(buf, t[: 36 * m]), 3)
        d2h = best_of(lambda: download(t[: 36 * m], buf), 3)
        out["pinned_copy_GBs"] = {"h2d": round(288 * m / h2d / 1e9, 1), "d2h": round(288 * m / d2h / 1e9, 1),
                                  "note": "fcamd_copy_to_device / _to_host of 36 doubles x %d points between the registered tangent array and one device" % m}
        big = out["sizes"].get(str(n_max), {}).get("registered")
        if isinstance(big, dict):
            down = n_max * 392 / (big["evaluate"]["ms"] * 1e-3) / 1e9
            out["evaluate_d2h_over_pinned_copy"] = round(down / (out["pinned_copy_GBs"]["d2h"] * (len(devices) if multi else 1)), 3)
    finally:
        if pin_target is not None:
            for x in (g, s, t, e, a):
                try:
                    pin_target.unregister_host_buffer(x)
                except Exception:
                    pass
    out["wall_s"] = round(time.perf_counter() - t_begin, 1)
    return out


def main_host(args):
    """--mode host: the single-process multi-GPU host path (fcamd_multi).  ONE process drives --gpus devices; under
    torch.distributed.run every rank but 0 leaves at once (nothing on this path needs a process group)."""
    rank = int(os.environ.get("RANK", "0"))
    if rank != 0:
        return 0
    import torch

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    have = torch.cuda.device_count()
    if args.host_devices:
        devices = [int(x) for x in args.host_devices.split(",")]
    else:
        devices = [k % have for k in range(args.gpus)]  # fewer GPUs than asked for: contexts share devices (rehearsal)
    n_total = args.n * len(devices) if args.scaling == "weak" else args.n
    t_start = time.perf_counter()
    import numpy as np

    import fenics_constitutive_amd as fc
    from fenics_constitutive_amd import _capi

    fig = host_path_figures(devices=devices, sizes=(min(1_000_000, n_total), n_total), latency_sizes=(1_000, 10_000),
                            reps=max(2, min(args.steps, 5)), budget_s=args.wall_budget / 2)
    # the timed steps proper: the reference contract (in-place evaluate on pageable NumPy arrays) over all devices
    rng = np.random.default_rng(5)
    law = fc.VonMises3D(VM_P).use_devices(devices)
    g = rng.standard_normal(9 * n_total)
    g *= np.repeat(10.0 ** (rng.random(n_total) * 2.0 - 4.0), 9)
    a0 = rng.random(n_total) * 0.02
    s, t, e, a = np.zeros(6 * n_total), np.zeros(36 * n_total), np.zeros(6 * n_total), a0.copy()

    def step():
        law.evaluate(0.0, 1.0, g, s, t, {"eps_n": e, "alpha": a})

    for _ in range(args.warmup):
        step()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    elapsed = time.perf_counter() - t0
    mode, used = law._multi().last_host_mode()
    n_pl = int(law.last_stats.n_plastic)
    bytes_step = n_total * 176 + (n_total - n_pl) * 336 + n_pl * 392
    out = {"metric": METRIC, "value": round(n_total * args.steps / elapsed / 1e6, 1), "unit": "Mpts/s", "n_gpus": len(devices),
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
           "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic", "mode": "host",
           "config": {"workload": f"host path: VonMises3D FULL-3D, {n_total} quadrature points in ONE process's pageable NumPy arrays, in-place "
                                  f"evaluate (the reference contract) spread over {len(devices)} device contexts by fcamd_multi_evaluate_host -- every "
                                  f"device on its own slice over its own PCIe link, no gather; PCIe-inclusive by construction",
                      "points_total": n_total, "devices": devices, "devices_used": used, "host_mode_flags": mode,
                      "plastic_fraction": round(n_pl / n_total, 4), "parallelism": f"one process x {len(devices)} device contexts"},
           "roofline": {"bound": "pcie", "achieved": round(bytes_step * args.steps / elapsed / 1e9, 2), "unit": "GB/s",
                        "peak": None if "pinned_copy_GBs" not in fig else round((fig["pinned_copy_GBs"]["h2d"] + fig["pinned_copy_GBs"]["d2h"]) * len(set(devices)), 1),
                        "frac": None, "traffic": None,
                        "note": "achieved = interface bytes over PCIe per step (176 B/pt up; 336 down for elastic, 392 for plastic points) / step time, "
                                "both directions counted; peak = measured pinned H2D + D2H copy rate of one link x distinct devices"},
           "host_path": fig, "cpu_baseline": None, "library": {"srchash": library_hash(), "kernel_hash": library_hash(kernels_only=True)}}
    if out["roofline"]["peak"]:
        out["roofline"]["frac"] = round(out["roofline"]["achieved"] / out["roofline"]["peak"], 4)
    out["wall_s"] = round(time.perf_counter() - t_start, 1)
    print(json.dumps(out), flush=True)
    return 0


def time_allgather(args, dist, torch, device, rank, world, n, stress_t, tangent, agree=None, budget_left=None):
    """The exchange step of the single-assembler mode (BASELINE config 5), timed separately
    and never part of `value`: every rank's stress slice (6/pt) in one piece and its tangent slice (36/pt)
    in chunks through two chunk buffers that are sized against the free device memory up front
    (fcamd_gather_chunk_plan) -- at 8 x 1e8 points the gathered tangent alone would be 230 GB.
      rccl_*    in-place all_gather_into_tensor (RCCL);
      direct_*  the C ABI's peer copies (fcamd_allgather_direct on IPC-mapped buffers): world-1 concurrent
                copies per rank, one per xGMI link;
      p2p_*     (--gather-direct) one batched isend/irecv group to all peers (RCCL point-to-point)."""
    from fenics_constitutive_amd.sharded import ChunkedGather, PeerBuffers, ShardedEvaluator, shared_empty

    ng = ((min(args.gather_points, n) if args.gather_points > 0 else n) // 64) * 64  # whole tiles: every slot is full
    if ng == 0:
        raise ValueError("fewer than 64 points per rank: nothing to gather")
    ev = ShardedEvaluator(None, ng * world)
    per = ev.plan.per_rank
    assert per == ng == ev.n_local  # whole tiles: every slot is full
    shard_bytes = 42 * 8 * ng
    nccl = args.backend == "nccl"
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info(device)
    # every rank must derive the SAME chunk plan (the chunks are collectives): the smallest free memory of all ranks
    fr = torch.tensor([free], dtype=torch.int64, device=device if nccl else "cpu")
    dist.all_reduce(fr, op=dist.ReduceOp.MIN)
    free = int(fr.item()) // (1 if nccl else world)  # gloo rehearsal: the ranks share one GPU
    reserve = 8 << 30
    out_s_bytes = 6 * per * world * 8
    budget = free - reserve - out_s_bytes
    if budget <= 0:  # the same on every rank: nobody enters a collective
        raise MemoryError(f"{free / 1e9:.1f} GB free: no room for the gathered stress ({out_s_bytes / 1e9:.1f} GB) + {reserve >> 30} GiB reserve")
    out_s = shared_empty(6 * per * world, device)  # mapped by the peers (direct variant): an IPC-safe allocation
    s_mine = out_s[6 * per * rank : 6 * per * rank + 6 * ng]
    s_mine.copy_(stress_t[: 6 * ng])
    t_mine = tangent[: 36 * ng]
    result = {"points_per_rank": ng, "shard_GB": round(shard_bytes / 1e9, 3), "free_GB_before": round(free / 1e9, 1),
              "note": "stress gathered whole (in place), tangent through 2 chunk buffers sized against free memory; outside the timed steps"}

    def timed(fn, reps=2):
        best = None
        for _ in range(reps):
            torch.cuda.synchronize()
            dist.barrier()
            t_ = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            dist.barrier()
            dt_ = time.perf_counter() - t_
            best = dt_ if best is None else min(best, dt_)
        tt_ = torch.tensor([best], dtype=torch.float64, device=device if nccl else "cpu")
        dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
        return float(tt_.item())

    def report(prefix, t):
        result[prefix + "_ms"] = round(t * 1e3, 3)
        result[prefix + "_recv_GBs_per_gpu"] = round(shard_bytes * (world - 1) / t / 1e9, 1)

    for variant in (["rccl"] if nccl else []) + ["direct"] + (["p2p"] if (nccl and args.gather_direct) else []):
        peer = variant == "direct"
        if agree is not None and not agree(budget_left() > 75):  # every variant is a set of collectives: all ranks or none
            result[variant + "_skipped"] = "wall budget"
            continue
        if rank == 0:
            print(f"# allgather leg: {variant}, {ng} points per rank, budget {budget / 1e9:.1f} GB", file=sys.stderr, flush=True)
        try:  # set-up failures are raised on all ranks together (PeerBuffers exchanges the outcome of every step)
            cg = ChunkedGather(ev, 36, budget, like=tangent, peer_copies=peer)  # raises up front if the budget holds no tile
            peers_s = PeerBuffers(out_s) if peer else None
        except Exception as e:
            result[variant + "_error"] = f"{type(e).__name__}: {e}"[:300]
            torch.cuda.empty_cache()
            continue
        result["tangent_chunks"], result["chunk_points"] = cg.plan.n_chunks, cg.plan.chunk
        result["chunk_buffers_GB"] = round(2 * cg.plan.buffer_numel * 8 / 1e9, 2)

        def run():
            if peer:
                ev.allgather_peer(s_mine, out_s, 6, peers_s)
            elif variant == "p2p":
                ev.allgather_direct(s_mine, out_s, 6)
            else:
                ev.allgather(s_mine, out_s, 6)
            for _k, _view in cg.chunks(t_mine):
                pass  # the consumer (the assembler) would read _view here

        try:
            report(variant, timed(run))
        finally:
            if peers_s is not None:
                peers_s.close()
            cg.close()
            del cg
            torch.cuda.empty_cache()
    del out_s
    return result


def library_hash(kernels_only=False):
    """Content hash of the sources libfcamd.so was built from (fenics_constitutive_amd/_build.py); kernels_only:
    of the device code alone, which is what measured HBM traffic depends on."""
    try:
        from fenics_constitutive_amd import _build

        if kernels_only:
            return _build.built_kernel_hash()
        with open(_build.HASHFILE) as f:
            return f.read().strip()
    except Exception:
        return None


def read_traffic_split(workload_key, n):
    """(read bytes, written bytes) per launch of the same PMC measurement, or None"""
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(tf) as f:
            e = json.load(f).get(workload_key)
        if e and int(e.get("n", 0)) == n and e.get("kernel_hash") and e.get("kernel_hash") == library_hash(kernels_only=True):
            return int(e["read_bytes"]), int(e["write_bytes"])
    except Exception:
        pass
    return None


def read_traffic(workload_key, n):
    """PMC-measured HBM bytes per launch (profiles/traffic.json, written by tools/summarize_profile.py) --
    only if they were measured with THIS build of the kernels (same hash of the device sources) at this size; a
    kernel change makes the figure stale and the line then says null."""
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(tf) as f:
            e = json.load(f).get(workload_key)
        if e and int(e.get("n", 0)) == n and e.get("kernel_hash") and e.get("kernel_hash") == library_hash(kernels_only=True):
            return e.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def under_profiler():
    """is THIS process running under rocprofv3 (a nested profiler must not be started)"""
    return any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def live_traffic(name, n, history, extra, budget_s, frow=False):
    """HBM bytes per timed launch of this workload, measured NOW as MI355X_MICROARCH.md ("HBM", rocprofv3) prescribes: two
    child runs of this file (4 timed steps each, first-allocation placement -- the traffic of a launch does not depend on
    where its arrays lie) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes; no trace domain next
    to --pmc), the evaluate dispatches of the TIMED phase picked out with the child's own launch_log.  Corrections: both
    counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide (16 B per lane) streaming read -> x 2.
    None if rocprofv3 is not there, the budget is short or anything goes wrong (the stored figure is reported then)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if shutil.which("rocprofv3") is None or under_profiler():
        return None
    t_end = time.perf_counter() + budget_s
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        left = t_end - time.perf_counter()
        if left < 25:
            return None
        d = tempfile.mkdtemp(prefix="fcamd_pmc_", dir="/tmp")
        try:
            cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__)]
            if frow:  # one row of SURVEY 8(f) alone (bench_frows.py)
                cmd += ["--frow", name, "--points", str(n), "--steps", "4", "--warmup", "2"]
            else:
                cmd += ["--workload", name, "--points", str(n), "--history", history, "--steps", "4", "--warmup", "2", "--configs", "none",
                        "--no-host-path", "--no-cpu-baseline", "--placement", "first", "--no-live-traffic"] + list(extra)
            # a process group of its own: on a timeout the whole pass (profiler + the profiled child) is ended, nothing else
            p = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                 text=True, start_new_session=True)
            try:
                stdout, _ = p.communicate(timeout=left)
            except subprocess.TimeoutExpired:
                import signal

                os.killpg(p.pid, signal.SIGKILL)
                p.communicate()
                return None
            lines = [ln for ln in stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
            if p.returncode != 0 or not lines:
                return None
            child = json.loads(lines[-1])
            files = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
            if not files:
                return None
            with open(files[-1]) as f:
                rows = [x for x in csv.DictReader(f) if "fcamd::evaluate" in x["Kernel_Name"] and x.get("Counter_Name", counter) == counter]
            rows.sort(key=lambda x: int(x["Dispatch_Id"]))
            vals, i, timed = [float(x["Counter_Value"]) for x in rows], 0, []
            for phase, k in child.get("launch_log", []):
                if phase == "timed":
                    timed = vals[i: i + k]
                i += k
            if len(timed) != 4 or i > len(vals):
                return None
            got[counter] = sum(timed) / len(timed)
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    read_b, write_b = 2.0 * 1024.0 * got["FETCH_SIZE"], 1024.0 * got["WRITE_SIZE"]
    return {"hbm_bytes_per_launch": int(read_b + write_b), "read_bytes": int(read_b), "write_bytes": int(write_b)}


def all_agree(flag, dist, device):
    """the same yes / no on every rank (a leg with collectives must be entered by all ranks or by none): the minimum over the ranks"""
    import torch

    f = torch.tensor([1 if flag else 0], dtype=torch.int64, device=device)
    dist.all_reduce(f, op=dist.ReduceOp.MIN)
    return bool(int(f.item()))


def main_frow(args):
    """`--frow NAME`: one SURVEY 8(f) row alone; prints one JSON line (with the launch_log the PMC slicing needs)"""
    import torch

    import bench_frows

    if args.frow not in bench_frows.FROWS:
        sys.exit(f"--frow: one of {sorted(bench_frows.FROWS)}")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    torch.cuda.set_device(device)
    out = bench_frows.run_frow(args.frow, args.n, device, launches=max(1, args.steps), warm=max(1, args.warmup), peak_gbs=HBM_PEAK_GBS)
    print(json.dumps({"metric": METRIC, "frow": args.frow, **out}), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help=f"time this workload only (default: {HEADLINE}, followed at N = 1 by the other BASELINE configurations)")
    ap.add_argument("--frow", default=None,
                    help="run ONE row of SURVEY 8(f) alone (bench_frows.FROWS: indexed evaluate, fused wrapper, low-dimensional kernels, "
                         "resident sparse-tangent iteration) and print its figures -- the child of the default run's PMC passes")
    ap.add_argument("--no-frows", action="store_true", help="N = 1 default run: skip the SURVEY 8(f) rows after the BASELINE configurations")
    ap.add_argument("--configs", choices=["auto", "all", "none"], default="auto",
                    help="the other single-GPU BASELINE configurations after the headline: auto = when no --workload is given and N = 1")
    ap.add_argument("--config-steps", type=int, default=6, help="timed launches per extra configuration (>= 5)")
    ap.add_argument("--n", "--points", dest="n", type=int, default=None,
                    help="quadrature points per GPU (use --points under torch.distributed.run, whose parser claims --n); default 1e8, "
                         "--mode host: 1e7 (the arrays are host memory: 568 B per point)")
    ap.add_argument("--grid", type=int, default=0, help="override the launch grid (workgroups)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--history", choices=["packed", "sparse", "full"], default="packed",
                    help="plasticity laws: how the trial history is written.  packed (default) = the protocol of the "
                         "product's device-resident Newton loop (ResidentState): trial == committed except at plastic / "
                         "formerly plastic points, so elastic points cost no history traffic, and both copies of the "
                         "plastic-strain array keep the rows of the ever-plastic points packed per tile (FCAMD_EVAL_PACKED_HISTORY; "
                         "the commit stays a pointer swap); sparse = the same protocol on the reference's array layout "
                         "(ResidentState(packed_history=False)); full = every launch rewrites the whole trial history")
    ap.add_argument("--sparse-history", action="store_true", help="same as --history sparse (kept for old command lines)")
    ap.add_argument("--no-split-history", action="store_true",
                    help="comfe-rs plasticity workloads: keep the reference's 7-double history rows in the state instead of "
                         "ResidentState's [scalar, eps_p rows] layout (FCAMD_EVAL_SPLIT_HISTORY)")
    ap.add_argument("--sparse-tangent", action="store_true",
                    help="with --history sparse: also the sparse-tangent protocol of ResidentState (FCAMD_EVAL_SPARSE_TANGENT: "
                         "rows of points that stay elastic are not rewritten).  Not the reference contract -- the reported "
                         "bytes stay the interface's 464/568 B/pt, so `frac` is an equivalent, not a traffic, figure")
    ap.add_argument("--placement", choices=["auto", "vmm", "tune", "first"], default="auto",
                    help="where the arrays of the timed steps live (DESIGN.md 6): vmm = one working set with 2 MiB physical "
                         "handles interleaved over the arrays; tune = the fastest of --placement-tries hipMalloc candidates of "
                         "the tangent; auto (default, = ResidentState's default) = the faster of the two; first = what the "
                         "allocator gives")
    ap.add_argument("--placement-tries", type=int, default=6,
                    help="hipMalloc candidate allocations of the tangent array, timed with the real kernel before the run "
                         "(fenics_constitutive_amd.placement): in tune mode the fastest is kept, in vmm mode they are "
                         "timed for the record only (frac_first_allocation / frac_median_candidate)")
    ap.add_argument("--gather-direct", action="store_true",
                    help="N>1: also time the batched isend/irecv gather (RCCL point-to-point) next to RCCL's all-gather "
                         "and the C ABI's peer copies")
    ap.add_argument("--gather-points", type=int, default=0,
                    help="points per rank of the separately timed stress/tangent all-gather (N>1); 0 = the whole shard")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the all-gather leg")
    ap.add_argument("--gather-timeout", type=float, default=240.0,
                    help="N>1: seconds after which an unfinished all-gather leg is given up (the line is printed without it)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo only to exercise the multi-rank control flow on a box "
                         "with fewer GPUs than ranks (ranks then share GPUs; of the gather variants only the C ABI's "
                         "peer copies run)")
    ap.add_argument("--verbose", action="store_true", help="stage-by-stage progress lines on stderr (every rank)")
    ap.add_argument("--mode", choices=["device", "host"], default="device",
                    help="device (default): device-resident steps, one rank per GPU; host: the single-process multi-GPU HOST path "
                         "(fcamd_multi: one process, --gpus device contexts, every device evaluates its slice of the process's NumPy "
                         "arrays in place over its own PCIe link) -- PCIe-inclusive, its own line")
    ap.add_argument("--host-devices", default="", help="--mode host: explicit device ordinals, e.g. 0,0 (two contexts on one GPU)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = --points per GPU (default; a strong-scaling leg on the first points/N of every shard is reported "
                         "next to it under \"strong_scaling\"); strong = --points in total, cut into N contiguous shards")
    ap.add_argument("--no-host-path", action="store_true", help="N = 1 default run: skip the host_path block (PCIe-inclusive figures of the ndarray entries)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="N = 1: do not measure roofline.traffic in this run (two child passes under rocprofv3 --pmc, about 40 s); the "
                         "stored figure of profiles/traffic.json is reported instead (labelled as stored)")
    ap.add_argument("--wall-budget", type=float, default=420.0,
                    help="seconds of wall clock after which the optional legs (strong-scaling leg, gather variants, extra configurations, host_path) "
                         "are skipped so that the line is printed inside the driver's limit")
    args = ap.parse_args()
    if args.n is None:
        args.n = 10_000_000 if args.mode == "host" else 100_000_000
    if args.mode == "host":
        sys.exit(main_host(args))
    if args.frow is not None:
        sys.exit(main_frow(args))

    def stage(msg):
        if args.verbose:
            print(f"# [rank {os.environ.get('RANK', '0')} +{time.perf_counter() - t_prog:.1f}s] {msg}", file=sys.stderr, flush=True)

    t_prog = time.perf_counter()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL, fcamd_ipc_*)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            # convenience: started without the launcher -> start it as a child (nothing has touched the
            # GPU in this process yet) and pass its exit code on
            import subprocess

            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                   "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29517"),
                   os.path.abspath(__file__)] + [a if a != "--n" else "--points" for a in sys.argv[1:]]
            sys.exit(subprocess.run(cmd).returncode)
        sys.exit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
    guard = LineGuard() if rank == 0 else None  # before the first call that initialises the GPU
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    dev_index = local_rank % torch.cuda.device_count()  # identity on a node with one GPU per rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # launched by torch.distributed.run (RANK set): one rank per GPU over RCCL, also for world 1
    distributed = "RANK" in os.environ
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    stage("process group up")
    t_start = time.perf_counter()
    name = args.workload or HEADLINE
    history = "sparse" if args.sparse_history else args.history
    n = args.n
    if args.scaling == "strong" and world > 1:
        # --points in total: rank r owns the contiguous, tile-aligned shard fcamd_shard_bounds(points, world, r)
        from fenics_constitutive_amd import _capi

        lo, hi = _capi.shard_bounds(args.n, world, rank)
        n = hi - lo
        if n <= 0:
            sys.exit(f"--scaling strong: rank {rank} owns no point of {args.n}")

    def budget_left():
        return args.wall_budget - (time.perf_counter() - t_start)

    def agree(flag):
        if not (distributed and world > 1):
            return bool(flag)
        return all_agree(flag, dist, device if args.backend == "nccl" else "cpu")
    wl = Workload(name, n, seed=1234 + rank, device=device, dev_index=dev_index, history=history,
                  sparse_tangent=args.sparse_tangent, grid=args.grid,
                  split_history=not args.no_split_history)
    tries = args.placement_tries
    if world > 1 and tries > 1:
        # the candidates are alive together while they are timed: never more than fit next to the working set
        from fenics_constitutive_amd.placement import max_tries_for_memory

        tries = max_tries_for_memory(36 * n, tries, device, reserve_bytes=16 << 30)
    stage(f"workload built, placement {args.placement} with {tries} hipMalloc candidates")
    wl.place(args.placement, tries)
    stage(f"placed: {wl.vmm_info or wl.placement}")
    wl.warmup(args.warmup)
    wl.count_plastic()
    n_pl, n_its = wl.mean_plastic(args.steps)
    stage("warm-up and plastic counts done")

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev0[i].record()
        wl.launch(i)
        ev1[i].record()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    stage("timed steps done")
    wl.launch_log.append(["timed", args.steps])
    red_dev = device if args.backend == "nccl" else "cpu"
    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kernel_ms = sorted(ev0[i].elapsed_time(ev1[i]) for i in range(args.steps))
    kernel_avg_ms = sum(kernel_ms) / len(kernel_ms)
    per_rank_ms = None
    if distributed:  # every rank's own average kernel time: shows the balance behind the max-over-ranks wall time
        tk = torch.zeros(world, dtype=torch.float64, device=red_dev)
        tk[rank] = kernel_avg_ms
        dist.all_reduce(tk, op=dist.ReduceOp.SUM)
        per_rank_ms = [round(float(x), 4) for x in tk.tolist()]

    # N > 1, weak run: the strong-scaling figure next to it -- the same total as ONE GPU's shard (--points), i.e. every rank
    # evaluates the first points/world points (tile-aligned) of its arrays; same bracket (barrier + synchronize, max over ranks)
    strong = None
    if distributed and world > 1 and args.scaling == "weak" and agree(budget_left() > 60):
        m = max(64, (args.n // world // 64) * 64)
        for i in range(max(2, args.warmup)):
            wl.launch(i, m=m)
        sev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        dist.barrier()
        torch.cuda.synchronize()
        ts0 = time.perf_counter()
        for i, (a_, b_) in enumerate(sev):
            a_.record()
            wl.launch(i, m=m)
            b_.record()
        torch.cuda.synchronize()
        dist.barrier()
        s_el = time.perf_counter() - ts0
        tt = torch.tensor([s_el], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        s_el = float(tt.item())
        tk = torch.zeros(world, dtype=torch.float64, device=red_dev)
        tk[rank] = sum(a_.elapsed_time(b_) for a_, b_ in sev) / len(sev)
        dist.all_reduce(tk, op=dist.ReduceOp.SUM)
        wl.launch_log.append(["strong_scaling_leg", max(2, args.warmup) + args.steps])
        strong = {"value": round(m * world * args.steps / s_el / 1e6, 1), "unit": "Mpts/s", "points_total": m * world, "points_per_gpu": m,
                  "ms_per_step": round(s_el / args.steps * 1e3, 4), "per_rank_kernel_ms": [round(float(x), 4) for x in tk.tolist()],
                  "note": "strong scaling: the total of ONE GPU's weak shard cut over all ranks (every rank evaluates the first points/world points "
                          "of its arrays), same timing bracket as `value`; compare with the N = 1 line's value"}
        for i in (0, 1):  # the sparse protocol's masks and trial rows back in step with whole-array launches
            wl.launch(i)

    # next to the headline: the same step without the sparse protocol (every launch rewrites the whole trial
    # history, fcamd_evaluate_device_from) -- six extra launches after the timed region
    full_ms = None
    if wl.sparse:
        ms = wl.timed_events(6, phase="full_trial_history", full_history=True, sparse_tangent=False)
        full_ms = sum(ms[1:]) / (len(ms) - 1)

    # ... the sparse protocol on the reference's array layout (ResidentState(packed_history=False); the headline of rounds 1-3)
    unpacked_ms = None
    if wl.packed:
        wl.launch(0, unpacked=True), wl.launch(1, unpacked=True)
        wl.launch_log.append(["sparse_unpacked_history_warm", 2])
        ms = wl.timed_events(6, phase="sparse_unpacked_history", unpacked=True, sparse_tangent=False)
        unpacked_ms = sum(ms) / len(ms)

    # "achievable" next to "peak" (SURVEY 8d): a plain device copy over half of the tangent array
    # (read + write counted)
    copy_gbs = stream_copy_gbs = fill_gbs = read_gbs = None
    try:
        half = (wl.tangent.numel() // 2) & ~1
        cs, ce = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = None
        for _ in range(4):
            cs.record()
            wl.tangent[:half].copy_(wl.tangent[half : 2 * half])
            ce.record()
            ce.synchronize()
            ms_ = cs.elapsed_time(ce)
            best = ms_ if best is None else min(best, ms_)
        copy_gbs = 2 * 8 * half / (best * 1e-3) / 1e9
        # ... and the library's own copy kernel (fcamd_copy_device: the evaluate kernels' access pattern -- 16 B per lane,
        # non-temporal -- with nothing but the copy): what this box's memory gives a kernel of this kind
        from fenics_constitutive_amd.hostio import copy_device

        best = None
        for _ in range(5):
            cs.record()
            copy_device(wl.tangent[:half], wl.tangent[half : 2 * half])
            ce.record()
            ce.synchronize()
            ms_ = cs.elapsed_time(ce)
            best = ms_ if best is None else min(best, ms_)
        stream_copy_gbs = 2 * 8 * half / (best * 1e-3) / 1e9
        # ... and the two directions on their own: a pure write (fill) and a pure read (sum) of the same memory
        rates = {}
        for key, fn in (("fill", lambda: wl.tangent[:half].fill_(1.0)), ("read", lambda: wl.tangent[half : 2 * half].sum())):
            fn()
            best = None
            for _ in range(4):
                cs.record()
                fn()
                ce.record()
                ce.synchronize()
                ms_ = cs.elapsed_time(ce)
                best = ms_ if best is None else min(best, ms_)
            rates[key] = 8 * half / (best * 1e-3) / 1e9
        fill_gbs, read_gbs = rates["fill"], rates["read"]
    except Exception:  # the probe is informational only
        copy_gbs = None

    cpu_args = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # sample of the headline arrays, taken before they are released for the other configurations
        cpu_args = (wl.kind, wl.params, wl.grads[0][: 9 * 2_000_000].clone(), wl.stress_c[: 12_000_000].clone(),
                    None if wl.hist_c is None else {k: v[: {"eps_n": 6, "alpha": 1, "strain_visco": 6, "strain": 6, "history": 7}[k] * 2_000_000].clone()
                                                    for k, v in wl.reference_history().items()}, wl.del_t)
    headline = {"placement": wl.placement, "vmm_info": wl.vmm_info, "launch_log": list(wl.launch_log), "config_text": wl.config_text(), "kind": wl.kind,
                "b_el": wl.b_el, "b_pl": wl.b_pl, "alg": wl.alg_bytes(n_pl), "alg0": wl.alg_bytes(wl.n_pl_ab[0]),
                "fracs": placement_fracs(wl, wl.alg_bytes(wl.n_pl_ab[0])), "sparse": wl.sparse, "plasticity": wl.plasticity,
                "tkey": traffic_key(wl), "packed": wl.packed}

    out = None
    if rank == 0:  # the line is complete up to here; what follows only adds to it
        total_pts = (args.n if (args.scaling == "strong" and world > 1) else n * world) * args.steps
        value = total_pts / elapsed / 1e6
        alg_bytes = headline["alg"]
        achieved = alg_bytes / (kernel_avg_ms * 1e-3) / 1e9
        tkey = headline["tkey"]
        traffic = read_traffic(tkey, n)
        # what this box's memory allows for THIS kernel's measured mix of reads and writes: read bytes at the pure-read rate
        # plus written bytes at the pure-write rate, both measured above on the same array
        streaming_model = None
        split = read_traffic_split(tkey, n)
        if split and fill_gbs and read_gbs:
            model_ms = (split[0] / read_gbs + split[1] / fill_gbs) / 1e6
            streaming_model = {"ms": round(model_ms, 3), "kernel_over_model": round(kernel_avg_ms / model_ms, 3),
                               "note": "PMC read bytes / read_GBs + PMC written bytes / fill_GBs: the time it takes to stream the kernel's bytes at "
                                       "the rates torch's fill_ and sum reach on the same (placed) tangent array of this box; "
                                       "kernel_over_model < 1: the evaluate kernel moves its bytes faster than that"}
        out = {
            "metric": METRIC,
            "value": round(value, 1),
            "unit": "Mpts/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": headline["config_text"],
                       "baseline_config": BASELINE_CONFIG.get(name),
                       "points_per_gpu": n, "points_total": args.n if (args.scaling == "strong" and world > 1) else n * world,
                       "plastic_fraction": round(n_pl / n, 4),
                       "mean_newton_iters": round(n_its / max(n_pl, 1), 3) if headline["kind"] in ("von_mises_3d", "comfe_drucker_prager") else None,
                       "parallelism": f"shard{world}"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": None if traffic is None else f"profiles/traffic.json[{tkey}]@kernel_hash={str(library_hash(kernels_only=True))[:16]} "
                                                                        "(stored rocprofv3 --pmc measurement of these kernels, not measured in this run)",
                         "kernel_ms_avg": round(kernel_avg_ms, 4), "kernel_ms_min": round(kernel_ms[0], 4),
                         "kernel_ms_median": round(kernel_ms[len(kernel_ms) // 2], 4),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "traffic_GBs": None if traffic is None else round(traffic / (kernel_avg_ms * 1e-3) / 1e9, 1),
                         "device_copy_GBs": None if copy_gbs is None else round(copy_gbs, 1),
                         "stream_copy_GBs": None if stream_copy_gbs is None else round(stream_copy_gbs, 1),
                         "fill_GBs": None if fill_gbs is None else round(fill_gbs, 1),
                         "read_GBs": None if read_gbs is None else round(read_gbs, 1),
                         "streaming_model": streaming_model,
                         "copy_note": "device_copy_GBs: torch's device copy; fill_GBs / read_GBs: torch fill_ / sum over one half of the tangent array; stream_copy_GBs: fcamd_copy_device (16 B per lane, non-temporal, "
                                      "the evaluate kernels' access pattern with nothing but the copy) over half of the tangent array, read + "
                                      "write counted -- the achievable rate of this box next to the 8 TB/s peak (SURVEY 8d); traffic_GBs is the "
                                      "evaluate kernel's PMC-measured bytes over its time",
                         "bytes_per_point": {"elastic": headline["b_el"], "plastic": headline["b_pl"]},
                         "placement_note": "frac = the timed steps, arrays placed as `placement.mode` says (the product default); "
                                           "frac_first_allocation = hipMalloc candidate 0, what a caller gets who takes the allocator's "
                                           "arrays as they come, frac_median/worst/best_candidate = the other hipMalloc draws "
                                           "(min of 3 launches each, iterate 0)"},
        }
        out["roofline"].update(headline["fracs"])
        if full_ms is not None:
            out["full_trial_history"] = {"kernel_ms_avg": round(full_ms, 4),
                                         "frac": round(alg_bytes / (full_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                         "note": "same step, whole trial history rewritten by every launch (--history full)"}
        if unpacked_ms is not None:
            out["sparse_unpacked_history"] = {"kernel_ms_avg": round(unpacked_ms, 4),
                                              "frac": round(alg_bytes / (unpacked_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                              "note": "same step, same sparse protocol, plastic-strain arrays in the reference's layout "
                                                      "(--history sparse; isolated 48-byte rows instead of one run per tile)"}
        out["placement"] = dict(headline["vmm_info"] or {"mode": "hipmalloc_tuned" if headline["placement"] else "first"})
        if headline["placement"] is not None:
            out["placement"].update({"hipmalloc_tangent_" + k: v for k, v in headline["placement"].items()})
            out["placement"]["tries"] = tries
        if per_rank_ms is not None:
            out["per_rank_kernel_ms"] = per_rank_ms
        if strong is not None:
            out["strong_scaling"] = strong
        out["launch_log"] = headline["launch_log"]
        out["library"] = {"srchash": library_hash(), "kernel_hash": library_hash(kernels_only=True)}

    def emit():
        out["wall_s"] = round(time.perf_counter() - t_start, 1)
        guard.final(json.dumps(out))

    def checkpoint(leg):  # the line as it stands, in case the process does not survive `leg`
        if rank == 0:
            guard.provisional(dict(out, wall_s=round(time.perf_counter() - t_start, 1)), leg)
            if os.environ.get("BENCH_DIE_IN") == leg.split(":")[0]:  # knob: fault injection (tests/test_gpu_bench_cli.py)
                os.kill(os.getpid(), 9)

    # the exchange step of config 5, timed separately (never part of `value`)
    do_gather = distributed and world > 1 and not args.no_gather
    if do_gather and not agree(budget_left() > 90):
        do_gather = False
        if rank == 0:
            out["allgather"] = {"skipped": f"wall budget: {budget_left():.0f} s left of --wall-budget {args.wall_budget:.0f}"}
    if do_gather:
        checkpoint("allgather")
        # An exchange between 8 processes can hang in ways a single GPU cannot rehearse (a peer mapping that never
        # returns, ranks leaving a collective in different places): the measured line must survive that.  If the leg
        # has not finished after --gather-timeout seconds, rank 0 prints the line with an error entry and every rank
        # leaves the process.
        import threading

        def give_up():
            if rank == 0:
                out["allgather"] = {"error": f"the all-gather leg did not finish within {args.gather_timeout} s; step timing above is complete"}
                emit()
            os._exit(3)  # the line is out, but a hung exchange is a FAILED leg: the driver must see it

        watchdog = threading.Timer(min(args.gather_timeout, max(30.0, budget_left() - 30.0)), give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            stage("all-gather leg")
            wl.launch(0, sparse_tangent=False)  # a complete trial stress / tangent for the gather to move
            keep_s, keep_t = wl.stress_t, wl.tangent
            wl.grads = wl.hist_t = wl.hmask = None  # the gather needs the room, the step timing is done
            torch.cuda.empty_cache()
            gather = time_allgather(args, dist, torch, device, rank, world, n, keep_s, keep_t, agree=agree, budget_left=budget_left)
        except Exception as e:  # e.g. no room on every rank alike: the step timing above stands
            gather = {"error": f"{type(e).__name__}: {e}"[:400]}
        watchdog.cancel()
        if rank == 0:
            out["allgather"] = gather
    wl.free()

    # every other single-GPU configuration of BASELINE.json, same method, >= 5 event-timed launches each
    do_configs = args.configs == "all" or (args.configs == "auto" and args.workload is None and world == 1)
    if do_configs and rank == 0:
        configs = {}
        for k, cname in enumerate(EXTRA_CONFIGS):
            out["configs"] = configs
            checkpoint(f"configs: {cname}")
            if budget_left() < 45:
                configs[cname] = {"skipped": "wall budget"}
                continue
            try:
                # (under rocprofv3 released VMM memory stays alive: five more working sets would not fit -- hipMalloc candidates only)
                configs[cname] = run_config(cname, n, 4321 + k, device, dev_index, max(5, args.config_steps), 2,
                                            min(tries, 4), history=history,
                                            placement="tune" if (args.placement == "auto" and under_profiler()) else args.placement,
                                            cpu=not args.no_cpu_baseline)
            except Exception as e:  # one configuration failing must not lose the line
                configs[cname] = {"error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.empty_cache()
            # the same streaming model as for the headline, with the rates measured on this box
            split = read_traffic_split(cname, n)
            if split and fill_gbs and read_gbs and "kernel_ms_avg" in configs[cname]:
                model_ms = (split[0] / read_gbs + split[1] / fill_gbs) / 1e6
                configs[cname]["streaming_model"] = {"ms": round(model_ms, 3),
                                                     "kernel_over_model": round(configs[cname]["kernel_ms_avg"] / model_ms, 3)}
        out["configs"] = configs

    # the "next" rows of SURVEY 8(f) on the roofline: indexed evaluate (f2), fused wrapper and low-dimensional kernels (f3), the
    # resident state's sparse-tangent Newton iteration (f1) -- same method as the configurations, appended to `configs`
    frow_names = []
    if do_configs and rank == 0 and not args.no_frows:
        import bench_frows

        for fname in bench_frows.FROWS:
            checkpoint(f"frows: {fname}")
            if budget_left() < 60:
                out["configs"][fname] = {"skipped": "wall budget"}
                continue
            try:
                out["configs"][fname] = bench_frows.run_frow(fname, n, device, launches=max(5, args.config_steps), peak_gbs=HBM_PEAK_GBS)
                frow_names.append(fname)
            except Exception as e:  # one row failing must not lose the line
                out["configs"][fname] = {"error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.empty_cache()

    # N > 1: the single-process form of the host path on THIS node's GPUs (fcamd_multi) -- rank 0 alone drives all
    # of them over their own PCIe links while the other ranks wait on the CPU (a key in the process group's store, not a
    # collective: an RCCL barrier would keep their GPUs busy with a spinning kernel)
    if distributed and world > 1 and not args.no_host_path:
        import datetime

        store = None
        try:
            store = dist.distributed_c10d._get_default_store()
        except Exception:
            pass
        if rank == 0:
            import threading

            checkpoint("host_path_multi")

            def give_up_host():  # as for the gather leg: a leg that hangs on hardware nobody could rehearse must not cost the line
                out["host_path_multi"] = {"error": "the single-process multi-GPU host leg did not finish within its time limit; everything above is complete"}
                emit()
                os._exit(3)

            host_watchdog = threading.Timer(max(45.0, min(150.0, budget_left() - 20.0)), give_up_host)
            host_watchdog.daemon = True
            host_watchdog.start()
            try:
                if budget_left() > 75 and store is not None:
                    have = torch.cuda.device_count()
                    devs = [k % have for k in range(world)]  # fewer GPUs than ranks (gloo rehearsal): contexts share devices
                    torch.cuda.empty_cache()
                    per_dev = min(n, 2_500_000)
                    out["host_path_multi"] = host_path_figures(devices=devs, sizes=(min(1_000_000, per_dev * world), per_dev * world),
                                                               latency_sizes=(), reps=2, budget_s=40.0)
                else:
                    out["host_path_multi"] = {"skipped": "wall budget" if store is not None else "no process-group store to wait on"}
            except Exception as e:  # informational: must not lose the line
                out["host_path_multi"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            host_watchdog.cancel()
            if store is not None:
                store.set("fcamd_host_multi_done", "1")
        elif store is not None:
            try:
                store.wait(["fcamd_host_multi_done"], datetime.timedelta(seconds=max(60.0, budget_left())))
            except Exception:
                pass  # rank 0 is late or gone: the closing barrier below decides

    if rank == 0:
        if world == 1 and not args.no_host_path and args.workload is None and budget_left() > 40:
            # the number a dolfinx user sees: the ndarray entries over PCIe ("timed separately and labelled as such")
            checkpoint("host_path")
            try:
                torch.cuda.empty_cache()
                out["host_path"] = host_path_figures(devices=None, sizes=(1_000_000, min(n, 10_000_000)) if n > 1_000_000 else (n,))
            except Exception as e:  # informational: must not lose the line
                out["host_path"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 and not args.no_live_traffic and budget_left() > 100:
            # roofline.traffic measured in THIS run (the kernels' memory is free by now); the stored figure stays next to it
            checkpoint("live_traffic")
            extra = (["--no-split-history"] if args.no_split_history else []) \
                + (["--sparse-tangent"] if args.sparse_tangent else [])
            torch.cuda.empty_cache()
            lt = live_traffic(name, n, history, extra, min(120.0, budget_left() - 45.0))
            if lt is not None:
                rf = out["roofline"]
                rf["traffic_stored"] = rf["traffic"]
                rf["traffic"] = lt["hbm_bytes_per_launch"]
                rf["traffic_read_write"] = [lt["read_bytes"], lt["write_bytes"]]
                rf["traffic_GBs"] = round(lt["hbm_bytes_per_launch"] / (kernel_avg_ms * 1e-3) / 1e9, 1)
                rf["traffic_source"] = ("measured in this run: two child passes of this workload under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE "
                                        "(4 timed launches each; KiB x 1024, FETCH_SIZE x 2 on gfx950); traffic_stored = profiles/traffic.json")
                # ... and the other configurations of the line, the same way, while the budget lasts
                for cname, c in (out.get("configs") or {}).items():
                    if "kernel_ms_avg" not in c or budget_left() < 110:
                        continue
                    checkpoint(f"live_traffic: {cname}")
                    lc = live_traffic(cname, n, history, extra, min(60.0, budget_left() - 60.0), frow=cname in frow_names)
                    if lc is not None:
                        c["traffic_stored"], c["traffic"] = c.get("traffic"), lc["hbm_bytes_per_launch"]
                        c["traffic_source"] = "measured in this run (rocprofv3 --pmc child passes, as roofline.traffic)"
                        c["traffic_over_algorithmic"] = round(lc["hbm_bytes_per_launch"] / c["algorithmic_bytes_per_launch"], 4)
        if world == 1:
            checkpoint("cpu_baseline")
            out["cpu_baseline"] = cpu_baseline(*cpu_args) if cpu_args is not None else None
        emit()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
