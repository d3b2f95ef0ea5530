"""The "next" rows of SURVEY.md 8(f) on the roofline: workloads of bench.py's default line beside the BASELINE configurations.

Every row is one evaluate kernel launch per step on device-resident synthetic arrays, timed like the configurations
(HIP events on the launch stream around each launch, state reset between launches outside the bracket), with its
ALGORITHMIC bytes per point -- the traffic the call's interface mandates -- stated here and in DESIGN.md 3:

  row                                reference                                  bytes per point
  indexed_runs / indexed_scattered_  solver/maps.py:82-123 folded into the      LinearElasticity 456 + 4 (the int32 parent row)
  cells / indexed_permuted           kernel (f2), LinearElasticityModel
  wrapped_plane_strain_von_mises     models/utils.py:332-412 around VonMises3D  grad 32 + stress 32 R + 32 W + tangent 128 W + cached
                                     (f3, fused wrapper kernel)                 3-D stress 48 R + 48 W + alpha 8 R = 328 elastic;
                                                                                + eps_n 48 R + 48 W + alpha 8 W = 432 plastic
  lowdim_le_plane_strain             models/utils.py:52-87,153-186 (f3)         32 + 64 + 128 = 224
  lowdim_le_uniaxial_strain                                                     8 + 16 + 8 = 32
  lowdim_maxwell_plane_strain        spring_maxwell_model.py:40-88, 2-D         224 + 2 x (32 R + 32 W) = 352
  lowdim_maxwell_uniaxial_stress                                                32 + 2 x (8 R + 8 W) = 64
  resident_sparse_tangent            solver/_lawonsubmesh.py:72-95 +            ResidentState.evaluate, VonMises3D mixed (_zoned: plastic zones): a point that stays
                                     _history.py:64-88 (f1)                     elastic 72 + 96 + 8 = 176 (its tangent row is not rewritten),
                                                                                a plastic / formerly plastic one 568

Used by bench.py (benchlib/frows.py): `python bench.py --frow NAME` runs one row alone (the child of the PMC passes); the default run
appends all of them to `configs`."""

from __future__ import annotations

import os


VM_P = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}
LE_P = {"E": 42.0, "nu": 0.3}
SLS_P = {"E0": 42.0, "E1": 10.0, "tau": 10.0, "nu": 0.2}


class FRow:
    """one row: build the arrays, `launch()` = exactly one evaluate kernel; `reset()` restores the inputs it overwrites"""

    name = ""
    reference = ""

    def __init__(self, n, device, seed=77):
        import torch

        self.torch, self.n, self.device = torch, int(n), device
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed)
        self.f = dict(dtype=torch.float64, device=device)
        self.launch_log = []
        self.extra = {}

    def randn(self, k, scale):
        return self.torch.randn(k, generator=self.gen, **self.f) * scale

    def mixed_gradient(self, width):
        """per-point scale log-uniform in [1e-4, 1e-2] (the headline's mixture)"""
        t = self.torch
        g = t.randn(width * self.n, generator=self.gen, **self.f)
        g.view(self.n, width).mul_(t.pow(10.0, t.rand(self.n, generator=self.gen, **self.f) * 2.0 - 4.0)[:, None])
        return g

    def zoned_gradient(self, width, zone=4096, share=0.22):
        """contiguous zones of `zone` points, `share` of them at strain scale 1e-2 (plastic), the others at 1e-4: what a plastic zone of
        a mesh-ordered point array looks like (the workloads `*_zoned` of benchlib/workloads.py)"""
        t = self.torch
        g = t.randn(width * self.n, generator=self.gen, **self.f)
        pl = t.rand((self.n + zone - 1) // zone, generator=self.gen, **self.f) < share
        g.view(self.n, width).mul_(t.where(pl, 1e-2, 1e-4).to(t.float64).repeat_interleave(zone)[: self.n][:, None])
        return g

    def reset(self):
        pass

    def alg_bytes(self):
        raise NotImplementedError

    def timed(self, launches, phase="timed"):
        t = self.torch
        ev = [(t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)) for _ in range(launches)]
        for a, b in ev:
            self.reset()
            a.record()
            self.launch()
            b.record()
        t.cuda.synchronize()
        self.launch_log.append([phase, launches])
        return [a.elapsed_time(b) for a, b in ev]

    def free(self):
        for k in list(self.__dict__):
            if k not in ("torch", "n", "device", "launch_log", "extra", "name"):
                setattr(self, k, None)
        self.torch.cuda.empty_cache()


class IndexedRow(FRow):
    """LinearElasticityModel on a submesh of half of the parent's rows: the stress / tangent rows are addressed through the int32
    parent-row index inside the kernel (fcamd_eval_args.parent_rows) instead of map_to_sub + evaluate + 2 x map_to_parent"""

    reference = "solver/maps.py:82-123, solver/_lawonsubmesh.py:58-95"

    def __init__(self, n, device, order):
        import fenics_constitutive_amd as fc

        n_sub = max(64, (n // 2 // 8192) * 8192)
        super().__init__(n_sub, device)
        t = self.torch
        self.name = "indexed_" + order
        self.n_parent = 2 * n_sub
        self.law = fc.LinearElasticityModel(LE_P, fc.StressStrainConstraint.FULL)
        self.hist = None
        if os.environ.get("FROW_INDEXED_LAW") == "maxwell":  # experiments (tools/ab_knobs.py): the indexed SpringMaxwell kernel, 648 + 4 B/pt
            self.law = fc.SpringMaxwellModel(SLS_P, fc.StressStrainConstraint.FULL)
            self.hist = [{k: t.zeros(6 * n_sub, **self.f) for k in ("strain_visco", "strain")} for _ in range(2)]
        if order == "permuted":  # a map no mesh produces: every lane its own 48-byte / 288-byte row anywhere in the parent arrays
            rows = t.randperm(self.n_parent, device=device, generator=self.gen)[:n_sub]
        elif order == "scattered_cells":
            # what build_subspace_map (maps.py:125-177) yields for a material whose cells are sprinkled over the mesh: ascending cells,
            # the 4 quadrature points of a cell consecutive (V.dofmap.cell_dofs), every other cell at random belongs to the material
            cells = t.randperm(self.n_parent // 4, device=device, generator=self.gen)[: n_sub // 4].sort().values
            rows = (cells[:, None] * 4 + t.arange(4, device=device)[None, :]).reshape(-1)
        else:  # cells numbered in runs (blocks of 4096 points, every other block is this material's): whole tiles are consecutive rows
            blk = t.arange(n_sub // 4096, device=device) * 2
            rows = (blk[:, None] * 4096 + t.arange(4096, device=device)[None, :]).reshape(-1)
        self.rows = rows.to(t.int32).contiguous()
        self.grad = self.randn(9 * n_sub, 1e-3)
        self.stress_prev = self.randn(6 * self.n_parent, 1.0)
        self.stress = t.empty_like(self.stress_prev)
        self.tangent = t.empty(36 * self.n_parent, **self.f)
        how = {"permuted": "a random permutation of the parent rows", "runs": "runs of 4096 consecutive parent rows",
               "scattered_cells": "ascending cells of 4 consecutive rows, half of the cells at random"}[order]
        self.text = (f"{self.name}: LinearElasticityModel on {n_sub} points whose stress / tangent rows live in PARENT arrays of {self.n_parent} rows "
                     f"({how}), committed -> trial")

    def launch(self):
        self.law.evaluate_indexed(0.0, 1.0, self.grad, self.stress_prev, self.stress, self.tangent, self.rows,
                                  None if self.hist is None else self.hist[0], None if self.hist is None else self.hist[1])

    def alg_bytes(self):
        return 460 * self.n


class WrappedRow(FRow):
    """PlaneStrainFrom3D(VonMises3D), fused: one kernel replaces map -> evaluate -> map, no 3-D gradient / tangent arrays exist"""

    name = "wrapped_plane_strain_von_mises"
    reference = "models/utils.py:332-412 around mises_plasticity_isotropic_hardening.py:57-175"

    def __init__(self, n, device):
        import fenics_constitutive_amd as fc

        super().__init__(n, device)
        t = self.torch
        self.law = fc.VonMises3D(VM_P)
        self.w = fc.PlaneStrainFrom3D(self.law)
        self.grad = self.mixed_gradient(4)
        self.stress0 = t.zeros(4 * n, **self.f)
        self.stress = t.zeros_like(self.stress0)
        self.tangent = t.empty(16 * n, **self.f)
        self.h0 = {"eps_n": t.zeros(6 * n, **self.f), "alpha": t.rand(n, generator=self.gen, **self.f) * 0.02}
        self.h = {k: v.clone() for k, v in self.h0.items()}
        self.n_pl = 0
        self.text = f"{self.name}: VonMises3D under PlaneStrainFrom3D (fused wrapper kernel), {n} points, in place, per-point strain scale log-uniform in [1e-4, 1e-2]"

    def reset(self):
        self.stress.copy_(self.stress0)
        if self.w.stress_3d is not None:
            self.w.stress_3d.zero_()  # the wrapper's cached 3-D stress (utils.py:253-266)
        for k in self.h:
            self.h[k].copy_(self.h0[k])

    def launch(self):
        self.w.evaluate(0.0, 1.0, self.grad, self.stress, self.tangent, self.h)

    def count(self):
        self.reset()
        self.launch()
        self.n_pl = int(self.law.device_stats(self.device.index or 0).n_plastic)
        self.launch_log.append(["plastic_counts", 1])
        self.extra["plastic_fraction"] = round(self.n_pl / self.n, 4)

    def alg_bytes(self):
        return 328 * (self.n - self.n_pl) + 432 * self.n_pl


class LowDimRow(FRow):
    """the native low-dimensional kernels of linear elasticity and the SLS laws (1 / 4 doubles per point instead of 6 / 9)"""

    reference = "models/utils.py:52-87,153-186; linear_elasticity_model.py:26-45 / spring_maxwell_model.py:40-88"

    def __init__(self, n, device, kind, constraint):
        import fenics_constitutive_amd as fc

        super().__init__(n, device)
        t = self.torch
        c = getattr(fc.StressStrainConstraint, constraint)
        self.name = f"lowdim_{kind}_{constraint.lower()}"
        self.law = fc.LinearElasticityModel(LE_P, c) if kind == "le" else fc.SpringMaxwellModel(SLS_P, c)
        gd2, sd = c.geometric_dim**2, c.stress_strain_dim
        self.grad = self.randn(gd2 * n, 1e-3)
        self.stress0 = self.randn(sd * n, 1.0)
        self.stress = self.stress0.clone()
        self.tangent = t.empty(sd * sd * n, **self.f)
        self.h = None if kind == "le" else {k: self.randn(sd * n, 1e-3) for k in ("strain_visco", "strain")}
        self.bytes_per_point = 8 * (gd2 + 2 * sd + sd * sd + (0 if kind == "le" else 4 * sd))
        self.text = f"{self.name}: {type(self.law).__name__} {constraint}, {n} points, in place ({self.bytes_per_point} B/pt)"

    def launch(self):
        self.law.evaluate(0.0, 2.0, self.grad, self.stress, self.tangent, self.h)

    def alg_bytes(self):
        return self.bytes_per_point * self.n


class ResidentSparseTangentRow(FRow):
    """what a device assembler's Newton loop runs by default: ResidentState.evaluate on VonMises3D -- packed plastic-strain
    history, sparse trial history AND sparse tangent (only the tangent rows of plastic / formerly plastic points are rewritten)"""

    name = "resident_sparse_tangent"
    max_draws = 1
    reference = "solver/_lawonsubmesh.py:72-95, solver/_history.py:64-88 (the Newton-iteration protocol, state resident)"

    def __init__(self, n, device, zoned=False):
        import fenics_constitutive_amd as fc
        from fenics_constitutive_amd.resident import ResidentState

        super().__init__(n, device)
        t = self.torch
        if zoned:
            self.name = "resident_sparse_tangent_zoned"
        draw = (lambda w: self.zoned_gradient(w)) if zoned else (lambda w: self.mixed_gradient(w))
        self.law = fc.VonMises3D(VM_P)
        h0 = {"eps_n": t.zeros(6 * n, **self.f), "alpha": t.rand(n, generator=self.gen, **self.f) * 0.02}
        # the product default: the state's own placement step (hipMalloc candidates of the tangent + one VMM set, timed on the real launch)
        # runs inside the first evaluate -- one set of allocations is enough then (max_draws)
        self.state = ResidentState(self.law, n, device=device, history0=h0, placement=os.environ.get("FROW_PLACEMENT", "auto"))
        del h0
        warm = draw(9)
        self.state.evaluate(0.0, 1.0, warm)  # a committed state "from a previous step"
        self.state.update()
        del warm
        g0 = draw(9)
        self.grads = [g0, g0 * 1.03]  # two Newton iterates of one increment, evaluated alternately
        self.i = 0
        # launches of that first evaluate: its own + the placement step's (4 per hipMalloc candidate of the tangent and 1 evaluate on the
        # chosen one, 4 on the VMM working set) -- the dispatch slicing of the PMC passes and of the rocprof summaries counts on it
        pl = self.state.placement or {}
        cands = len(pl.get("candidate_ms", []))
        self.launch_log.append(["warm_increment_and_placement", 1 + (4 * cands + 1 if cands else 0) + (4 if "vmm_ms" in pl else 0)])
        self.extra["placement_mode"] = pl.get("mode", "torch")
        self.text = (f"{self.name}: ResidentState.evaluate, VonMises3D, {n} points{' in plastic zones of 4096 points (22 % of the zones)' if zoned else ''}, two alternating Newton iterates; packed plastic-strain history, "
                     "sparse trial history, sparse tangent, the state's own placement step (product default of a device assembler)")
        self.n_touched = 0

    def launch(self):
        self.state.evaluate(0.0, 1.0, self.grads[self.i & 1])
        self.i += 1

    def count(self):
        """points whose tangent / history rows one steady-state launch touches: plastic now or at the previous evaluate"""
        t = self.torch
        for _ in range(2):
            self.launch()
        old = self.state._mask.clone()
        self.launch()
        t.cuda.synchronize()
        shifts = t.arange(64, device=self.device, dtype=t.int64)
        self.n_touched = int((((old | self.state._mask)[:, None] >> shifts[None, :]) & 1).sum())
        self.launch_log.append(["touched_counts", 3])
        self.extra["touched_fraction"] = round(self.n_touched / self.n, 4)
        self.extra["plastic_fraction"] = round(int(self.state.check().n_plastic) / self.n, 4)

    def alg_bytes(self):
        return 176 * (self.n - self.n_touched) + 568 * self.n_touched

    def mem_floor(self, launches):
        """The row's SYNTHETIC TWIN, measured ("measure, don't compute, the bound"): the very launches of the
        timed phase issued as evaluate_twin_kernel (context option "twin_masks", csrc/fcamd_kernels.hip) -- the same loads and stores
        at the same addresses on the same buffers, every tile's plastic ballot read from a recording of the real step instead of
        computed, no constitutive arithmetic.  What the memory system alone takes for this request stream.  The state's trial arrays
        hold meaningless values afterwards: the last thing a row does."""
        t = self.torch
        ctx = self.law._handle(self.device.index or 0).ctx
        recorded = [None, None]
        for _ in range(2):  # the ballots of the two alternating iterates, as the real kernel leaves them in the state's mask
            p = self.i & 1
            self.launch()
            t.cuda.synchronize()
            recorded[p] = self.state._mask.clone()
        self.launch_log.append(["mem_floor_record", 2])
        ev = [(t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)) for _ in range(launches + 2)]
        try:
            for a, b in ev:
                ctx.set_option("twin_masks", recorded[self.i & 1].data_ptr())
                a.record()
                self.launch()
                b.record()
            t.cuda.synchronize()
        finally:
            ctx.set_option("twin_masks", 0)
        self.launch_log.append(["mem_floor_twin", launches + 2])
        same = bool(t.equal(self.state._mask, recorded[(self.i - 1) & 1]))  # the twin left the ballots the real step leaves
        ms = [a.elapsed_time(b) for a, b in ev][2:]
        self.extra["mem_floor_ms"] = round(sum(ms) / len(ms), 4)
        self.extra["mem_floor_ballots_reproduced"] = same


FROWS = {
    "indexed_runs": lambda n, d: IndexedRow(n, d, "runs"),
    "indexed_scattered_cells": lambda n, d: IndexedRow(n, d, "scattered_cells"),
    "indexed_permuted": lambda n, d: IndexedRow(n, d, "permuted"),
    "wrapped_plane_strain_von_mises": WrappedRow,
    "lowdim_le_plane_strain": lambda n, d: LowDimRow(n, d, "le", "PLANE_STRAIN"),
    "lowdim_le_uniaxial_strain": lambda n, d: LowDimRow(n, d, "le", "UNIAXIAL_STRAIN"),
    "lowdim_maxwell_plane_strain": lambda n, d: LowDimRow(n, d, "maxwell", "PLANE_STRAIN"),
    "lowdim_maxwell_uniaxial_stress": lambda n, d: LowDimRow(n, d, "maxwell", "UNIAXIAL_STRESS"),
    "resident_sparse_tangent": ResidentSparseTangentRow,
    "resident_sparse_tangent_zoned": lambda n, d: ResidentSparseTangentRow(n, d, zoned=True),
}
SURVEY_ROW = {"indexed_runs": "f2", "indexed_scattered_cells": "f2", "indexed_permuted": "f2", "wrapped_plane_strain_von_mises": "f3", "lowdim_le_plane_strain": "f3",
              "lowdim_le_uniaxial_strain": "f3", "lowdim_maxwell_plane_strain": "f3", "lowdim_maxwell_uniaxial_stress": "f3",
              "resident_sparse_tangent": "f1", "resident_sparse_tangent_zoned": "f1"}


def run_frow(name, n, device, launches=6, warm=2, peak_gbs=8000.0, draws=3, redraw=True, floor_launches=0):
    """one row, measured: warm launches, the row's own counts, `launches` event-timed launches -- on up to `draws` fresh sets
    of allocations (the kernel time follows where the driver puts the written arrays, DESIGN.md 6: `frac` is the fastest
    set, as the headline's is the fastest tangent candidate; `frac_first_allocation` what the first set gave)"""
    import torch

    rows, results = [], []
    try:
        k = -1
        while True:
            k += 1
            if k >= max(1, draws):
                # every set of allocations a slow one (recognisable from the byte rate alone)?  One more, once -- not for
                # the row whose rate is low by construction
                slow = max(r[2].alg_bytes() / (r[0] * 1e-3) / 1e9 / peak_gbs for r in results) < 0.77
                if not (slow and redraw and k == draws and draws > 1 and name != "indexed_permuted"):
                    break
            if rows and k >= getattr(rows[0], "max_draws", draws + 1):
                break  # a row that places its own arrays
            free_b = torch.cuda.mem_get_info(device)[0]
            if k > 0 and free_b < 1.3 * rows[0].extra.get("_bytes", 0):
                break  # the earlier sets stay alive (so that the allocator must find new memory): only while they fit
            before = torch.cuda.memory_allocated(device)
            row = FROWS[name](n, device)
            rows.append(row)
            for _ in range(warm):
                row.reset()
                row.launch()
            row.launch_log.append(["warmup", warm])
            if hasattr(row, "count"):
                row.count()
            ms = row.timed(launches)
            row.extra["_bytes"] = max(row.extra.get("_bytes", 0), torch.cuda.memory_allocated(device) - before)
            results.append((sum(ms) / len(ms), min(ms), row))
        avg, best_min, row = min(results, key=lambda r: r[0])
        if floor_launches and hasattr(row, "mem_floor"):
            row.mem_floor(floor_launches)  # (after the timed launches: the twin leaves the trial state meaningless)
            row.extra["kernel_over_mem_floor"] = round(avg / row.extra["mem_floor_ms"], 4)
        alg = row.alg_bytes()
        frac = lambda t: round(alg / (t * 1e-3) / 1e9 / peak_gbs, 4)  # noqa: E731
        log = [x for r in rows for x in r.launch_log] if len(rows) > 1 else row.launch_log
        out = {"survey_row": SURVEY_ROW[name], "reference": row.reference, "workload": row.text, "points": row.n, "launches": launches,
               "kernel_ms_avg": round(avg, 4), "kernel_ms_min": round(best_min, 4), "Mpts_s": round(row.n / (avg * 1e-3) / 1e6, 1),
               "algorithmic_bytes_per_launch": int(alg), "bytes_per_point": round(alg / row.n, 1),
               "achieved_GBs": round(alg / (avg * 1e-3) / 1e9, 1), "frac": frac(avg), "frac_first_allocation": frac(results[0][0]),
               "allocation_draws_ms": [round(r[0], 4) for r in results],
               "traffic": None, "traffic_over_algorithmic": None, "launch_log": log}
        out.update({k: v for k, v in row.extra.items() if not k.startswith("_")})
        return out
    finally:
        for r in rows:
            r.free()
