#!/usr/bin/env python3
"""Six materials on one mesh, uniaxial tension, DEVICE assembler: the whole Newton loop stays on the GPU.

The reference's ``form()`` calls one ``LawOnSubMesh.evaluate`` per material (solver/_solver.py:143-144).  Here the six laws -- three
VonMises3D variants, linear elasticity, the two SLS laws, dealt over the cells at random -- live in one ``ResidentProblemState`` and a
Newton iteration is ``state.evaluate(gradients)``: ONE ``fcamd_evaluate_batch``, i.e. one launch of the batch kernel for all six laws
(their arguments are kept in a table on the device and only re-uploaded when a pointer changes), commit = pointer swap.  The loop is
the one ``IncrSmallStrainProblem`` drives, on the lateral strains of every quadrature point until its lateral stresses vanish.

    python examples/many_materials_device.py [n_cells]        prints the time per Newton iteration with the batch kernel on and off
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FCAMD_SMALL_CALL_WARNING", "0")
import fenics_constitutive_amd as fc  # noqa: E402
from fenics_constitutive_amd.problem import ResidentProblemState, rows_of_cells  # noqa: E402

n_cells, q = (int(sys.argv[1]) if len(sys.argv) > 1 else 6000), 4
n = n_cells * q
FULL = fc.StressStrainConstraint.FULL
VM = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}


def make_laws():
    return [fc.VonMises3D(VM), fc.VonMises3D({**VM, "p_y0": 900.0}), fc.VonMises3D({**VM, "p_y00": 3000.0, "p_w": 120.0}),
            fc.LinearElasticityModel({"E": 70000.0, "nu": 0.3}, FULL),
            fc.SpringMaxwellModel({"E0": 60000.0, "E1": 20000.0, "tau": 10.0, "nu": 0.25}, FULL),
            fc.SpringKelvinModel({"E0": 60000.0, "E1": 20000.0, "tau": 10.0, "nu": 0.25}, FULL)]


def run(batch):
    rng = np.random.default_rng(0)
    owner = rng.integers(0, 6, size=n_cells)
    rows = [rows_of_cells(np.flatnonzero(owner == k), q) for k in range(6)]
    state = ResidentProblemState(list(zip(make_laws(), rows)), n, del_t=1.0, batch_launches=batch, placement="torch")
    dev = state.device
    rows_d = [torch.from_numpy(r.astype(np.int64)).to(dev) for r in rows]
    grads = [torch.zeros(r.size, 9, dtype=torch.float64, device=dev) for r in rows]  # the SAME tensors every iteration: the kept call is replayed
    max_strain = torch.linspace(0.6, 1.0, n, dtype=torch.float64, device=dev) * 0.02
    prev = torch.zeros(n, dtype=torch.float64, device=dev)
    iterations, t_eval = 0, []
    for step in np.linspace(0, 1, 21)[1:]:
        cur = step * max_strain
        d_eps = torch.zeros(n, 3, dtype=torch.float64, device=dev)
        d_eps[:, 0] = cur - prev
        while True:
            for g, r in zip(grads, rows_d):
                g[:, 0], g[:, 4], g[:, 8] = d_eps[r, 0], d_eps[r, 1], d_eps[r, 2]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            state.evaluate([g.view(-1) for g in grads])
            torch.cuda.synchronize()
            t_eval.append(time.perf_counter() - t0)
            iterations += 1
            res = state.stress_1.view(n, 6)[:, 1:3]
            if float(res.abs().max()) < 1e-8:
                break
            J = state.tangent.view(n, 6, 6)[:, 1:3, 1:3]
            d_eps[:, 1:3] -= torch.linalg.solve(J, res.unsqueeze(-1)).squeeze(-1)
        state.update()
        prev = cur
    s = state.stress_0.view(n, 6)
    # (median: the very first launches of a process load the kernels' code objects, tens of milliseconds once)
    return iterations, sorted(t_eval)[len(t_eval) // 2] * 1e6, s[:, 0].clone(), [int(ls.law.last_stats.n_plastic) if ls.counters is not None else 0 for ls in state._laws]


it_b, us_b, sxx_b, plastic = run(True)
it_s, us_s, sxx_s, _ = run(False)
assert it_b == it_s and torch.equal(sxx_b, sxx_s), "the batch kernel must give the states of the launches made one by one, bit for bit"
print(f"{n} points, 6 materials, 20 increments, {it_b} Newton iterations; sigma_xx max {float(sxx_b.max()):.2f}; plastic points per law {plastic}")
print(f"constitutive part of one Newton iteration (median): {us_b:.1f} us as ONE batch launch, {us_s:.1f} us law by law ({us_s / us_b:.2f}x)")
