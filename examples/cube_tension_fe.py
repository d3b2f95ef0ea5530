#!/usr/bin/env python3
"""A partly yielding cube under tilted tension, solved with a global Newton iteration around the GPU law (examples/fe_mini.py:
hexahedra + SciPy, the role dolfinx plays for the reference).  Prints the Newton residuals of every load step: with the
consistent tangent VonMises3D returns they fall quadratically.

    python examples/cube_tension_fe.py [cells per edge] [resident|ndarray]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fe_mini as FE  # noqa: E402

import fenics_constitutive_amd as fc  # noqa: E402
from fenics_constitutive_amd.resident import ResidentState  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mode = sys.argv[2] if len(sys.argv) > 2 else "resident"
mesh = FE.Cube(m, m, m)
law = fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0})
state = FE.ResidentProtocolState(ResidentState(law, mesh.n_points), mesh.n_points) if mode == "resident" else FE.CopyProtocolState(law, mesh.n_points)
reactions, norms, u = FE.tension_test(mesh, state, steps=8)
for k, (r, h) in enumerate(zip(reactions, norms), 1):
    print(f"load step {k}: reaction {r:10.3f}   Newton residuals " + "  ".join(f"{x:.2e}" for x in h))
orders = FE.convergence_orders(norms)
print(f"{mesh.n_points} quadrature points, {mesh.n_dofs} dofs, {sum(len(h) - 1 for h in norms)} Newton iterations, "
      f"observed convergence orders {min(orders):.2f} .. {max(orders):.2f}")
assert max(orders) >= 1.8
