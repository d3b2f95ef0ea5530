"""The miniature incremental small-strain solver (examples/fe_mini.py: hexahedra, global Newton iteration, the reference's
state protocol around evaluate) driven by the CPU ORACLE laws: pins the scaffolding itself before the GPU laws go through it
(tests/test_gpu_fe_mini.py) -- an elastic cube reproduces the closed form, a partly yielding one converges quadratically
with the consistent tangent the law returns (what the reference's NewtonSolver relies on, solver/_solver.py:130-147)."""

import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
import fe_mini as FE  # noqa: E402
from oracle import numpy_oracle as O  # noqa: E402

VM_P = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}


def test_elastic_cube_matches_uniaxial_strain_closed_form():
    """uniform pull (tilt = 0) with the lateral faces free and nu = 0: sigma_zz = E eps_zz, reaction = E * d * area"""
    mesh = FE.Cube(3, 2, 4)
    law = FE.OracleLaw(O.linear_elasticity, {"E": 42.0, "nu": 0.0}, None)
    reactions, norms, u = FE.tension_test(mesh, FE.CopyProtocolState(law, mesh.n_points), steps=2, top_displacement=0.01, tilt=0.0)
    assert np.allclose(reactions, 42.0 * 0.01 * np.array([0.5, 1.0]), rtol=1e-12)
    assert all(len(h) <= 2 for h in norms)  # linear problem: one solve


def test_yielding_cube_converges_quadratically():
    mesh = FE.Cube(4, 3, 4)
    law = FE.OracleLaw(O.von_mises_3d, VM_P, {"eps_n": 6, "alpha": 1})
    state = FE.CopyProtocolState(law, mesh.n_points)
    reactions, norms, u = FE.tension_test(mesh, state, steps=6)
    assert (state.hist_c["alpha"] > 0).mean() > 0.3 and (state.hist_c["alpha"] == 0).any()  # a part of the cube has yielded
    assert np.all(np.diff(reactions) > 0) and reactions[-1] - reactions[-2] < 0.8 * (reactions[1] - reactions[0])  # it softens
    worst = max(len(h) for h in norms)
    assert worst <= 7, norms
    orders = FE.convergence_orders(norms)
    assert orders and min(orders) >= 1.3 and max(orders) >= 1.8, norms  # superlinear from the first iterations on, quadratic near the solution
