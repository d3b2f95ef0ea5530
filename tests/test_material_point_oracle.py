"""The reference's FE-level model assertions (uniaxial stress/cyclic plasticity, relaxation, creep,
Kelvin-Maxwell equivalence, plane strain vs 3-D, elastic constraints) on the CPU oracle through the
material-point harness -- pins the oracle beyond the golden vectors with the analytic limits the
reference itself tests."""

import numpy as np
import pytest

import material_point_cases as cases
from material_point import HostState, MaterialPoints, OracleLaw
from oracle import numpy_oracle as O

HIST = {
    "linear_elasticity": None,
    "von_mises_3d": {"eps_n": 6, "alpha": 1},
    "comfe_mises_plasticity": {"history": 7},
}
SHORT = {"linear_elasticity": "le", "spring_maxwell": "maxwell", "spring_kelvin": "kelvin"}


def build(kind, params, constraint, n):
    if kind in SHORT:
        sd = O.DIMS[constraint][1]
        hd = None if kind == "linear_elasticity" else {"strain_visco": sd, "strain": sd}
        law = OracleLaw(O.MODELS_C[SHORT[kind]], params, hd, constraint, pass_constraint=True)
    elif kind.startswith("comfe_drucker_prager"):
        law = OracleLaw(O.comfe_drucker_prager, params, {"history": 7}, hyperbolic=kind.endswith("hyperbolic"))
    else:
        assert constraint == "FULL"
        law = OracleLaw(O.MODELS[kind], params, HIST[kind])
    return MaterialPoints(HostState(law, n), constraint, tol=1e-11)


@pytest.mark.parametrize("kind", ["von_mises_3d", "comfe_mises_plasticity"])
def test_uniaxial_stress_3d(kind):
    load, _ = cases.uniaxial_stress_3d(build, kind)
    if kind == "von_mises_3d":  # the curve of the reference's own VonMises3D class
        cases.assert_matches_reference_curve("uniaxial_stress_3d.load", load, 1e-9)


def test_uniaxial_cyclic_strain_3d():
    load, _ = cases.uniaxial_cyclic_strain_3d(build)
    cases.assert_matches_reference_curve("uniaxial_cyclic_strain_3d.load", load, 1e-9)


@pytest.mark.parametrize("kind", ["spring_kelvin", "spring_maxwell"])
@pytest.mark.parametrize("constraint", ["UNIAXIAL_STRESS", "PLANE_STRESS", "FULL"])
def test_relaxation(kind, constraint):
    cases.assert_matches_reference_curve(f"relaxation.{kind}.{constraint}", cases.relaxation(build, kind, constraint), 1e-11)


@pytest.mark.parametrize("kind", ["spring_kelvin", "spring_maxwell"])
@pytest.mark.parametrize("constraint", ["PLANE_STRESS", "FULL"])
def test_creep(kind, constraint):
    cases.assert_matches_reference_curve(f"creep.{kind}.{constraint}", cases.creep(build, kind, constraint), 1e-11)


def test_kelvin_vs_maxwell():
    cases.kelvin_vs_maxwell(build)


@pytest.mark.parametrize("kind", ["spring_kelvin", "spring_maxwell"])
def test_plane_strain_vs_3d(kind):
    cases.plane_strain_vs_3d(build, kind)


def test_elasticity_constraints():
    cases.elasticity_constraints(build)


@pytest.mark.parametrize("hyperbolic", [False, True])
def test_drucker_prager_uniaxial(hyperbolic):
    cases.drucker_prager_uniaxial(build, hyperbolic)
