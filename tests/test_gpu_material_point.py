"""The reference's FE-level model assertions on the HIP path (through the model classes and the C
ABI): the same material-point scenarios as tests/test_material_point_oracle.py, once with NumPy
arrays through the in-place host path (the unchanged solver's call protocol) and once with the
device-resident committed/trial state (ResidentState, pointer-swap commit; device tensors, and
the host assembler's pipelined ``evaluate_into``)."""

import numpy as np
import pytest

import material_point_cases as cases
from material_point import HostState, MaterialPoints, MultiResidentAdapter, ResidentAdapter

pytestmark = pytest.mark.gpu


def make_law(kind, params, constraint):
    import fenics_constitutive_amd as fc

    c = fc.StressStrainConstraint[constraint]
    if kind == "linear_elasticity":
        return fc.LinearElasticityModel(params, c)
    if kind == "spring_maxwell":
        return fc.SpringMaxwellModel(params, c)
    if kind == "spring_kelvin":
        return fc.SpringKelvinModel(params, c)
    if kind == "von_mises_3d":
        return fc.VonMises3D(params)
    if kind == "comfe_mises_plasticity":
        return fc.MisesPlasticityLinearHardening3D({k: np.array([v]) for k, v in params.items()})
    if kind == "comfe_drucker_prager":
        return fc.DruckerPrager3D({k: np.array([v]) for k, v in params.items()})
    if kind == "comfe_drucker_prager_hyperbolic":
        return fc.DruckerPragerHyperbolic3D({k: np.array([v]) for k, v in params.items()})
    raise KeyError(kind)


def builder(mode):
    def build(kind, params, constraint, n):
        law = make_law(kind, params, constraint)
        if mode == "multi_host":  # the in-place ndarray evaluate spread over three device contexts, one tile per context at least
            law.use_devices([0, 0, 0])
            law._multi().set_option("min_points", 32)
        if mode == "multi_resident":
            state = MultiResidentAdapter(law, n)
        else:
            state = HostState(law, n) if mode.endswith("host") and mode != "resident_host" else ResidentAdapter(law, n, host_assembler=(mode == "resident_host"))
        return MaterialPoints(state, constraint, tol=1e-11)

    return build


MODES = ["host", "resident", "resident_host", "multi_host", "multi_resident"]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("kind", ["von_mises_3d", "comfe_mises_plasticity"])
def test_uniaxial_stress_3d(kind, mode):
    load, _ = cases.uniaxial_stress_3d(builder(mode), kind, n=70)  # one full tile + a ragged tail
    # and the whole load path agrees with the oracle's
    from test_material_point_oracle import build as oracle_build

    ref, _ = cases.uniaxial_stress_3d(oracle_build, kind, n=70)
    assert np.max(np.abs(load - ref)) < 1e-6 * np.max(np.abs(ref))
    if kind == "von_mises_3d":  # and, at the fixture's 8 amplitudes, with the reference's own class
        load8, _ = cases.uniaxial_stress_3d(builder(mode), kind, n=8)
        cases.assert_matches_reference_curve("uniaxial_stress_3d.load", load8, 1e-6)


@pytest.mark.parametrize("mode", MODES)
def test_uniaxial_cyclic_strain_3d(mode):
    cases.uniaxial_cyclic_strain_3d(builder(mode), n=6)
    load, _ = cases.uniaxial_cyclic_strain_3d(builder(mode), n=4)
    cases.assert_matches_reference_curve("uniaxial_cyclic_strain_3d.load", load, 1e-6)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("kind", ["spring_kelvin", "spring_maxwell"])
@pytest.mark.parametrize("constraint", ["UNIAXIAL_STRESS", "PLANE_STRESS", "FULL"])
def test_relaxation(kind, constraint, mode):
    cases.assert_matches_reference_curve(f"relaxation.{kind}.{constraint}", cases.relaxation(builder(mode), kind, constraint), 1e-10)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("kind", ["spring_kelvin", "spring_maxwell"])
@pytest.mark.parametrize("constraint", ["PLANE_STRESS", "FULL"])
def test_creep(kind, constraint, mode):
    cases.assert_matches_reference_curve(f"creep.{kind}.{constraint}", cases.creep(builder(mode), kind, constraint), 1e-10)


@pytest.mark.parametrize("mode", MODES)
def test_kelvin_vs_maxwell(mode):
    cases.kelvin_vs_maxwell(builder(mode))


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("kind", ["spring_kelvin", "spring_maxwell"])
def test_plane_strain_vs_3d(kind, mode):
    cases.plane_strain_vs_3d(builder(mode), kind)


@pytest.mark.parametrize("mode", MODES)
def test_elasticity_constraints(mode):
    cases.elasticity_constraints(builder(mode))


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("hyperbolic", [False, True])
def test_drucker_prager_uniaxial(hyperbolic, mode):
    from test_material_point_oracle import build as oracle_build

    load = cases.drucker_prager_uniaxial(builder(mode), hyperbolic, n=70)
    ref = cases.drucker_prager_uniaxial(oracle_build, hyperbolic, n=70)
    assert np.max(np.abs(load - ref)) < 1e-8 * np.max(np.abs(ref))
