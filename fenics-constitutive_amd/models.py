"""GPU-backed constitutive laws with the reference's class names, constructor arguments,
``evaluate`` signature, ``history_dim`` and error behaviour.

Each class is a thin host-side mirror: it keeps the material parameters, and ``evaluate``
goes through the C ABI (``include/fcamd.h``) into the HIP kernels -- NumPy arrays via the
staged host path, torch ROCm tensors zero-copy (see ``device.py``).

Reference classes mirrored:
  LinearElasticityModel              models/linear_elasticity_model.py:9-56
  VonMises3D                         models/mises_plasticity_isotropic_hardening.py:9-186
  SpringMaxwellModel                 models/spring_maxwell_model.py:8-102
  SpringKelvinModel                  models/spring_kelvin_model.py:9-102
  LinearElasticity3D                 models/rust_models.py:84-94  (comfe-rs linear_elasticity.rs)
  MisesPlasticityLinearHardening3D   models/rust_models.py:145-161 (comfe-rs mises_plasticity.rs)
"""

from __future__ import annotations

import numpy as np

from . import _capi
from .device import DeviceLaw
from .interfaces import StressStrainConstraint
from .utils import get_elastic_tangent, get_identity, lame_parameters

__all__ = [
    "DruckerPrager3D",
    "DruckerPragerHyperbolic3D",
    "LinearElasticityModel",
    "VonMises3D",
    "SpringMaxwellModel",
    "SpringKelvinModel",
    "LinearElasticity3D",
    "MisesPlasticityLinearHardening3D",
]


class LinearElasticityModel(DeviceLaw):
    """Linear elasticity: ``sigma += d_eps . D``, ``tangent = D`` at every point.

    Args:
        parameters: ``{"E": Young's modulus, "nu": Poisson ratio}``.
        constraint: stress/strain constraint (device kernels: FULL).
    """

    _model_id = _capi.LINEAR_ELASTICITY

    def __init__(self, parameters: dict[str, float], constraint: StressStrainConstraint):
        E, nu = parameters["E"], parameters["nu"]
        super().__init__([E, nu], constraint)
        self.D = get_elastic_tangent(E, nu, constraint)

    @property
    def history_dim(self) -> None:
        return None


class VonMises3D(DeviceLaw):
    r"""Von Mises plasticity with nonlinear (saturation) isotropic hardening; per-point scalar
    Newton iteration on the plastic multiplier, algorithmically consistent tangent.

    Args:
        param: ``p_ka`` bulk modulus, ``p_mu`` shear modulus, ``p_y0`` initial yield stress,
            ``p_y00`` saturated yield stress, ``p_w`` saturation parameter.
    """

    _model_id = _capi.VON_MISES_3D

    def __init__(self, param: dict[str, float]):
        self.p_ka = param["p_ka"]
        self.p_mu = param["p_mu"]
        self.p_y0 = param["p_y0"]
        self.p_y00 = param["p_y00"]
        self.p_w = param["p_w"]
        super().__init__([self.p_ka, self.p_mu, self.p_y0, self.p_y00, self.p_w], StressStrainConstraint.FULL)
        # host mirrors of the reference's public attributes (:33-49)
        self.xioi = np.zeros((6, 6), dtype=np.int64)
        self.xioi[:3, :3] = 1
        self.I2 = get_identity(6, StressStrainConstraint.FULL)
        self.I4 = np.eye(6, dtype=np.float64)
        self.xpp = self.I4 - (1 / 3) * self.xioi

    @property
    def history_dim(self) -> dict[str, int]:
        return {"eps_n": 6, "alpha": 1}


class _SpringBase(DeviceLaw):
    def __init__(self, parameters: dict[str, float], constraint: StressStrainConstraint):
        self.E0 = parameters["E0"]
        self.E1 = parameters["E1"]
        self.tau = parameters["tau"]
        self.nu = 0.0 if constraint.name == "UNIAXIAL_STRESS" else parameters["nu"]
        super().__init__([self.E0, self.E1, self.tau, self.nu], constraint)

    @property
    def history_dim(self) -> dict[str, int]:
        return {"strain_visco": self.stress_strain_dim, "strain": self.stress_strain_dim}

    def evaluate(self, t, del_t, grad_del_u, stress, tangent, history, check: bool = False) -> None:
        if history is None:
            raise ValueError("history must not be None")
        assert del_t > 0, "Time step must be defined and positive."
        super().evaluate(t, del_t, grad_del_u, stress, tangent, history, check=check)


class SpringMaxwellModel(_SpringBase):
    """Standard linear solid, spring parallel to a Maxwell arm, backward Euler.

    Args:
        parameters: ``E0`` elastic modulus, ``E1`` viscous-arm modulus, ``tau`` relaxation
            time, ``nu`` Poisson ratio.
        constraint: stress/strain constraint (device kernels: FULL).
    """

    _model_id = _capi.SPRING_MAXWELL

    def __init__(self, parameters, constraint):
        super().__init__(parameters, constraint)
        self.D_0 = get_elastic_tangent(self.E0, self.nu, constraint)
        self.D_1 = get_elastic_tangent(self.E1, self.nu, constraint)
        self.mu1, _ = lame_parameters(self.E1, self.nu)


class SpringKelvinModel(_SpringBase):
    """Standard linear solid, spring in series with a Kelvin body, backward Euler.

    Args: as :class:`SpringMaxwellModel`.
    """

    _model_id = _capi.SPRING_KELVIN

    def __init__(self, parameters, constraint):
        super().__init__(parameters, constraint)
        self.D_0 = get_elastic_tangent(self.E0, self.nu, constraint)
        self.I2 = get_identity(self.stress_strain_dim, constraint)
        self.mu0, self.lam0 = lame_parameters(self.E0, self.nu)
        self.mu1, _ = lame_parameters(self.E1, self.nu)


def _scalar(parameters: dict, key: str) -> float:
    """Rust-style parameter dicts carry one-element arrays (tests/models/test_plasticity.py:26-31)."""
    v = np.asarray(parameters[key], dtype=np.float64).reshape(-1)
    if v.size != 1:
        raise ValueError(f"parameter '{key}' must have exactly one entry")
    return float(v[0])


class LinearElasticity3D(DeviceLaw):
    """comfe-rs ``LinearElasticity3D``: ``sigma += C . d_eps`` with ``C = 2 mu P_dev + 3 kappa P_vol``.

    Args:
        parameters: ``{"mu": array([..]), "kappa": array([..])}``.
    """

    _model_id = _capi.COMFE_LINEAR_ELASTICITY

    def __init__(self, parameters: dict[str, np.ndarray]):
        super().__init__([_scalar(parameters, "mu"), _scalar(parameters, "kappa")], StressStrainConstraint.FULL)

    @property
    def history_dim(self) -> None:
        return None


class MisesPlasticityLinearHardening3D(DeviceLaw):
    """comfe-rs ``MisesPlasticity3D``: von Mises yield function, linear isotropic hardening,
    closed-form radial return.  History ``{"history": 7}`` = ``[alpha, plastic_strain(6)]``.

    Args:
        parameters: ``{"mu", "kappa", "y_0", "h"}`` as one-element arrays.
    """

    _model_id = _capi.COMFE_MISES_PLASTICITY

    def __init__(self, parameters: dict[str, np.ndarray]):
        super().__init__(
            [_scalar(parameters, k) for k in ("mu", "kappa", "y_0", "h")], StressStrainConstraint.FULL
        )

    @property
    def history_dim(self) -> dict[str, int]:
        return {"history": 7}


class DruckerPrager3D(DeviceLaw):
    """comfe-rs classic Drucker-Prager plasticity (``f = sqrt(J2) + b I1 - a``) solved by the general
    return mapping (reference: models/rust_models.py:97-118, comfe-rs/src/plasticity/general.rs,
    drucker_prager_classic.rs).  History ``{"history": 7}`` = ``[alpha, plastic_strain(6)]``.

    Args:
        parameters: ``{"mu", "kappa", "a", "b", "b_flow"}`` as one-element arrays.
    """

    _model_id = _capi.COMFE_DRUCKER_PRAGER

    def __init__(self, parameters: dict[str, np.ndarray]):
        super().__init__([_scalar(parameters, k) for k in ("mu", "kappa", "a", "b", "b_flow")], StressStrainConstraint.FULL)

    @property
    def history_dim(self) -> dict[str, int]:
        return {"history": 7}


class DruckerPragerHyperbolic3D(DeviceLaw):
    """comfe-rs hyperbolically smoothed Drucker-Prager plasticity (``f = sqrt(J2 + d^2) + b I1 - a``),
    reference: models/rust_models.py:121-142, drucker_prager_hyperbolic.rs.

    Args:
        parameters: ``{"mu", "kappa", "a", "b", "d", "b_flow"}`` as one-element arrays.
    """

    _model_id = _capi.COMFE_DRUCKER_PRAGER_HYPERBOLIC

    def __init__(self, parameters: dict[str, np.ndarray]):
        super().__init__([_scalar(parameters, k) for k in ("mu", "kappa", "a", "b", "d", "b_flow")], StressStrainConstraint.FULL)

    @property
    def history_dim(self) -> dict[str, int]:
        return {"history": 7}
