"""Host-side helpers of the model interface (reference: ``models/utils.py:18-208``).

``lame_parameters`` / ``get_elastic_tangent`` / ``get_identity`` are tiny per-model
constants (host, once per model).  ``strain_from_grad_u`` is array-sized and runs on the
GPU (``fcamd_strain_from_grad_u_device``, with the component maps of ``fcamd_convert_device``
for the 1-D / 2-D constraints).
"""

from __future__ import annotations

import numpy as np

from .interfaces import StressStrainConstraint

__all__ = ["lame_parameters", "get_elastic_tangent", "get_identity", "strain_from_grad_u"]


def lame_parameters(E: float, nu: float) -> tuple[float, float]:
    """(mu, lam) from Young's modulus and Poisson's ratio (utils.py:18-22)."""
    mu = E / (2.0 * (1.0 + nu))
    lam = E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu))
    return mu, lam


def get_elastic_tangent(E: float, nu: float, constraint: StressStrainConstraint) -> np.ndarray:
    """Linear-elastic tangent in Mandel notation for every constraint (utils.py:25-93)."""
    mu, lam = lame_parameters(E, nu)
    name = constraint.name
    if name in ("FULL", "PLANE_STRAIN"):
        dim = 6 if name == "FULL" else 4
        D = np.zeros((dim, dim))
        D[:3, :3] = lam
        for i in range(3):
            D[i, i] = 2.0 * mu + lam
        for i in range(3, dim):
            D[i, i] = 2.0 * mu
        return D
    if name == "PLANE_STRESS":
        return E / (1 - nu**2.0) * np.array(
            [[1.0, nu, 0.0, 0.0], [nu, 1.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, (1.0 - nu)]]
        )
    if name == "UNIAXIAL_STRAIN":
        return np.array([[E * (1.0 - nu) / ((1.0 + nu) * (1.0 - 2.0 * nu))]])
    if name == "UNIAXIAL_STRESS":
        return np.array([[E]])
    raise NotImplementedError("Constraint not implemented")


def get_identity(stress_strain_dim: int, constraint: StressStrainConstraint) -> np.ndarray:
    """Second-order identity in Mandel notation (utils.py:96-129)."""
    ones = {"FULL": 3, "PLANE_STRAIN": 3, "PLANE_STRESS": 2, "UNIAXIAL_STRAIN": 1, "UNIAXIAL_STRESS": 1}
    if constraint.name not in ones:
        raise NotImplementedError("Constraint not implemented")
    I2 = np.zeros(stress_strain_dim, dtype=np.float64)
    I2[: ones[constraint.name]] = 1.0
    return I2


def strain_from_grad_u(grad_u, constraint: StressStrainConstraint):
    """Mandel strain from the displacement gradient (utils.py:132-208).

    Runs on the GPU for every constraint: NumPy in -> NumPy out (staged), torch CUDA tensor in ->
    CUDA tensor out (zero copy).
    """
    name = constraint.name
    if name == "FULL":
        from .device import strain_from_grad_u_full

        return strain_from_grad_u_full(grad_u)
    if name in ("UNIAXIAL_STRAIN", "UNIAXIAL_STRESS", "PLANE_STRAIN", "PLANE_STRESS"):
        from .device import strain_from_grad_u_lowdim

        return strain_from_grad_u_lowdim(grad_u, constraint)
    raise NotImplementedError("Constraint not supported.")
