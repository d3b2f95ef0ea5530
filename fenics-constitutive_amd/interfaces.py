"""The model interface of the hot path: ``StressStrainConstraint`` and the ABC
``IncrSmallStrainModel`` -- our statement of
``src/fenics_constitutive/models/interfaces.py:14-143`` of the reference.

Drop-in mode: when the reference package ``fenics_constitutive`` is importable (a dolfinx
installation), its own two classes are re-exported instead, so that the device-backed
models below are *real* subclasses of the reference ABC and carry the reference enum --
``IncrSmallStrainProblem`` (``solver/_solver.py:67-73``) then accepts them unchanged.
"""

from __future__ import annotations

from abc import ABC, abstractmethod
from enum import Enum

import numpy as np

__all__ = ["IncrSmallStrainModel", "StressStrainConstraint", "REFERENCE_INTERFACES"]

try:  # pragma: no cover - exercised only where dolfinx + the reference are installed
    from fenics_constitutive.models.interfaces import (  # type: ignore
        IncrSmallStrainModel,
        StressStrainConstraint,
    )

    REFERENCE_INTERFACES = True
except Exception:  # the normal case on a bare GPU box
    REFERENCE_INTERFACES = False

    class StressStrainConstraint(Enum):
        """Constraint on stresses/strains; values and dimensions as in the reference
        (interfaces.py:14-73)."""

        UNIAXIAL_STRAIN = 1
        UNIAXIAL_STRESS = 2
        PLANE_STRAIN = 3
        PLANE_STRESS = 4
        FULL = 5

        @property
        def stress_strain_dim(self) -> int:
            """Length of the Mandel stress/strain vector."""
            return {1: 1, 2: 1, 3: 4, 4: 4, 5: 6}[self.value]

        @property
        def geometric_dim(self) -> int:
            """Spatial dimension of the displacement gradient."""
            return {1: 1, 2: 1, 3: 2, 4: 2, 5: 3}[self.value]

    class IncrSmallStrainModel(ABC):
        """Interface for incremental small-strain models (interfaces.py:76-143)."""

        @abstractmethod
        def evaluate(
            self,
            t: float,
            del_t: float,
            grad_del_u: np.ndarray,
            stress: np.ndarray,
            tangent: np.ndarray,
            history: dict[str, np.ndarray] | None,
        ) -> None:
            """Evaluate the law at all quadrature points and overwrite ``stress``,
            ``tangent`` and ``history`` in place.

            ``t`` is the time at the start of the increment, ``del_t`` the increment,
            ``grad_del_u`` the flat row-major gradient of ``u_{n+1} - u_n``; stress and
            tangent are in Mandel notation."""

        @property
        @abstractmethod
        def constraint(self) -> StressStrainConstraint:
            """The stress/strain constraint the model is implemented for."""

        @property
        def stress_strain_dim(self) -> int:
            return self.constraint.stress_strain_dim

        @property
        def geometric_dim(self) -> int:
            return self.constraint.geometric_dim

        @property
        @abstractmethod
        def history_dim(self) -> dict[str, int | tuple[int, int]] | None:
            """Per-point dimension of every history field, or ``None``."""
