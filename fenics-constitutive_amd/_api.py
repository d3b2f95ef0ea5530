"""MI355X-native quadrature-point constitutive-update engine.

Drop-in for the hot path of BAMresearch/fenics-constitutive,
``IncrSmallStrainModel.evaluate(t, del_t, grad_del_u, stress, tangent, history)``:
hand-written HIP kernels for gfx950 behind a C ABI (``include/fcamd.h``) behind the
reference's own model interface.  Import as ``fenics_constitutive_amd``.

(The public names live here so that the package's own ``__init__`` and the importable alias
``fenics_constitutive_amd/__init__.py`` -- a hyphen cannot appear in an import -- are both two lines.)
"""

from .interfaces import IncrSmallStrainModel, StressStrainConstraint  # noqa: F401
from .models import (  # noqa: F401
    DruckerPrager3D,
    DruckerPragerHyperbolic3D,
    LinearElasticity3D,
    LinearElasticityModel,
    MisesPlasticityLinearHardening3D,
    SpringKelvinModel,
    SpringMaxwellModel,
    VonMises3D,
)
from .wrappers import PlaneStrainFrom3D, UniaxialStrainFrom3D  # noqa: F401
from .utils import get_elastic_tangent, get_identity, lame_parameters, strain_from_grad_u  # noqa: F401

__version__ = "0.1.0"

__all__ = [
    "IncrSmallStrainModel",
    "StressStrainConstraint",
    "LinearElasticityModel",
    "VonMises3D",
    "SpringMaxwellModel",
    "SpringKelvinModel",
    "LinearElasticity3D",
    "MisesPlasticityLinearHardening3D",
    "DruckerPrager3D",
    "DruckerPragerHyperbolic3D",
    "UniaxialStrainFrom3D",
    "PlaneStrainFrom3D",
    "lame_parameters",
    "get_elastic_tangent",
    "get_identity",
    "strain_from_grad_u",
]
