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


def _mirror_reference_import_paths() -> None:
    """The reference's import paths resolve under this package too: ``fenics_constitutive.models`` is a package whose
    namespace carries the interface, the helpers and the wrapper classes (``models/__init__.py:7-15``: ``from .interfaces
    import *``, ``from .utils import *``) and whose submodules users import from directly
    (``from fenics_constitutive.models.utils import PlaneStrainFrom3D``, ``... .models.rust_models import DruckerPrager3D``).
    Replacing ``fenics_constitutive.models`` by ``fenics_constitutive_amd.models`` in any such statement works."""
    import sys
    import types

    from . import interfaces, models, utils, wrappers

    helpers = {n: getattr(utils, n) for n in ("lame_parameters", "get_elastic_tangent", "get_identity", "strain_from_grad_u")}
    helpers.update(UniaxialStrainFrom3D=wrappers.UniaxialStrainFrom3D, PlaneStrainFrom3D=wrappers.PlaneStrainFrom3D)
    iface = {"IncrSmallStrainModel": interfaces.IncrSmallStrainModel, "StressStrainConstraint": interfaces.StressStrainConstraint}
    submodules = {
        "interfaces": iface,                                                                  # models/interfaces.py
        "utils": helpers,                                                                     # models/utils.py
        "linear_elasticity_model": {"LinearElasticityModel": models.LinearElasticityModel},
        "mises_plasticity_isotropic_hardening": {"VonMises3D": models.VonMises3D},
        "spring_maxwell_model": {"SpringMaxwellModel": models.SpringMaxwellModel},
        "spring_kelvin_model": {"SpringKelvinModel": models.SpringKelvinModel},
        "rust_models": {n: getattr(models, n) for n in ("LinearElasticity3D", "DruckerPrager3D", "DruckerPragerHyperbolic3D",
                                                        "MisesPlasticityLinearHardening3D")},
    }
    for name, value in {**iface, **helpers}.items():
        setattr(models, name, value)
    for sub, names in submodules.items():
        full = f"{models.__name__}.{sub}"
        mod = types.ModuleType(full, f"names of the reference's fenics_constitutive.models.{sub}, GPU-backed")
        mod.__dict__.update(names)
        mod.__all__ = sorted(names)
        sys.modules[full] = mod
        setattr(models, sub, mod)


_mirror_reference_import_paths()
