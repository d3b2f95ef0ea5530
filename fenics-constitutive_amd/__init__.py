"""MI355X-native quadrature-point constitutive-update engine (see ``_api.py``); import as ``fenics_constitutive_amd``."""

from ._api import *  # noqa: F401,F403
from ._api import __all__, __version__  # noqa: F401
