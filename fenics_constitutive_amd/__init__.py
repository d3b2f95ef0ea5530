"""Importable name of the package directory ``fenics-constitutive_amd/``.

The product lives in ``fenics-constitutive_amd/`` (the name the build contract asks for); a hyphen cannot appear in a
Python import, so this package has no modules of its own: its ``__path__`` is that directory, and every submodule
(``fenics_constitutive_amd.models``, ``._capi``, ...) is the file there -- an ordinary import, visible to coverage, type
checkers and ``importlib.reload``.
"""

import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "fenics-constitutive_amd")]

from ._api import *  # noqa: E402,F401,F403
from ._api import __all__, __version__  # noqa: E402,F401
