"""Importable alias of the package directory ``fenics-constitutive_amd/``.

The product lives in ``fenics-constitutive_amd/`` (the name the build contract asks for);
a hyphen cannot appear in a Python import, so this stub points ``__path__`` at that
directory and executes its ``__init__``.  ``import fenics_constitutive_amd`` therefore
gives exactly the package in ``fenics-constitutive_amd/``.
"""

import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "fenics-constitutive_amd")
__path__ = [_real]
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__, "r") as _f:
    exec(compile(_f.read(), __file__, "exec"))
del _f, _os, _real
