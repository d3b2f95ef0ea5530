"""ctypes wrapper of the C oracle (oracle/oracle.c).  TEST INFRASTRUCTURE / CPU baseline only:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product package."""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "liboracle.so")
_lib = None


def _src_hash() -> str:
    import hashlib

    h = hashlib.sha256()
    for f in ("oracle.c", "Makefile"):
        with open(os.path.join(HERE, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force: bool = False) -> str:
    """(Re)build liboracle.so when its sources changed (content hash, not mtimes: snapshot copies
    do not keep mtime order)."""
    stamp = LIB + ".srchash"
    fresh = os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == _src_hash()
    if force or not fresh:
        subprocess.run(["make", "-C", HERE, "-B", "-s"], check=True, capture_output=True)
        with open(stamp, "w") as fh:
            fh.write(_src_hash())
    return LIB


def build_flags() -> str:
    """Compiler and flags of the build (the CFLAGS line of oracle/Makefile), for bench.py's cpu_baseline.sample."""
    import re

    mk = open(os.path.join(HERE, "Makefile")).read()
    cc = re.search(r"^CC \?= (.*)$", mk, re.M).group(1).strip()
    return f"{os.environ.get('CC', cc)} {os.environ.get('CFLAGS', re.search(r'^CFLAGS [?]= (.*)$', mk, re.M).group(1).strip())}"


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
        _lib.oracle_von_mises_3d.restype = C.c_longlong
        _lib.oracle_comfe_mises.restype = C.c_longlong
        _lib.oracle_comfe_drucker_prager.restype = C.c_longlong
    return _lib


def set_num_threads(n: int) -> None:
    """Threads of the point loops (default 1 = the reference's serial loop)."""
    lib().oracle_set_num_threads(C.c_int(int(n)))


def max_threads() -> int:
    return int(lib().oracle_max_threads())


def _p(a):
    if a is None:
        return C.c_void_p(0)
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return C.c_void_p(a.ctypes.data)


def _d(x):
    return C.c_double(float(x))


def strain_from_grad_u(grad, rust=False):
    n = grad.size // 9
    out = np.empty(6 * n)
    lib().oracle_strain_from_grad_u(C.c_longlong(n), _p(grad), _p(out), C.c_int(int(rust)))
    return out


def linear_elasticity(p, t, del_t, grad, stress, tangent, history=None):
    lib().oracle_linear_elasticity(_d(p["E"]), _d(p["nu"]), C.c_longlong(grad.size // 9), _p(grad), _p(stress), _p(tangent))


def von_mises_3d(p, t, del_t, grad, stress, tangent, history):
    npl, nit = C.c_longlong(0), C.c_longlong(0)
    bad = lib().oracle_von_mises_3d(_d(p["p_ka"]), _d(p["p_mu"]), _d(p["p_y0"]), _d(p["p_y00"]), _d(p["p_w"]),
                                    C.c_longlong(grad.size // 9), _p(grad), _p(stress), _p(tangent),
                                    _p(history["eps_n"]), _p(history["alpha"]), C.byref(npl), C.byref(nit))
    if bad:
        raise RuntimeError("Newton-Raphson method did not converge for plastic multiplier.")
    return npl.value, nit.value


def spring_maxwell(p, t, del_t, grad, stress, tangent, history):
    lib().oracle_spring_maxwell(_d(p["E0"]), _d(p["E1"]), _d(p["tau"]), _d(p["nu"]), _d(del_t), C.c_longlong(grad.size // 9),
                                _p(grad), _p(stress), _p(tangent), _p(history["strain_visco"]), _p(history["strain"]))


def spring_kelvin(p, t, del_t, grad, stress, tangent, history):
    lib().oracle_spring_kelvin(_d(p["E0"]), _d(p["E1"]), _d(p["tau"]), _d(p["nu"]), _d(del_t), C.c_longlong(grad.size // 9),
                               _p(grad), _p(stress), _p(tangent), _p(history["strain_visco"]), _p(history["strain"]))


def comfe_linear_elasticity(p, t, del_t, grad, stress, tangent, history=None):
    lib().oracle_comfe_linear_elasticity(_d(p["mu"]), _d(p["kappa"]), C.c_longlong(grad.size // 9), _p(grad), _p(stress), _p(tangent))


def comfe_mises_plasticity(p, t, del_t, grad, stress, tangent, history):
    return lib().oracle_comfe_mises(_d(p["mu"]), _d(p["kappa"]), _d(p["y_0"]), _d(p["h"]), C.c_longlong(grad.size // 9),
                                    _p(grad), _p(stress), _p(tangent), _p(history["history"]))


def comfe_drucker_prager(p, t, del_t, grad, stress, tangent, history, hyperbolic=False):
    """Returns (n_plastic, total Newton iterations); raises like numpy_oracle.comfe_drucker_prager."""
    nit, flags = C.c_longlong(0), C.c_int(0)
    npl = lib().oracle_comfe_drucker_prager(
        C.c_int(1 if hyperbolic else 0), _d(p["mu"]), _d(p["kappa"]), _d(p["a"]), _d(p["b"]), _d(p.get("d", 0.0)),
        _d(p["b_flow"]), C.c_longlong(grad.size // 9), _p(grad), _p(stress), _p(tangent), _p(history["history"]),
        C.byref(nit), C.byref(flags))
    if flags.value & 1:
        raise AssertionError("non-differentiable tip of Drucker-Prager surface reached")
    if flags.value & 6:
        raise RuntimeError("Plasticity3D: Newton-Raphson did not converge.")
    return int(npl), int(nit.value)


MODELS = {
    "linear_elasticity": linear_elasticity,
    "von_mises_3d": von_mises_3d,
    "spring_maxwell": spring_maxwell,
    "spring_kelvin": spring_kelvin,
    "comfe_linear_elasticity": comfe_linear_elasticity,
    "comfe_mises_plasticity": comfe_mises_plasticity,
    "comfe_drucker_prager": comfe_drucker_prager,
}
