"""Loader of tests/golden/wrappers.npz (written by oracle/gen_golden.py --wrappers)."""
import os

import numpy as np
from golden_util import GOLDEN

VM_P = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}
SLS_P = {"E0": 42.0, "E1": 10.0, "tau": 10.0, "nu": 0.2}
LE_P = {"E": 42.0, "nu": 0.3}
PARAMS = {"le": LE_P, "vm": VM_P, "maxwell": SLS_P}


def load_sequences():
    """-> list of (wrapper kind, law name, [call dicts in order])"""
    z = np.load(os.path.join(GOLDEN, "wrappers.npz"))
    seqs = {}
    for i in range(int(z["n_calls"])):
        p = f"c{i}."
        key = (str(z[p + "wrapper"]), str(z[p + "law"]))
        c = {"grad": z[p + "grad"], "stress_in": z[p + "stress_in"], "stress_out": z[p + "stress_out"],
             "tangent_out": z[p + "tangent_out"], "hist_in": None, "hist_out": None}
        if p + "hist_keys" in z:
            keys = [str(k) for k in z[p + "hist_keys"]]
            c["hist_in"] = {k: z[p + "hist_in." + k] for k in keys}
            c["hist_out"] = {k: z[p + "hist_out." + k] for k in keys}
        seqs.setdefault(key, []).append(c)
    return [(k[0], k[1], v) for k, v in seqs.items()]


def load_constraint_calls(fname="constraints.npz"):
    """tests/golden/constraints.npz (or random_parameters_constraints.npz: every call with parameters of its own) -> list
    of call dicts (sequences are in file order)."""
    z = np.load(os.path.join(GOLDEN, fname))
    out = []
    for i in range(int(z["n_calls"])):
        p = f"c{i}."
        law = str(z[p + "law"])
        params = dict(zip([str(k) for k in z[p + "param_keys"]], [float(v) for v in z[p + "param_vals"]])) if p + "param_keys" in z \
            else {"le": LE_P, "maxwell": SLS_P, "kelvin": SLS_P}[law]
        c = {"constraint": str(z[p + "constraint"]), "law": law, "del_t": float(z[p + "del_t"]), "params": params,
             "grad": z[p + "grad"], "stress_in": z[p + "stress_in"], "stress_out": z[p + "stress_out"],
             "tangent_out": z[p + "tangent_out"], "hist_in": None, "hist_out": None}
        if p + "hist_in.strain" in z:
            c["hist_in"] = {k: z[p + "hist_in." + k] for k in ("strain_visco", "strain")}
            c["hist_out"] = {k: z[p + "hist_out." + k] for k in ("strain_visco", "strain")}
        out.append(c)
    return out


CPARAMS = {"le": LE_P, "maxwell": SLS_P, "kelvin": SLS_P}
