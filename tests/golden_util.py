"""Loader for the golden fixtures written by oracle/gen_golden.py."""

from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@dataclass
class Call:
    name: str
    params: dict
    del_t: float
    grad: np.ndarray
    stress_in: np.ndarray
    stress_out: np.ndarray
    tangent_out: np.ndarray
    hist_in: dict | None = None
    hist_out: dict | None = None
    extra: dict = field(default_factory=dict)

    @property
    def n(self):
        return self.grad.size // 9

    def fresh(self):
        """(stress, tangent, history) ready to be overwritten in place."""
        h = None if self.hist_in is None else {k: v.copy() for k, v in self.hist_in.items()}
        return self.stress_in.copy(), np.full(36 * self.n, np.nan), h


def load_calls(fname: str) -> list[Call]:
    z = np.load(os.path.join(GOLDEN, fname))
    out = []
    for i, name in enumerate(z["calls"]):
        p = f"c{i}."
        params = dict(zip([str(k) for k in z[p + "param_keys"]], [float(v) for v in z[p + "param_vals"]]))
        hin = hout = None
        if p + "hist_keys" in z:
            keys = [str(k) for k in z[p + "hist_keys"]]
            hin = {k: z[p + "hist_in." + k] for k in keys}
            hout = {k: z[p + "hist_out." + k] for k in keys}
        out.append(
            Call(str(name), params, float(z[p + "del_t"]), z[p + "grad"], z[p + "stress_in"],
                 z[p + "stress_out"], z[p + "tangent_out"], hin, hout)
        )
    return out


def rel_err(a, b):
    """max |a-b| / max(|b|) -- the relative measure used for all parity gates."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = np.max(np.abs(b)) if b.size else 0.0
    if scale == 0.0:
        return float(np.max(np.abs(a - b))) if a.size else 0.0
    return float(np.max(np.abs(a - b)) / scale)
