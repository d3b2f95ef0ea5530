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


def mises_limit_cases():
    """tests/golden/mises_linear_hardening_limit.npz (oracle/gen_golden.py: main_mises_limit) -> for every call the inputs of
    comfe-rs MisesPlasticity3D and what it must return, derived from the outputs of the IMPORTED Python VonMises3D in its
    linear-hardening limit (w = 1e-6, h = (y00 - y0) w):

    * stress and alpha: the same numbers (both laws are radial return with sigma_y = y0 + h alpha in this limit);
    * plastic strain: the Rust text adds del_gamma * s_tr / s_tr_eq (mises_plasticity.rs:110-112, a flow direction of length
      sqrt(2/3)), the Python one gamma * s_tr / |s_tr| (mises_plasticity_isotropic_hardening.py:161) -> a factor sqrt(2/3);
    * tangent: kappa I x I + 2 mu theta P_dev is common; the last term is -2 mu theta_bar n n^T with a unit n in Python
      (:172-176) and +2 mu theta_bar (2/3) n n^T in Rust (mises_plasticity.rs:117-121) -> the Rust tangent is the Python one
      plus (10/3) mu theta_bar n n^T at plastic points.

    ``tol``: the limit's own error (truncation w alpha / 2 and the cancellation in 1 - exp(-w alpha), both ~1e-8 of the
    hardening term h alpha) relative to the stress, with a margin."""
    out = []
    for c in load_calls("mises_linear_hardening_limit.npz"):
        p, n = c.params, c.n
        mu, h = p["p_mu"], (p["p_y00"] - p["p_y0"]) * p["p_w"]
        rs = {"mu": mu, "kappa": p["p_ka"], "y_0": p["p_y0"], "h": h}
        f = np.sqrt(2.0 / 3.0)
        a_in, a_out = c.hist_in["alpha"], c.hist_out["alpha"]
        e_in, e_out = c.hist_in["eps_n"].reshape(n, 6), c.hist_out["eps_n"].reshape(n, 6)
        h_in = np.concatenate([a_in[:, None], f * e_in], axis=1).reshape(-1)
        gamma = (a_out - a_in) / f
        pl = gamma > 0
        xn = (e_out - e_in)[pl] / gamma[pl, None]
        s_out = c.stress_out.reshape(n, 6)
        dev = s_out.copy()
        dev[:, :3] -= s_out[:, :3].mean(axis=1, keepdims=True)
        norm_tr = np.linalg.norm(dev[pl], axis=1) + 2.0 * mu * gamma[pl]   # s_out_dev = theta s_tr = s_tr - 2 mu gamma n
        theta = 1.0 - 2.0 * mu * gamma[pl] / norm_tr
        theta_bar = 1.0 / (1.0 + h / (3.0 * mu)) - (1.0 - theta)
        T = c.tangent_out.reshape(n, 6, 6).copy()
        T[pl] += (10.0 / 3.0) * mu * theta_bar[:, None, None] * xn[:, :, None] * xn[:, None, :]
        expected = {"stress": c.stress_out, "alpha": a_out, "eps_p": f * e_out, "tangent": T.reshape(-1)}
        tol = 2e-9 if h / p["p_y0"] < 1.0 else 2e-7
        out.append({"name": c.name, "params": rs, "grad": c.grad, "stress_in": c.stress_in, "history_in": h_in, "expected": expected,
                    "plastic": pl, "tol": tol})
    return out


def check_mises_limit(case, stress, tangent, history):
    n = case["grad"].size // 9
    e, tol = case["expected"], case["tol"]
    h = np.asarray(history).reshape(n, 7)
    assert rel_err(stress, e["stress"]) <= tol, (case["name"], "stress", rel_err(stress, e["stress"]))
    assert rel_err(h[:, 0], e["alpha"]) <= tol, (case["name"], "alpha", rel_err(h[:, 0], e["alpha"]))
    assert rel_err(h[:, 1:], e["eps_p"].reshape(n, 6)) <= 4 * tol, (case["name"], "eps_p", rel_err(h[:, 1:], e["eps_p"].reshape(n, 6)))
    assert rel_err(tangent, e["tangent"]) <= 4 * tol, (case["name"], "tangent", rel_err(tangent, e["tangent"]))
    assert 0.5 < case["plastic"].mean() < 1.0  # the branch under test, with elastic points next to it


def dp_j2_cases():
    """tests/golden/von_mises_perfect_plasticity.npz (oracle/gen_golden.py: main_mises_limit) -> inputs of comfe-rs
    DruckerPrager3D / DruckerPragerHyperbolic3D with b = b_flow = 0 and what the general return mapping
    (plasticity/general.rs:105-266) must return for them, from the outputs of the IMPORTED Python VonMises3D with
    y00 = y0 (perfect plasticity):

    * b = b_flow = 0 makes both surfaces J2 cylinders, sqrt(J2) = a (drucker_prager_classic.rs:88) resp. sqrt(a^2 - d^2)
      (drucker_prager_hyperbolic.rs:87), with the flow direction along s: the closest-point projection the 8 x 8 Newton
      iteration solves IS the radial return -> the same stress; the same plastic-strain increment (general.rs:243 with
      drucker_prager_classic.rs:108: del_eps - C^-1 (sigma_1 - sigma_0)); the same consistent tangent (general.rs:244-253; it
      is symmetric here, so the transposed store does not show);
    * the hardening variable follows the Rust text alone: res_kappa = alpha_1 - alpha_0 - k with k = sqrt(2/3) |g| and no
      del_lambda (general.rs:208) -> + 1/sqrt(3) (classic), + sqrt((a^2 - d^2) / 3) / a (hyperbolic) at every plastic point."""
    out = []
    for hyper in (False, True):
        for c in load_calls("von_mises_perfect_plasticity.npz"):
            p, n = c.params, c.n
            y0 = p["p_y0"]
            d = 0.03 * y0
            a = float(np.sqrt(y0**2 / 3.0 + d**2)) if hyper else y0 / float(np.sqrt(3.0))
            dp = {"mu": p["p_mu"], "kappa": p["p_ka"], "a": a, "b": 0.0, "b_flow": 0.0}
            if hyper:
                dp = {"mu": p["p_mu"], "kappa": p["p_ka"], "a": a, "b": 0.0, "d": d, "b_flow": 0.0}
            pl = c.hist_out["alpha"] > c.hist_in["alpha"]
            k_incr = float(np.sqrt((a * a - d * d) / 3.0) / a) if hyper else 1.0 / float(np.sqrt(3.0))
            h_in = np.concatenate([np.zeros((n, 1)), c.hist_in["eps_n"].reshape(n, 6)], axis=1).reshape(-1)
            out.append({"name": ("hyperbolic-" if hyper else "classic-") + c.name, "hyperbolic": hyper, "params": dp, "grad": c.grad,
                        "stress_in": c.stress_in, "history_in": h_in, "plastic": pl,
                        "expected": {"stress": c.stress_out, "eps_p": c.hist_out["eps_n"], "tangent": c.tangent_out, "kappa": np.where(pl, k_incr, 0.0)}})
    return out


def check_dp_j2(case, stress, tangent, history, tol=1e-11):
    """tolerance: the Rust iteration stops at 1e-8 (general.rs:174-175, 219-225) but converges quadratically -- its last
    iterate is at rounding level here; 1e-11 leaves a margin over the 1e-14 the restatements measure"""
    n = case["grad"].size // 9
    e = case["expected"]
    h = np.asarray(history).reshape(n, 7)
    assert rel_err(stress, e["stress"]) <= tol, (case["name"], "stress", rel_err(stress, e["stress"]))
    assert rel_err(h[:, 1:], e["eps_p"].reshape(n, 6)) <= tol, (case["name"], "eps_p")
    assert rel_err(tangent, e["tangent"]) <= tol, (case["name"], "tangent", rel_err(tangent, e["tangent"]))
    # the hardening variable follows the state of the LAST BUT ONE iterate (its row of the Newton system is linearised): with the
    # hyperbolic surface, whose |g| depends on J2, it carries the size of the last step, bounded by the iteration's 1e-8
    assert rel_err(h[:, 0], e["kappa"]) <= (1e-7 if case["hyperbolic"] else tol), (case["name"], "kappa", rel_err(h[:, 0], e["kappa"]))
    assert 0.1 < case["plastic"].mean() < 1.0  # the branch under test, with elastic points next to it


def dp_pressure_cases():
    """tests/golden/drucker_prager_deviatoric_flow.npz (oracle/gen_golden.py: main_mises_limit) -> inputs of the comfe-rs
    Drucker-Prager laws with b != 0, b_flow = 0 and what the general return mapping must return, from POINT-BY-POINT calls
    of the imported Python VonMises3D:

    * b_flow = 0 makes the flow direction purely deviatoric (drucker_prager_classic.rs:96-103): the pressure stays at its trial
      value and every point returns radially onto sqrt(J2) = R with R = a - b I1_trial (classic, :88) resp.
      sqrt((a - b I1_trial)^2 - d^2) (hyperbolic.rs:87) -- the radial return of VonMises3D with y00 = y0 = sqrt(3) R of that
      point: same stress, same plastic strain;
    * the tangent d sigma / d eps of the Rust law additionally carries the derivative of R: with sigma_dev = sqrt(2) R n,
      d R / d eps = -3 kappa b (A / R) I2 (A = a - b I1_trial; A / R = 1 for the classic surface) -> the Python tangent
      - 3 sqrt(2) kappa b (A / R) n (x) I2 at plastic points, rows = n, columns = I2: a NON-SYMMETRIC term, which also pins the
      orientation of the stored 6 x 6 block (general.rs:244-253 transposes before the column-major store);
    * hardening variable: + sqrt(2/3) |g| with |g| = R / (sqrt(2) A) (general.rs:208: no del_lambda)."""
    z = np.load(os.path.join(GOLDEN, "drucker_prager_deviatoric_flow.npz"))
    i2 = np.array([1.0, 1.0, 1.0, 0.0, 0.0, 0.0])
    out = []
    for i in range(int(z["n_calls"])):
        q = f"c{i}."
        p = dict(zip([str(k) for k in z[q + "param_keys"]], [float(v) for v in z[q + "param_vals"]]))
        hyper = p["d"] != 0.0
        if not hyper:
            del p["d"]
        n = z[q + "grad"].size // 9
        e_in, e_out = z[q + "eps_p_in"].reshape(n, 6), z[q + "eps_p_out"].reshape(n, 6)
        pl = np.abs(e_out - e_in).max(axis=1) > 0.0
        s_out = z[q + "stress_out"].reshape(n, 6)
        dev = s_out.copy()
        dev[:, :3] -= s_out[:, :3].mean(axis=1, keepdims=True)
        nvec = dev[pl] / np.linalg.norm(dev[pl], axis=1, keepdims=True)
        R, A = z[q + "radius"][pl], z[q + "a_minus_b_i1"][pl]
        T = z[q + "tangent_py"].reshape(n, 6, 6).copy()
        T[pl] -= (3.0 * np.sqrt(2.0) * p["kappa"] * p["b"] * (A / R))[:, None, None] * nvec[:, :, None] * i2[None, None, :]
        kappa = np.zeros(n)
        kappa[pl] = np.sqrt(2.0 / 3.0) * R / (np.sqrt(2.0) * A)
        h_in = np.concatenate([np.zeros((n, 1)), e_in], axis=1).reshape(-1)
        out.append({"name": str(z[q + "name"]), "hyperbolic": hyper, "params": p, "grad": z[q + "grad"], "stress_in": z[q + "stress_in"],
                    "history_in": h_in, "plastic": pl,
                    "expected": {"stress": z[q + "stress_out"], "eps_p": z[q + "eps_p_out"], "tangent": T.reshape(-1), "kappa": kappa}})
    return out


def dp_volumetric_cases():
    """tests/golden/drucker_prager_volumetric_flow.npz -> DruckerPrager3D with b_flow != 0 (non-associated and associated).
    From point-by-point outputs of the imported Python VonMises3D (radial return onto the radius the generator chose) and
    the Rust text:

    * plastic multiplier: the Rust flow direction is g = b_flow I2 + s / (2 sqrt(J2)) (drucker_prager_classic.rs:96-103); its
      deviatoric part has length 1/sqrt(2), the Python return moves gamma along a unit direction -> del_lambda = sqrt(2) gamma
      = sqrt(3) alpha_python;
    * stress: the Python one (deviatoric return) - 3 kappa b_flow del_lambda I2 (the volumetric part of C g);
    * plastic strain: the Python one + b_flow del_lambda I2;
    * THE CHECK THAT THE RADIUS WAS THE RIGHT ONE: the state so assembled satisfies the Rust yield function
      sqrt(J2) + b I1 - a = 0 (:88) at every plastic point -- asserted here, to 1e-12 a;
    * tangent: classic surface: Python's + 2 mu n (x) n - (3 kappa b_flow I2 + sqrt(2) mu n) (x) (sqrt(2) mu n + 3 kappa b I2) / (mu + 9 kappa b b_flow)
      (the derivative of del_lambda = f_trial / (mu + 9 kappa b b_flow) replaces the one of the fixed-radius return); both surfaces in
      the general form stated next to the code below;
    * hardening variable: + sqrt(2/3) |g| = sqrt(2/3) sqrt(3 b_flow^2 + 1/2) (general.rs:208: no del_lambda)."""
    z = np.load(os.path.join(GOLDEN, "drucker_prager_volumetric_flow.npz"))
    i2 = np.array([1.0, 1.0, 1.0, 0.0, 0.0, 0.0])
    out = []
    for i in range(int(z["n_calls"])):
        q = f"c{i}."
        p = dict(zip([str(k) for k in z[q + "param_keys"]], [float(v) for v in z[q + "param_vals"]]))
        mu, ka, a, b, bf = p["mu"], p["kappa"], p["a"], p["b"], p["b_flow"]
        dd = p.get("d", 0.0)
        hyper = dd != 0.0
        n = z[q + "grad"].size // 9
        al = z[q + "alpha_py"]
        pl = al > 0.0
        s_py = z[q + "stress_py"].reshape(n, 6)
        dev_py = s_py.copy()
        dev_py[:, :3] -= s_py[:, :3].mean(axis=1, keepdims=True)
        j2_1 = 0.5 * (dev_py**2).sum(axis=1)
        # hyperbolic: the deviatoric part of g is s / (2 Q), Q = sqrt(J2 + d^2) (hyperbolic.rs:87-100) -> del_lambda = sqrt(2) gamma Q / sqrt(J2)
        dl = np.sqrt(3.0) * al * (np.sqrt(j2_1 + dd * dd) / np.sqrt(np.where(j2_1 > 0.0, j2_1, 1.0)) if hyper else 1.0)
        stress = s_py - (3.0 * ka * bf * dl)[:, None] * i2[None, :]
        e_in = z[q + "eps_p_in"].reshape(n, 6)
        eps_p = e_in + z[q + "deps_py"].reshape(n, 6) + (bf * dl)[:, None] * i2[None, :]
        dev = stress.copy()
        dev[:, :3] -= stress[:, :3].mean(axis=1, keepdims=True)
        f_new = np.sqrt(0.5 * (dev[pl] ** 2).sum(axis=1) + dd * dd) + b * stress[pl, :3].sum(axis=1) - a
        assert np.abs(f_new).max() <= 1e-11 * a, np.abs(f_new).max()   # the assembled state lies ON the Rust yield surface
        nvec = dev[pl] / np.linalg.norm(dev[pl], axis=1, keepdims=True)
        # tangent, both surfaces: sigma_1 = (p_trial - 3 kappa b_flow del_lambda) I2 + sqrt(2) r n with r = sqrt(J2_1), Q = sqrt(r^2 + d^2)
        # = A + c del_lambda (A = a - b I1_trial, c = 9 kappa b b_flow) and r (1 + mu del_lambda / Q) = sqrt(J2_trial); the Python
        # tangent is the derivative at FIXED r, so T = T_py - 3 kappa b_flow I2 (x) w + sqrt(2) (Q / r) n (x) (c w - 3 kappa b I2) with
        # w = d del_lambda / d eps = (sqrt(2) mu n + 3 kappa b G I2) / (c G + mu r / Q), G = (Q / r)(1 + mu del_lambda / Q) - mu del_lambda r / Q^2
        # (classic surface, d = 0: Q = r, G = 1 -- the closed form in the docstring)
        r_ = np.sqrt(j2_1[pl])
        Q = np.sqrt(j2_1[pl] + dd * dd)
        c = 9.0 * ka * b * bf
        G = (Q / r_) * (1.0 + mu * dl[pl] / Q) - mu * dl[pl] * r_ / (Q * Q)
        w = (np.sqrt(2.0) * mu * nvec + (3.0 * ka * b * G)[:, None] * i2[None, :]) / (c * G + mu * r_ / Q)[:, None]
        T = z[q + "tangent_py"].reshape(n, 6, 6).copy()
        T[pl] += -3.0 * ka * bf * i2[None, :, None] * w[:, None, :] \
            + (np.sqrt(2.0) * Q / r_)[:, None, None] * nvec[:, :, None] * (c * w - 3.0 * ka * b * i2[None, :])[:, None, :]
        T = T.reshape(-1)
        # |g|^2 = 3 b_flow^2 + J2 / (2 Q^2)
        g_norm = np.sqrt(3.0 * bf * bf + 0.5 * j2_1 / (j2_1 + dd * dd)) if hyper else np.sqrt(3.0 * bf * bf + 0.5)
        kappa = np.where(pl, np.sqrt(2.0 / 3.0) * g_norm, 0.0)
        h_in = np.concatenate([np.zeros((n, 1)), e_in], axis=1).reshape(-1)
        if hyper:
            del p["d"]
            p = {"mu": mu, "kappa": ka, "a": a, "b": b, "d": dd, "b_flow": bf}
        out.append({"name": str(z[q + "name"]), "hyperbolic": hyper, "params": p, "grad": z[q + "grad"], "stress_in": z[q + "stress_in"],
                    "history_in": h_in, "plastic": pl,
                    "expected": {"stress": stress.reshape(-1), "eps_p": eps_p.reshape(-1), "tangent": T, "kappa": kappa}})
    return out
