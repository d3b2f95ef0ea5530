#!/usr/bin/env python3
"""Self-derived high-precision vectors for the two comfe-rs plasticity updates the reference holds NO known answer for.

TEST INFRASTRUCTURE (like everything under oracle/): nothing in the product imports this.

The reference's own tests never pin the plastic branch of comfe-rs ``MisesPlasticity3D`` nor anything of the general
return mapping, and the crate cannot be built in this image (no rustc).  What CAN be separated is
transcription error from rounding error: this script transcribes both updates statement by statement from the Rust
sources into 50-digit arithmetic (mpmath) --

    comfe-rs/src/mises_plasticity.rs:58-126                 (closed-form radial return, linear hardening)
    comfe-rs/src/plasticity/general.rs:105-266              (Newton on sigma(6), lambda, kappa; LU; tangent from the inverse)
    comfe-rs/src/plasticity/drucker_prager_classic.rs:72-116, drucker_prager_hyperbolic.rs:74-112   (set_model_state)
    comfe-rs/src/plasticity/general.rs:38-74                (update_newton_matrix)
    comfe-rs/src/mandel.rs:30-33,54-69,126-141, consts.rs   (vol_dev, trace_dev, mises_norm, elastic tangent and its inverse)

-- with the full 8 x 8 system (no invariant reduction, no closed-form inverse: a third restatement, independent of
oracle/numpy_oracle.py, oracle/oracle.c and the HIP kernels), and stores float64 inputs with the float64-rounded
50-digit results as ``tests/golden/comfe_selfderived_*.npz``.  ``tests/test_oracle_golden.py`` bounds both oracles
against them.  The vectors are SELF-DERIVED (the file name says so): they show that the oracles compute what the
Rust text says to rounding level; they are not outputs of the reference, and "parity unpinned" stays in force for
these laws.

    python oracle/mp_pins.py            # rewrites the two fixture files (needs mpmath)
"""

from __future__ import annotations

import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
FRAC_1_SQRT_2 = float.fromhex("0x1.6a09e667f3bcdp-1")  # core::f64::consts::FRAC_1_SQRT_2 (mandel.rs:147)

MISES_P = {"mu": 80769.0, "kappa": 175000.0, "y_0": 1200.0, "h": 200.0}  # tests/models/test_plasticity.py:26-31
DP_P = {"mu": 80769.0, "kappa": 175000.0, "a": 100.0, "b": 0.05, "b_flow": 0.02}
DP_H = dict(DP_P, d=40.0)


def strain_f64(grad):
    """nonsymmetric_tensor_to_mandel (mandel.rs:143-171) in float64, as evaluate_model does before the law sees anything
    (interfaces.rs:441-455): the high-precision part starts from the SAME float64 strain increment."""
    g = np.asarray(grad, dtype=np.float64).reshape(-1, 9)
    f = FRAC_1_SQRT_2
    return np.stack([g[:, 0], g[:, 4], g[:, 8], f * (g[:, 1] + g[:, 3]), f * (g[:, 2] + g[:, 6]), f * (g[:, 5] + g[:, 7])], axis=1)


def _mp():
    import mpmath as mp

    mp.mp.dps = 50
    return mp


def _consts(mp):
    sym_id = mp.matrix([1, 1, 1, 0, 0, 0])
    soo = sym_id * sym_id.T
    p_vol = soo / 3
    p_dev = mp.eye(6) - p_vol
    return sym_id, soo, p_vol, p_dev


def mises_point(mp, p, de, sig, hist):
    """mises_plasticity.rs:58-126 for one point.  Returns (stress[6], tangent[36] as `.data.0`, history[7], plastic)."""
    sym_id, soo, _, p_dev = _consts(mp)
    mu, kappa, y_0, h = (mp.mpf(p[k]) for k in ("mu", "kappa", "y_0", "h"))
    alpha = mp.mpf(hist[0])
    de, sig = mp.matrix([mp.mpf(x) for x in de]), mp.matrix([mp.mpf(x) for x in sig])
    p_0 = (sig[0] + sig[1] + sig[2]) / 3                      # vol_dev
    s_0 = sig - p_0 * sym_id
    eps_trace = de[0] + de[1] + de[2]                         # trace_dev
    eps_dev = de - (eps_trace / 3) * sym_id
    p_1 = p_0 + kappa * eps_trace
    s_tr = s_0 + (2 * mu) * eps_dev
    dev = s_tr - ((s_tr[0] + s_tr[1] + s_tr[2]) / 3) * sym_id  # mises_norm: deviator once more
    s_tr_eq = mp.sqrt(3 * (mp.mpf(1) / 2 * sum(x * x for x in dev)))
    sigma_y = y_0 + h * alpha
    out_h = [mp.mpf(x) for x in hist]
    if s_tr_eq < sigma_y:
        stress = p_1 * sym_id + s_tr
        tangent = kappa * soo + (2 * mu) * p_dev
        return stress, tangent, out_h, False
    del_alpha = (s_tr_eq - sigma_y) / (3 * mu + h)
    del_gamma = mp.sqrt(mp.mpf(3) / 2) * del_alpha
    theta = 1 - (3 * mu * del_alpha) / s_tr_eq
    n = s_tr / s_tr_eq
    for i in range(6):
        out_h[1 + i] += del_gamma * n[i]
    out_h[0] += del_alpha
    stress = p_1 * sym_id + theta * s_tr
    theta_bar = 1 / (1 + (h / (3 * mu))) - (1 - theta)
    tangent = kappa * soo + (2 * mu * theta) * p_dev + (2 * mu * theta_bar) * (n * n.T)
    return stress, tangent, out_h, True


class _DP:
    """set_model_state of DruckerPrager3D / DruckerPragerHyperbolic3D (drucker_prager_classic.rs:72-116,
    drucker_prager_hyperbolic.rs:74-112)."""

    def __init__(self, mp, p, hyper):
        self.mp, self.p, self.hyper = mp, {k: mp.mpf(v) for k, v in p.items()}, hyper
        self.sym_id, _, p_vol, self.p_dev = _consts(mp)
        mu, kappa = self.p["mu"], self.p["kappa"]
        self.E = (2 * mu) * self.p_dev + (3 * kappa) * p_vol                                  # isotropic_elastic_tangent
        self.Einv = (2 * (1 / (4 * mu))) * self.p_dev + (3 * (1 / (9 * kappa))) * p_vol      # ..._inv (mandel.rs:130-141)
        self.dg_dkappa = mp.zeros(6, 1)  # Default::default(): never assigned
        self.df_dkappa = mp.mpf(0)

    def set_state(self, sigma_0, sigma_1, del_eps):
        mp, p = self.mp, self.p
        i_1 = sigma_1[0] + sigma_1[1] + sigma_1[2]
        s = sigma_1 - (i_1 / 3) * self.sym_id
        j_2 = mp.mpf(1) / 2 * sum(x * x for x in s)
        if self.hyper:
            r = j_2 + p["d"] ** 2
            self.f = mp.sqrt(r) + p["b"] * i_1 - p["a"]
            df_dj_2 = mp.mpf(1) / 2 / mp.sqrt(r)
            df_dj_2j_2 = -mp.mpf(1) / 4 * r ** (-mp.mpf(3) / 2)
        else:
            if not i_1 < p["a"] / p["b"]:
                raise AssertionError("non-differentiable tip of Drucker-Prager surface reached")
            self.f = mp.sqrt(j_2) + p["b"] * i_1 - p["a"]
            df_dj_2 = mp.mpf(1) / 2 / mp.sqrt(j_2)
            df_dj_2j_2 = -mp.mpf(1) / 4 / (j_2 * mp.sqrt(j_2))
        df_dsigma = p["b"] * self.sym_id + df_dj_2 * s
        self.df_dsigma = df_dsigma.T
        self.g = df_dsigma if p["b"] == p["b_flow"] else p["b_flow"] * self.sym_id + df_dj_2 * s
        self.dg_dsigma = (s * df_dj_2j_2) * s.T + df_dj_2 * self.p_dev
        self.del_plastic_strain = del_eps - self.Einv * (sigma_1 - sigma_0)
        g_norm = mp.sqrt(sum(x * x for x in self.g))
        c = mp.sqrt(mp.mpf(2) / 3)
        self.k = c * g_norm
        self.dk_dsigma = (c / g_norm) * (self.g.T * self.dg_dsigma)   # 1 x 6
        self.dk_dkappa = ((c / g_norm) * (self.g.T * self.dg_dkappa))[0]

    def newton_matrix(self, dl):
        """update_newton_matrix (general.rs:38-74); unknowns [sigma(6), lambda, kappa]"""
        mp = self.mp
        d = mp.zeros(8, 8)
        A = mp.eye(6) + (self.E * dl) * self.dg_dsigma
        Eg = self.E * self.g
        Edk = (self.E * dl) * self.dg_dkappa
        for i in range(6):
            for j in range(6):
                d[i, j] = A[i, j]
            d[i, 6] = Eg[i]
            d[i, 7] = Edk[i]
            d[6, i] = self.df_dsigma[0, i]
            d[7, i] = (-dl) * self.dk_dsigma[0, i]
        d[6, 6] = 0
        d[6, 7] = self.df_dkappa
        d[7, 6] = -self.k
        d[7, 7] = 1 - dl * self.dk_dkappa
        return d


def dp_point(mp, p, hyper, de, sig, hist):
    """general.rs:105-266 for one point.  Returns (stress, tangent row-major as stored, history, plastic, iterations)."""
    m = _DP(mp, p, hyper)
    del_eps = mp.matrix([mp.mpf(x) for x in de])
    sigma_0 = mp.matrix([mp.mpf(x) for x in sig])
    sigma_tr = m.E * del_eps + sigma_0
    alpha_0 = mp.mpf(hist[0])
    out_h = [mp.mpf(x) for x in hist]
    m.set_state(sigma_0, sigma_tr, del_eps)
    if m.f <= 0:
        return sigma_tr, m.E, out_h, False, 0   # elastic_tangent().data.0 (symmetric)
    sol_1 = mp.matrix([sigma_tr[i] for i in range(6)] + [0, alpha_0])
    res = mp.matrix([0, 0, 0, 0, 0, 0, m.f, 0])
    dres = m.newton_matrix(mp.mpf(0))
    atol = rtol = mp.mpf("1e-8")
    i, maxit = 0, 25
    norm = lambda v: mp.sqrt(sum(x * x for x in v))  # noqa: E731
    while True:
        sol_0 = sol_1
        sol_1 = sol_0 - mp.lu_solve(dres, res)
        sigma_1, alpha_1, dl = sol_1[0:6], sol_1[7], sol_1[6]
        sigma_prev, alpha_prev, dl_prev = sol_0[0:6], sol_0[7], sol_0[6]
        m.set_state(sigma_0, sigma_1, del_eps)
        dres = m.newton_matrix(dl)
        res_sigma = sigma_1 - sigma_tr + dl * (m.E * m.g)
        res_kappa = alpha_1 - alpha_0 - m.k
        res_f = m.f
        res = mp.matrix([res_sigma[j] for j in range(6)] + [res_f, res_kappa])
        conv_res = norm(res_sigma) < atol and abs(res_kappa) < atol and abs(res_f) < atol
        conv_inc = (norm(sigma_1 - sigma_prev) < atol + rtol * norm(sigma_1) and abs(alpha_1 - alpha_prev) < atol + rtol * abs(alpha_1)
                    and abs(dl - dl_prev) < atol + rtol * abs(dl))
        if conv_res or conv_inc:
            break
        if i > maxit:
            raise RuntimeError("Plasticity3D: Newton-Raphson did not converge.")
        i += 1
    out_h[0] = alpha_1
    for j in range(6):
        out_h[1 + j] += m.del_plastic_strain[j]
    inv = mp.inverse(dres)
    pt = inv[0:6, 0:6] * m.E   # transposed, then stored column-major: flat[6 i + j] = pt[i, j]
    return sigma_1, pt, out_h, True, i + 1


def _f64(v):
    return np.array([float(x) for x in v], dtype=np.float64)


def mises_inputs(n=64, seed=11):
    rng = np.random.default_rng(seed)
    g = rng.normal(size=(n, 9)) * (10 ** rng.uniform(-4.5, -1.8, size=n))[:, None]
    s = rng.normal(scale=300.0, size=(n, 6))
    h = rng.normal(scale=1e-3, size=(n, 7))
    h[:, 0] = rng.uniform(0, 0.05, size=n)
    return g.reshape(-1), s.reshape(-1), h.reshape(-1)


def dp_inputs(n=48, seed=12):
    """mostly isochoric increments on a compressive prestress (away from the tip of the classic surface)"""
    rng = np.random.default_rng(seed)
    g = rng.normal(size=(n, 9)) * (10 ** rng.uniform(-4, -2.3, size=n))[:, None]
    g[:, [0, 4, 8]] -= (0.95 * g[:, [0, 4, 8]].sum(axis=1) / 3.0)[:, None]
    s = rng.normal(scale=50.0, size=(n, 6))
    s[:, :3] -= 1000.0
    h = rng.normal(scale=1e-4, size=(n, 7))
    h[:, 0] = rng.uniform(0, 0.1, size=n)
    return g.reshape(-1), s.reshape(-1), h.reshape(-1)


def run_mises(grad, stress, hist, p=MISES_P):
    mp = _mp()
    de = strain_f64(grad)
    n = de.shape[0]
    s_out, t_out, h_out, pl = np.empty((n, 6)), np.empty((n, 36)), np.empty((n, 7)), np.zeros(n, dtype=bool)
    for i in range(n):
        s, t, h, pl[i] = mises_point(mp, p, de[i], stress.reshape(-1, 6)[i], hist.reshape(-1, 7)[i])
        s_out[i], h_out[i] = _f64(s), _f64(h)
        t_out[i] = _f64([t[r, c] for c in range(6) for r in range(6)])  # `.data.0`: column-major
    return s_out.reshape(-1), t_out.reshape(-1), h_out.reshape(-1), pl


def run_dp(grad, stress, hist, p, hyper):
    mp = _mp()
    de = strain_f64(grad)
    n = de.shape[0]
    s_out, t_out, h_out = np.empty((n, 6)), np.empty((n, 36)), np.empty((n, 7))
    pl, its = np.zeros(n, dtype=bool), np.zeros(n, dtype=np.int64)
    for i in range(n):
        s, t, h, pl[i], its[i] = dp_point(mp, p, hyper, de[i], stress.reshape(-1, 6)[i], hist.reshape(-1, 7)[i])
        s_out[i], h_out[i] = _f64(s), _f64(h)
        t_out[i] = _f64([t[r, c] for r in range(6) for c in range(6)])
    return s_out.reshape(-1), t_out.reshape(-1), h_out.reshape(-1), pl, its


def main():
    g, s, h = mises_inputs()
    so, to, ho, pl = run_mises(g, s, h)
    np.savez(os.path.join(GOLDEN, "comfe_selfderived_mises.npz"), param_keys=np.array(list(MISES_P)), param_vals=np.array(list(MISES_P.values())),
             grad=g, stress_in=s, hist_in=h, stress_out=so, tangent_out=to, hist_out=ho, plastic=pl, digits=np.array(50))
    print(f"mises: {pl.sum()} of {pl.size} points plastic")
    for name, p, hyper in (("classic", DP_P, False), ("hyperbolic", DP_H, True)):
        g, s, h = dp_inputs(seed=12 + hyper)
        so, to, ho, pl, its = run_dp(g, s, h, p, hyper)
        np.savez(os.path.join(GOLDEN, f"comfe_selfderived_drucker_prager_{name}.npz"), param_keys=np.array(list(p)),
                 param_vals=np.array(list(p.values())), grad=g, stress_in=s, hist_in=h, stress_out=so, tangent_out=to, hist_out=ho,
                 plastic=pl, iterations=its, digits=np.array(50))
        print(f"drucker-prager {name}: {pl.sum()} of {pl.size} points plastic, Newton iterations {its[pl].min()} .. {its[pl].max()}")


if __name__ == "__main__":
    main()
