"""Pin the CPU oracle (oracle/numpy_oracle.py) against

* golden vectors captured from the imported reference (oracle/gen_golden.py),
* the reference's own known-answer tests for this path
  (tests/models/test_conversions.py:14-44; comfe-rs/src/mandel.rs:181-243,
  comfe-rs/src/consts.rs:121-131),
* closed-form identities for the comfe-rs Mises law whose plastic values the
  reference does not pin.

CPU only (no GPU, no reference import).
"""

import numpy as np
import pytest
from golden_util import load_calls, rel_err

from oracle import numpy_oracle as O

TOL_LE = 1e-13  # oracle vs reference on the same machine: rounding-level only
TOL_PL = 1e-11


def _run(fn, c):
    s, t, h = c.fresh()
    fn(c.params, 0.0, c.del_t, c.grad.copy(), s, t, h)
    return s, t, h


def _check(c, s, t, h, tol):
    assert rel_err(s, c.stress_out) <= tol, c.name
    assert rel_err(t, c.tangent_out) <= tol, c.name
    if c.hist_out is not None:
        for k in c.hist_out:
            assert rel_err(h[k], c.hist_out[k]) <= tol, (c.name, k)


# ---- known-answer tests taken from the reference's own tests ---------------


def test_strain_from_grad_u_known_answer():
    # tests/models/test_conversions.py:29-44
    g = np.array([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0], [7.0, 8.0, 9.0]])
    e = O.strain_from_grad_u_full(g.reshape(-1))
    ref = np.array([1.0, 5.0, 9.0, 0.5 * (2.0 + 4.0) * 2**0.5, 0.5 * (3.0 + 7.0) * 2**0.5, 0.5 * (6.0 + 8.0) * 2**0.5])
    assert np.allclose(e, ref)
    # comfe-rs/src/mandel.rs:196-205,223-229 (Rust twin, FRAC_1_SQRT_2)
    e_rs = O.strain_from_grad_u_full(g.reshape(-1), O.F_RS)
    ref_rs = np.array([1.0, 5.0, 9.0, 6.0 * O.F_RS, 10.0 * O.F_RS, 14.0 * O.F_RS])
    assert np.linalg.norm(e_rs - ref_rs) < 1e-14


def test_strain_factor_bit_patterns():
    # the two references differ by one ULP
    assert O.F_PY.hex() == "0x1.6a09e667f3bccp-1"
    assert O.F_RS.hex() == "0x1.6a09e667f3bcdp-1"


def test_strain_from_grad_u_golden():
    z = np.load(__import__("os").path.join(__import__("golden_util").GOLDEN, "strain_from_grad_u.npz"))
    assert np.array_equal(O.strain_from_grad_u_full(z["grad"]), z["strain"])
    assert np.array_equal(O.strain_from_grad_u_full(z["grad_ka"]), z["strain_ka"])


def test_comfe_projector_identities():
    # comfe-rs/src/consts.rs:121-131
    _, soo, pvol, pdev = O.comfe_projections()
    assert np.linalg.norm(soo @ pdev) < 1e-14
    assert np.linalg.norm(pvol @ pdev) < 1e-14
    assert np.linalg.norm(pvol @ pvol - pvol) < 1e-14
    assert np.linalg.norm(pdev @ pdev - pdev) < 1e-14


def test_comfe_tangent_known_answer():
    # comfe-rs/src/mandel.rs:181-221
    MU, KAPPA = 1.2e9, 1.6e9
    LAM = KAPPA - 2.0 * MU / 3.0
    T = np.zeros((6, 6))
    T[:3, :3] = LAM
    for i in range(3):
        T[i, i] = 2.0 * MU + LAM
        T[3 + i, 3 + i] = 2.0 * MU
    C = O.comfe_isotropic_elastic_tangent(MU, KAPPA)
    assert np.linalg.norm(C - T) < 1e-14 + 1e-14 * np.linalg.norm(T)
    Ci = O.comfe_isotropic_elastic_tangent_inv(MU, KAPPA)
    assert np.linalg.norm(C @ Ci - np.eye(6)) < 1e-14


# ---- golden vectors from the imported reference ----------------------------


RANDOM_PARAMETERS = [(kind, c) for kind in ("linear_elasticity", "von_mises_3d", "spring_maxwell", "spring_kelvin")
                     for c in load_calls(f"random_parameters_{kind}.npz")]


@pytest.mark.parametrize("kind,c", RANDOM_PARAMETERS, ids=[f"{k}-{c.name}" for k, c in RANDOM_PARAMETERS])
def test_random_parameters_golden(kind, c):
    """eight random parameter sets per law over many decades (oracle/gen_golden.py: main_random_parameters), outputs of the
    imported reference"""
    _check(c, *_run(getattr(O, kind), c), TOL_PL if kind == "von_mises_3d" else TOL_LE)


@pytest.mark.parametrize("c", load_calls("linear_elasticity.npz"), ids=lambda c: c.name)
def test_linear_elasticity_golden(c):
    _check(c, *_run(O.linear_elasticity, c), TOL_LE)


@pytest.mark.parametrize("c", load_calls("spring_maxwell.npz"), ids=lambda c: c.name)
def test_spring_maxwell_golden(c):
    _check(c, *_run(O.spring_maxwell, c), TOL_LE)


@pytest.mark.parametrize("c", load_calls("spring_kelvin.npz"), ids=lambda c: c.name)
def test_spring_kelvin_golden(c):
    _check(c, *_run(O.spring_kelvin, c), TOL_LE)


@pytest.mark.parametrize("c", load_calls("von_mises_3d.npz"), ids=lambda c: c.name)
def test_von_mises_golden_vectorised(c):
    _check(c, *_run(O.von_mises_3d, c), TOL_PL)


@pytest.mark.parametrize("c", load_calls("von_mises_3d.npz")[:4], ids=lambda c: c.name)
def test_von_mises_golden_loop(c):
    _check(c, *_run(O.von_mises_3d_loop, c), TOL_PL)


def test_von_mises_multistep_protocol():
    """The mixed sequence chains: committed output of step k is the input of k+1."""
    calls = {c.name: c for c in load_calls("von_mises_3d.npz")}
    for k in range(3):
        a, b = calls[f"mixed_step{k}_iter1"], calls[f"mixed_step{k + 1}_iter0"]
        assert np.array_equal(a.stress_out, b.stress_in)
        assert np.array_equal(a.hist_out["alpha"], b.hist_in["alpha"])
    # plastic fraction of the mixed case is neither 0 nor 1
    c = calls["mixed_step0_iter1"]
    s, t, h = c.fresh()
    npl, nit = O.von_mises_3d(c.params, 0.0, c.del_t, c.grad, s, t, h)
    assert 0 < npl < c.n and 3 * npl <= nit <= 6 * npl


def test_von_mises_newton_nonconvergence_raises():
    p = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}
    g = np.zeros(9)
    g[0] = np.nan
    h = {"eps_n": np.zeros(6), "alpha": np.zeros(1)}
    # NaN trial state: phitr > 0 is False -> elastic branch, no exception (reference behaviour)
    O.von_mises_3d_loop(p, 0, 1, g, np.zeros(6), np.zeros(36), h)


# ---- comfe-rs laws: identities that follow from the reference code ---------

P_RS = {"mu": 80769.0, "kappa": 175000.0, "y_0": 1200.0, "h": 200.0}


def _dev(s):
    d = s.reshape(-1, 6).copy()
    d[:, :3] -= d[:, :3].sum(axis=1, keepdims=True) / 3
    return d


def test_comfe_mises_identities():
    rng = np.random.default_rng(3)
    n = 500
    scale = np.repeat(10 ** rng.uniform(-4, -2, size=n), 9)
    g = rng.normal(size=9 * n) * scale
    s0 = rng.normal(scale=50.0, size=6 * n)
    hist = {"history": np.zeros(7 * n)}
    hist["history"].reshape(-1, 7)[:, 0] = rng.uniform(0, 0.02, size=n)
    a0 = hist["history"].reshape(-1, 7)[:, 0].copy()
    s = s0.copy()
    t = np.zeros(36 * n)
    npl = O.comfe_mises_plasticity(P_RS, 0, 1, g, s, t, hist)
    assert 0 < npl < n
    a1 = hist["history"].reshape(-1, 7)[:, 0]
    pl = a1 > a0
    assert pl.sum() == npl
    de = O.strain_from_grad_u_full(g, O.F_RS).reshape(-1, 6)
    # pressure update: tr(sigma_new)/3 = p0 + kappa tr(d_eps)   (mises_plasticity.rs:85-87)
    p0 = s0.reshape(-1, 6)[:, :3].sum(axis=1) / 3
    p1 = s.reshape(-1, 6)[:, :3].sum(axis=1) / 3
    assert np.allclose(p1, p0 + P_RS["kappa"] * de[:, :3].sum(axis=1), rtol=1e-12, atol=1e-9)
    # consistency: after a plastic step sqrt(3/2)|dev sigma| = y0 + h alpha_new
    q1 = np.sqrt(1.5 * (_dev(s) ** 2).sum(axis=1))
    assert np.allclose(q1[pl], P_RS["y_0"] + P_RS["h"] * a1[pl], rtol=1e-11)
    # elastic points stay inside the yield surface and keep their history
    assert np.all(q1[~pl] < P_RS["y_0"] + P_RS["h"] * a1[~pl])
    assert np.array_equal(a1[~pl], a0[~pl])
    # alpha increment (mises_plasticity.rs:105)
    s_tr = _dev(s0) + 2 * P_RS["mu"] * _dev(de)
    q = np.sqrt(1.5 * (s_tr**2).sum(axis=1))
    assert np.allclose((a1 - a0)[pl], ((q - (P_RS["y_0"] + P_RS["h"] * a0)) / (3 * P_RS["mu"] + P_RS["h"]))[pl], rtol=1e-9)
    # elastic tangent = kappa 1x1 + 2 mu P_dev = isotropic tangent
    C = O.comfe_isotropic_elastic_tangent(P_RS["mu"], P_RS["kappa"])
    assert np.allclose(t.reshape(-1, 36)[~pl], C.reshape(36), rtol=1e-14)


def test_comfe_le_agrees_with_python_le():
    """Elastic steps of comfe LE == Python LE up to the 1-ULP strain factor."""
    rng = np.random.default_rng(4)
    n = 300
    E, nu = 42.0, 0.3
    mu, kappa = E / (2 * (1 + nu)), E / (3 * (1 - 2 * nu))  # tests/models/test_elasticity.py:347-352
    g = rng.normal(scale=1e-3, size=9 * n)
    s_py, s_rs = rng.normal(size=6 * n), None
    s_rs = s_py.copy()
    t_py, t_rs = np.zeros(36 * n), np.zeros(36 * n)
    O.linear_elasticity({"E": E, "nu": nu}, 0, 1, g, s_py, t_py)
    O.comfe_linear_elasticity({"mu": mu, "kappa": kappa}, 0, 1, g, s_rs, t_rs)
    assert rel_err(s_rs, s_py) < 1e-13
    assert rel_err(t_rs, t_py) < 1e-13
    # tangent=None is accepted by the Rust entry (interfaces.rs:383-394)
    s2 = s_py.copy()
    O.comfe_linear_elasticity({"mu": mu, "kappa": kappa}, 0, 1, g, s2, None)


def test_perfect_plasticity_cross_check():
    """with y00 = y0 and h = 0 both Mises laws give the same stress
    and the same alpha."""
    rng = np.random.default_rng(5)
    n = 400
    g = rng.normal(scale=5e-3, size=9 * n)
    p_py = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 1200.0, "p_w": 200.0}
    p_rs = {"mu": 80769.0, "kappa": 175000.0, "y_0": 1200.0, "h": 0.0}
    s_py, s_rs = np.zeros(6 * n), np.zeros(6 * n)
    h_py = {"eps_n": np.zeros(6 * n), "alpha": np.zeros(n)}
    h_rs = {"history": np.zeros(7 * n)}
    O.von_mises_3d(p_py, 0, 1, g, s_py, np.zeros(36 * n), h_py)
    O.comfe_mises_plasticity(p_rs, 0, 1, g, s_rs, np.zeros(36 * n), h_rs)
    assert rel_err(s_rs, s_py) < 1e-10
    assert rel_err(h_rs["history"].reshape(-1, 7)[:, 0], h_py["alpha"]) < 1e-9


# ---- 3D -> 1D/2D wrappers (models/utils.py:211-412) --------------------------


def test_wrappers_golden():
    from wrappers_util import PARAMS, load_sequences

    fns = {"le": O.linear_elasticity, "vm": O.von_mises_3d_loop, "maxwell": O.spring_maxwell}
    for kind, lname, calls in load_sequences():
        w = O.From3D(kind, fns[lname], PARAMS[lname])
        for c in calls:  # one wrapper instance per sequence: the cached 3-D arrays persist
            s = c["stress_in"].copy()
            t = np.full_like(c["tangent_out"], np.nan)
            h = None if c["hist_in"] is None else {k: v.copy() for k, v in c["hist_in"].items()}
            w.evaluate(0.0, 2.0, c["grad"], s, t, h)
            assert rel_err(s, c["stress_out"]) <= 1e-13 and rel_err(t, c["tangent_out"]) <= 1e-13, (kind, lname)
            if h is not None:
                for k in h:
                    assert rel_err(h[k], c["hist_out"][k]) <= 1e-13


def test_constraints_golden():
    """LE / Maxwell / Kelvin under the four non-FULL constraints, against the reference."""
    from wrappers_util import load_constraint_calls

    for c in load_constraint_calls() + load_constraint_calls("random_parameters_constraints.npz"):
        s, t = c["stress_in"].copy(), np.full_like(c["tangent_out"], np.nan)
        h = None if c["hist_in"] is None else {k: v.copy() for k, v in c["hist_in"].items()}
        O.MODELS_C[c["law"]](c["params"], c["constraint"], 0.0, c["del_t"], c["grad"], s, t, h)
        assert rel_err(s, c["stress_out"]) <= 1e-13 and rel_err(t, c["tangent_out"]) <= 1e-13, (c["law"], c["constraint"])
        if h is not None:
            for k in h:
                assert rel_err(h[k], c["hist_out"][k]) <= 1e-13


# ---- f4: Drucker-Prager general return mapping (parity unpinned: identities only) -------------

DP_P = {"mu": 80769.0, "kappa": 175000.0, "a": 100.0, "b": 0.05, "b_flow": 0.02}
DP_H = dict(DP_P, d=40.0)


def dp_inputs(n, seed, smax=-2.3):
    """Mostly isochoric strain increments on a compressive prestress: the trial states stay away from
    the tip of the classic surface (i_1 < a / b) while 10-90 % of the points yield."""
    rng = np.random.default_rng(seed)
    g = (rng.normal(size=9 * n) * np.repeat(10 ** rng.uniform(-4, smax, size=n), 9)).reshape(-1, 9)
    tr = g[:, [0, 4, 8]].sum(axis=1)
    g[:, [0, 4, 8]] -= (0.95 * tr / 3.0)[:, None]
    s = rng.normal(scale=50.0, size=6 * n)
    s.reshape(-1, 6)[:, :3] -= 1000.0
    h = rng.normal(scale=1e-4, size=7 * n)
    h.reshape(-1, 7)[:, 0] = rng.uniform(0, 0.1, size=n)
    return g.reshape(-1).copy(), s, {"history": h}


@pytest.mark.parametrize("hyper", [False, True])
def test_drucker_prager_identities(hyper):
    p = DP_H if hyper else DP_P
    n = 1500
    g, s0, h0 = dp_inputs(n, 3)
    s, t, h = s0.copy(), np.zeros(36 * n), {"history": h0["history"].copy()}
    npl, nit = O.comfe_drucker_prager(p, 0, 1, g, s, t, h, hyperbolic=hyper)
    assert 0.1 * n < npl < 0.9 * n and npl <= nit <= 8 * npl
    f1, *_ = O._dp_state(p, hyper, s.reshape(-1, 6))
    a1, a0 = h["history"].reshape(-1, 7)[:, 0], h0["history"].reshape(-1, 7)[:, 0]
    pl = a1 != a0
    assert pl.sum() == npl
    # return mapping: plastic points end on the yield surface, elastic points inside
    assert np.abs(f1[pl]).max() < 1e-7 and f1[~pl].max() <= 0.0
    # elastic points: sigma = sigma_0 + E d_eps, tangent = E, history untouched
    E = O.comfe_isotropic_elastic_tangent(p["mu"], p["kappa"])
    de = O.strain_from_grad_u_full(g, O.F_RS).reshape(-1, 6)
    assert np.allclose(s.reshape(-1, 6)[~pl], s0.reshape(-1, 6)[~pl] + de[~pl] @ E.T, rtol=1e-13, atol=1e-10)
    assert np.array_equal(t.reshape(-1, 36)[~pl], np.tile(E.T.reshape(36), ((~pl).sum(), 1)))
    assert np.array_equal(h["history"].reshape(-1, 7)[~pl], h0["history"].reshape(-1, 7)[~pl])
    # plastic strain increment = d_eps - E^-1 (sigma_1 - sigma_0) points along the flow direction g
    _, _, gdir, *_ = O._dp_state(p, hyper, s.reshape(-1, 6)[pl])
    dep = (h["history"].reshape(-1, 7)[pl, 1:] - h0["history"].reshape(-1, 7)[pl, 1:])
    cos = np.einsum("ij,ij->i", dep, gdir) / (np.linalg.norm(dep, axis=1) * np.linalg.norm(gdir, axis=1))
    assert cos.min() > 1 - 1e-8
    # kappa quirk kept as read: alpha_1 = alpha_0 + sqrt(2/3) |g|
    assert np.allclose(a1[pl] - a0[pl], np.sqrt(2 / 3) * np.linalg.norm(gdir, axis=1), rtol=1e-7)
    # consistent tangent = d sigma / d eps (central differences on three plastic points)
    for i in np.nonzero(pl)[0][:3]:
        T = t.reshape(-1, 36)[i].reshape(6, 6)
        Tfd = np.zeros((6, 6))
        for j in range(6):
            for sgn in (1, -1):
                gp = g[9 * i : 9 * i + 9].copy()
                if j < 3:
                    gp[[0, 4, 8][j]] += sgn * 1e-7
                else:
                    u, v = [(1, 3), (2, 6), (5, 7)][j - 3]
                    gp[u] += sgn * 1e-7 / (2 * O.F_RS)
                    gp[v] += sgn * 1e-7 / (2 * O.F_RS)
                ss = s0[6 * i : 6 * i + 6].copy()
                O.comfe_drucker_prager(p, 0, 1, gp, ss, np.zeros(36), {"history": h0["history"][7 * i : 7 * i + 7].copy()}, hyperbolic=hyper)
                Tfd[:, j] += sgn * ss / 2e-7
        assert np.abs(T - Tfd).max() / np.abs(T).max() < 1e-6


def test_drucker_prager_tip_assertion():
    g, s, h = dp_inputs(10, 1)
    s.reshape(-1, 6)[3, :3] = 700.0  # i_1 = 2100 > a / b = 2000
    with pytest.raises(O.DruckerPragerTip):
        O.comfe_drucker_prager(DP_P, 0, 1, g, s, np.zeros(360), h)
    # the hyperbolic surface has no tip assertion
    O.comfe_drucker_prager(DP_H, 0, 1, g, s, np.zeros(360), h, hyperbolic=True)


# ---- self-derived 50-digit vectors for the comfe-rs plasticity updates (oracle/mp_pins.py) --------------------------
# NOT reference-held: a third, independent transcription of the Rust text (full 8 x 8 Newton system, mpmath) against
# which both oracles are bounded at rounding level -- it separates transcription errors from rounding, it does not lift
# "parity unpinned" (only a Rust build or vectors held by the reference would).

SELFDERIVED = [("comfe_selfderived_mises.npz", "mises", False), ("comfe_selfderived_drucker_prager_classic.npz", "dp", False),
               ("comfe_selfderived_drucker_prager_hyperbolic.npz", "dp", True)]


def _selfderived(fname):
    import os

    from golden_util import GOLDEN

    z = np.load(os.path.join(GOLDEN, fname))
    return z, dict(zip([str(k) for k in z["param_keys"]], [float(v) for v in z["param_vals"]]))


def _worst_point(a, b, d):
    """largest per-point relative error: max_i ( max|a_i - b_i| / max|b_i| )"""
    a, b = a.reshape(-1, d), b.reshape(-1, d)
    return float((np.abs(a - b).max(axis=1) / np.abs(b).max(axis=1)).max())


@pytest.mark.parametrize("which", ["numpy", "c"])
@pytest.mark.parametrize("fname,kind,hyper", SELFDERIVED)
def test_oracles_against_the_50_digit_transcription(fname, kind, hyper, which):
    from oracle import c_oracle as CO

    z, p = _selfderived(fname)
    mod = O if which == "numpy" else CO
    s, t, h = z["stress_in"].copy(), np.full(z["tangent_out"].size, np.nan), {"history": z["hist_in"].copy()}
    if kind == "mises":
        mod.comfe_mises_plasticity(p, 0.0, 1.0, z["grad"], s, t, h)
    else:
        mod.comfe_drucker_prager(p, 0.0, 1.0, z["grad"], s, t, h, hyperbolic=hyper)
    pl = z["plastic"]
    assert 0.2 * pl.size < pl.sum() < 0.8 * pl.size  # both branches are in the sample
    hv, ho = h["history"].reshape(-1, 7), z["hist_out"].reshape(-1, 7)
    assert np.array_equal(hv[:, 0] != z["hist_in"].reshape(-1, 7)[:, 0], pl)  # the same points yield
    # PER POINT, relative to the point's own largest entry: rounding level (measured 2e-16 .. 1e-14), bound 1e-13
    assert _worst_point(s, z["stress_out"], 6) <= 1e-13
    assert _worst_point(t, z["tangent_out"], 36) <= 1e-13
    assert np.abs(hv[:, 0] - ho[:, 0]).max() <= 1e-13 * np.abs(ho[:, 0]).max()
    assert _worst_point(hv[:, 1:].copy(), ho[:, 1:].copy(), 6) <= 1e-13


def test_selfderived_fixtures_reproduce():
    """the committed vectors are what oracle/mp_pins.py computes (first points of every file; needs mpmath)"""
    pytest.importorskip("mpmath")
    from oracle import mp_pins

    for fname, kind, hyper in SELFDERIVED:
        z, p = _selfderived(fname)
        k = 6
        g, s, h = z["grad"][: 9 * k], z["stress_in"][: 6 * k], z["hist_in"][: 7 * k]
        out = mp_pins.run_mises(g, s, h, p) if kind == "mises" else mp_pins.run_dp(g, s, h, p, hyper)
        assert np.array_equal(out[0], z["stress_out"][: 6 * k]) and np.array_equal(out[1], z["tangent_out"][: 36 * k])
        assert np.array_equal(out[2], z["hist_out"][: 7 * k]) and np.array_equal(out[3], z["plastic"][:k])


# ---- a8: the plastic branch of comfe-rs MisesPlasticity3D against the IMPORTED Python reference in its linear-hardening limit ----

from golden_util import check_mises_limit, mises_limit_cases  # noqa: E402

MISES_LIMIT = mises_limit_cases()


@pytest.mark.parametrize("oracle", ["numpy", "c"])
@pytest.mark.parametrize("case", MISES_LIMIT, ids=[c["name"] for c in MISES_LIMIT])
def test_comfe_mises_reproduces_the_reference_von_mises_in_the_linear_hardening_limit(case, oracle):
    """The reference holds no vector for the Rust law's plastic branch, but its Python VonMises3D tends to the same law for
    w -> 0: outputs of the imported reference pin stress, alpha, plastic strain (x sqrt(2/3)) and the tangent (+ the
    rank-one term of the Rust text) of both restatements (golden_util.mises_limit_cases)."""
    from oracle import c_oracle as CO

    fn = O.comfe_mises_plasticity if oracle == "numpy" else CO.comfe_mises_plasticity
    n = case["grad"].size // 9
    s, t, h = case["stress_in"].copy(), np.full(36 * n, np.nan), {"history": case["history_in"].copy()}
    fn(case["params"], 0.0, 1.0, case["grad"].copy(), s, t, h)
    check_mises_limit(case, s, t, h["history"])


# ---- f4: the general return mapping against the IMPORTED Python reference on the J2 sub-family (b = b_flow = 0) ----

from golden_util import check_dp_j2, dp_j2_cases, dp_pressure_cases, dp_volumetric_cases  # noqa: E402

DP_J2 = dp_j2_cases()
DP_PRESSURE = dp_pressure_cases()
DP_VOLUMETRIC = dp_volumetric_cases()


@pytest.mark.parametrize("oracle", ["numpy", "c"])
@pytest.mark.parametrize("case", DP_VOLUMETRIC, ids=[c["name"] for c in DP_VOLUMETRIC])
def test_general_return_mapping_with_volumetric_flow_against_pointwise_reference_calls(case, oracle):
    """b_flow != 0 (non-associated, associated, a steep surface): the deviatoric part of the return is the reference's radial
    return, the volumetric part follows from the reference's own plastic multiplier through the Rust flow rule, and the
    assembled state is verified to lie on the Rust yield surface (golden_util.dp_volumetric_cases) -- stress, plastic strain
    and the full consistent tangent of the 8 x 8 Newton machinery, both surfaces."""
    from oracle import c_oracle as CO

    fn = O.comfe_drucker_prager if oracle == "numpy" else CO.comfe_drucker_prager
    n = case["grad"].size // 9
    s, t, h = case["stress_in"].copy(), np.full(36 * n, np.nan), {"history": case["history_in"].copy()}
    fn(case["params"], 0.0, 1.0, case["grad"].copy(), s, t, h, hyperbolic=case["hyperbolic"])
    check_dp_j2(case, s, t, h["history"])


@pytest.mark.parametrize("oracle", ["numpy", "c"])
@pytest.mark.parametrize("case", DP_PRESSURE, ids=[c["name"] for c in DP_PRESSURE])
def test_general_return_mapping_pressure_dependence_against_pointwise_reference_calls(case, oracle):
    """b != 0, b_flow = 0: every point returns radially onto a J2 cylinder whose radius follows from ITS trial pressure --
    stress and plastic strain are those of the imported Python VonMises3D called point by point with that yield stress, the
    tangent is the Python one plus a non-symmetric rank-one term (golden_util.dp_pressure_cases), which also pins the
    orientation of the stored tangent.  Still without a Python counterpart: b_flow != 0 (volumetric plastic flow)."""
    from oracle import c_oracle as CO

    fn = O.comfe_drucker_prager if oracle == "numpy" else CO.comfe_drucker_prager
    n = case["grad"].size // 9
    s, t, h = case["stress_in"].copy(), np.full(36 * n, np.nan), {"history": case["history_in"].copy()}
    fn(case["params"], 0.0, 1.0, case["grad"].copy(), s, t, h, hyperbolic=case["hyperbolic"])
    check_dp_j2(case, s, t, h["history"])
    transposed = case["expected"]["tangent"].reshape(n, 6, 6).transpose(0, 2, 1).reshape(-1)
    assert rel_err(t, transposed) > 0.1  # the term is not symmetric: the other orientation is far off


@pytest.mark.parametrize("oracle", ["numpy", "c"])
@pytest.mark.parametrize("case", DP_J2, ids=[c["name"] for c in DP_J2])
def test_general_return_mapping_reproduces_the_reference_von_mises_for_b_zero(case, oracle):
    """No reference test touches plasticity/general.rs, but for b = b_flow = 0 its closest-point projection onto the
    Drucker-Prager surfaces is the radial return of the Python VonMises3D without hardening: outputs of the imported
    reference pin stress, plastic strain and the consistent tangent of the 8 x 8 Newton machinery
    (golden_util.dp_j2_cases).  The pressure-dependent terms (b, b_flow != 0) stay with the identities and the 50-digit
    transcription above."""
    from oracle import c_oracle as CO

    fn = O.comfe_drucker_prager if oracle == "numpy" else CO.comfe_drucker_prager
    n = case["grad"].size // 9
    s, t, h = case["stress_in"].copy(), np.full(36 * n, np.nan), {"history": case["history_in"].copy()}
    fn(case["params"], 0.0, 1.0, case["grad"].copy(), s, t, h, hyperbolic=case["hyperbolic"])
    check_dp_j2(case, s, t, h["history"])
