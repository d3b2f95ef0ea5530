"""Generate golden vectors from the *imported reference* (build container only).

TEST INFRASTRUCTURE.  Run here (``python oracle/gen_golden.py``) where
``/root/reference`` exists; it writes ``tests/golden/*.npz``.  The fixtures are
data only (inputs, parameters, expected outputs); the reference sources never
travel.  Import recipe: SURVEY.md Appendix A (two ``sys.modules`` stubs for
``dolfinx.common.timed`` and the compiled ``_bindings``).

Each fixture file holds a list of independent *calls*
``(params, del_t, grad, stress_in, hist_in) -> (stress_out, tangent_out, hist_out)``;
multi-step sequences follow the protocol of ``LawOnSubMesh.evaluate`` /
``IncrSmallStrainProblem.update`` (solver/_lawonsubmesh.py:72-95,
solver/_solver.py:149-159): every call starts from a fresh copy of the committed
state; a step may be evaluated several times (Newton iterations) before the last
result is committed.
"""

from __future__ import annotations

import os
import sys
import types

import numpy as np

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def import_reference():
    df = types.ModuleType("dolfinx")
    common = types.ModuleType("dolfinx.common")
    common.timed = lambda name: (lambda f: f)
    df.common = common
    sys.modules["dolfinx"] = df
    sys.modules["dolfinx.common"] = common
    b = types.ModuleType("fenics_constitutive._bindings")
    for n in ("PyDruckerPrager3D", "PyDruckerPragerHyperbolic3D", "PyLinearElasticity3D", "PyMisesPlasticity3D"):
        setattr(b, n, type(n, (), {}))
    sys.modules["fenics_constitutive._bindings"] = b
    sys.path.insert(0, REF)
    import fenics_constitutive.models as m

    return m


class Recorder:
    def __init__(self):
        self.d = {}
        self.calls = []

    def call(self, name, law, params, del_t, grad, stress_in, hist_in):
        s = stress_in.copy()
        h = None if hist_in is None else {k: v.copy() for k, v in hist_in.items()}
        n = grad.size // 9
        tan = np.full(36 * n, np.nan)
        law.evaluate(0.0, del_t, grad.copy(), s, tan, h)
        i = len(self.calls)
        self.calls.append(name)
        p = f"c{i}."
        self.d[p + "del_t"] = np.float64(del_t)
        self.d[p + "param_keys"] = np.array(list(params.keys()))
        self.d[p + "param_vals"] = np.array([float(v) for v in params.values()])
        self.d[p + "grad"] = grad.copy()
        self.d[p + "stress_in"] = stress_in.copy()
        self.d[p + "stress_out"] = s
        self.d[p + "tangent_out"] = tan
        if hist_in is not None:
            self.d[p + "hist_keys"] = np.array(list(hist_in.keys()))
            for k in hist_in:
                self.d[p + "hist_in." + k] = hist_in[k].copy()
                self.d[p + "hist_out." + k] = h[k]
        return s, h

    def save(self, fname):
        self.d["calls"] = np.array(self.calls)
        os.makedirs(OUT, exist_ok=True)
        np.savez_compressed(os.path.join(OUT, fname), **self.d)
        print(fname, len(self.calls), "calls")


def main():
    m = import_reference()
    FULL = m.StressStrainConstraint.FULL
    rng = np.random.default_rng(20251114)

    # ---- a1: strain_from_grad_u (FULL) -----------------------------------
    g = rng.normal(size=9 * 131)
    np.savez_compressed(
        os.path.join(OUT, "strain_from_grad_u.npz"),
        grad=g,
        strain=m.strain_from_grad_u(g, FULL),
        grad_ka=np.arange(1.0, 10.0),
        strain_ka=m.strain_from_grad_u(np.arange(1.0, 10.0), FULL),
    )

    # ---- a3: LinearElasticityModel ---------------------------------------
    r = Recorder()
    for params, n, gs, ss in [
        ({"E": 42.0, "nu": 0.3}, 257, 1e-3, 0.0),  # cfg1 distribution, sigma_in = 0
        ({"E": 42.0, "nu": 0.3}, 257, 1e-3, 1.0),  # cfg2 distribution
        ({"E": 210e9, "nu": 0.25}, 64, 1e-4, 1e6),
        ({"E": 42.0, "nu": 0.3}, 1, 1e-3, 1.0),
        ({"E": 42.0, "nu": 0.0}, 63, 1.0, 1.0),
    ]:
        law = m.LinearElasticityModel(params, FULL)
        r.call("rand", law, params, 1.0, rng.normal(scale=gs, size=9 * n), rng.normal(scale=ss, size=6 * n) if ss else np.zeros(6 * n), None)
    params = {"E": 42.0, "nu": 0.3}
    law = m.LinearElasticityModel(params, FULL)
    r.call("zero_strain", law, params, 1.0, np.zeros(9 * 65), rng.normal(size=6 * 65), None)
    r.save("linear_elasticity.npz")

    # ---- a4: VonMises3D ---------------------------------------------------
    r = Recorder()
    params = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}
    law = m.VonMises3D(params)
    n = 193

    def zero_hist(n):
        return {"eps_n": np.zeros(6 * n), "alpha": np.zeros(n)}

    r.call("all_elastic", law, params, 1.0, rng.normal(scale=1e-4, size=9 * n), np.zeros(6 * n), zero_hist(n))
    r.call("all_plastic", law, params, 1.0, rng.normal(scale=1e-2, size=9 * n), np.zeros(6 * n), zero_hist(n))
    r.call("zero_strain", law, params, 1.0, np.zeros(9 * 67), np.zeros(6 * 67), zero_hist(67))
    # mixed multi-step with carried history; each step: a discarded Newton iterate, then the committed one
    scale = np.repeat(10 ** rng.uniform(-4, -2, size=n), 9)
    s, h = np.zeros(6 * n), zero_hist(n)
    h["alpha"] = rng.uniform(0, 0.02, size=n)
    for k in range(4):
        gk = rng.normal(size=9 * n) * scale
        r.call(f"mixed_step{k}_iter0", law, params, 1.0, 0.9 * gk, s, h)
        s, h = r.call(f"mixed_step{k}_iter1", law, params, 1.0, gk, s, h)
    # near-yield: uniaxial-like strain ramp through the yield point
    n2 = 129
    gk = np.zeros((n2, 9))
    gk[:, 0] = np.linspace(0.8, 1.2, n2) * (1200.0 * np.sqrt(2.0 / 3.0) / (2 * 80769.0)) * 1.5
    gk[:, 4] = -0.5 * gk[:, 0]
    gk[:, 8] = -0.5 * gk[:, 0]
    r.call("near_yield", law, params, 1.0, gk.reshape(-1), np.zeros(6 * n2), zero_hist(n2))
    # perfect plasticity (y00 = y0) and a softer hardening set
    for pp in (
        {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 1200.0, "p_w": 200.0},
        {"p_ka": 1.6e9, "p_mu": 1.2e9, "p_y0": 3.0e6, "p_y00": 9.0e6, "p_w": 35.0},
    ):
        lw = m.VonMises3D(pp)
        sc = 1e-2 if pp["p_mu"] < 1e6 else 3e-3
        r.call("other_params", lw, pp, 1.0, rng.normal(scale=sc, size=9 * 97), np.zeros(6 * 97), zero_hist(97))
    r.save("von_mises_3d.npz")

    # ---- a5/a6: SLS Maxwell / Kelvin -------------------------------------
    for cls, fname in ((m.SpringMaxwellModel, "spring_maxwell.npz"), (m.SpringKelvinModel, "spring_kelvin.npz")):
        r = Recorder()
        params = {"E0": 42.0, "E1": 10.0, "tau": 10.0, "nu": 0.2}
        law = cls(params, FULL)
        n = 201
        s = np.zeros(6 * n)
        h = {"strain_visco": np.zeros(6 * n), "strain": np.zeros(6 * n)}
        for k, dt in enumerate([1e-8, 2.0, 2.0, 5.0, 0.1]):
            gk = rng.normal(scale=1e-3, size=9 * n)
            r.call(f"step{k}_iter0", law, params, dt, 1.1 * gk, s, h)
            s, h = r.call(f"step{k}_iter1", law, params, dt, gk, s, h)
        p2 = {"E0": 3.0e10, "E1": 1.2e10, "tau": 0.7, "nu": 0.35}
        law2 = cls(p2, FULL)
        hh = {"strain_visco": rng.normal(scale=1e-4, size=6 * 70), "strain": rng.normal(scale=1e-3, size=6 * 70)}
        r.call("other_params", law2, p2, 0.05, rng.normal(scale=1e-4, size=9 * 70), rng.normal(scale=1e6, size=6 * 70), hh)
        r.call("zero_strain", law, params, 2.0, np.zeros(9 * 66), s[: 6 * 66], {k: v[: 6 * 66] for k, v in h.items()})
        r.save(fname)


def main_wrappers():
    """Golden vectors of the 3D->1D/2D wrappers (models/utils.py:211-412) around LE, VonMises3D and
    Maxwell; multi-call sequences on ONE wrapper instance so the cached 3-D arrays persist as in
    the reference."""
    m = import_reference()
    FULL = m.StressStrainConstraint.FULL
    rng = np.random.default_rng(7)
    d = {}
    idx = 0
    vm_p = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}
    sls_p = {"E0": 42.0, "E1": 10.0, "tau": 10.0, "nu": 0.2}
    le_p = {"E": 42.0, "nu": 0.3}
    cases = [
        ("plane_strain", "le", m.PlaneStrainFrom3D, lambda: m.LinearElasticityModel(le_p, FULL), None, 1e-3),
        ("uniaxial_strain", "le", m.UniaxialStrainFrom3D, lambda: m.LinearElasticityModel(le_p, FULL), None, 1e-3),
        ("plane_strain", "vm", m.PlaneStrainFrom3D, lambda: m.VonMises3D(vm_p), {"eps_n": 6, "alpha": 1}, 4e-3),
        ("uniaxial_strain", "vm", m.UniaxialStrainFrom3D, lambda: m.VonMises3D(vm_p), {"eps_n": 6, "alpha": 1}, 6e-3),
        ("plane_strain", "maxwell", m.PlaneStrainFrom3D, lambda: m.SpringMaxwellModel(sls_p, FULL), {"strain_visco": 6, "strain": 6}, 1e-3),
    ]
    for cname, lname, wcls, mk, hd, gscale in cases:
        w = wcls(mk())
        gd2, sd = w.geometric_dim**2, w.stress_strain_dim
        n = 150
        s = np.zeros(sd * n)
        h = None if hd is None else {k: np.zeros(dim * n) for k, dim in hd.items()}
        for step in range(3):
            g = rng.normal(scale=gscale, size=gd2 * n)
            s_in = s.copy()
            h_in = None if h is None else {k: v.copy() for k, v in h.items()}
            tan = np.full(sd * sd * n, np.nan)
            w.evaluate(0.0, 2.0, g, s, tan, h)
            p = f"c{idx}."
            d[p + "wrapper"], d[p + "law"], d[p + "step"] = np.array(cname), np.array(lname), np.int64(step)
            d[p + "grad"], d[p + "stress_in"], d[p + "stress_out"], d[p + "tangent_out"] = g, s_in, s.copy(), tan
            if h is not None:
                d[p + "hist_keys"] = np.array(list(h.keys()))
                for k in h:
                    d[p + "hist_in." + k], d[p + "hist_out." + k] = h_in[k], h[k].copy()
            idx += 1
    d["n_calls"] = np.int64(idx)
    np.savez_compressed(os.path.join(OUT, "wrappers.npz"), **d)
    print("wrappers.npz", idx, "calls")


def main_constraints():
    """LE / Maxwell / Kelvin under the four non-FULL constraints (multi-step, carried history)."""
    m = import_reference()
    C = m.StressStrainConstraint
    rng = np.random.default_rng(99)
    d, idx = {}, 0
    sls_p = {"E0": 42.0, "E1": 10.0, "tau": 10.0, "nu": 0.2}
    le_p = {"E": 42.0, "nu": 0.3}
    for c in (C.UNIAXIAL_STRAIN, C.UNIAXIAL_STRESS, C.PLANE_STRAIN, C.PLANE_STRESS):
        for lname, mk, hist in (("le", lambda: m.LinearElasticityModel(le_p, c), False),
                                ("maxwell", lambda: m.SpringMaxwellModel(sls_p, c), True),
                                ("kelvin", lambda: m.SpringKelvinModel(sls_p, c), True)):
            law = mk()
            gd2, sd = law.geometric_dim**2, law.stress_strain_dim
            n = 203
            s = rng.normal(size=sd * n)
            h = {"strain_visco": np.zeros(sd * n), "strain": np.zeros(sd * n)} if hist else None
            for step, dt in enumerate([1e-8, 2.0, 0.1]):
                g = rng.normal(scale=1e-3, size=gd2 * n)
                s_in, h_in = s.copy(), None if h is None else {k: v.copy() for k, v in h.items()}
                tan = np.full(sd * sd * n, np.nan)
                law.evaluate(0.0, dt, g, s, tan, h)
                p = f"c{idx}."
                d[p + "constraint"], d[p + "law"], d[p + "del_t"] = np.array(c.name), np.array(lname), np.float64(dt)
                d[p + "grad"], d[p + "stress_in"], d[p + "stress_out"], d[p + "tangent_out"] = g, s_in, s.copy(), tan
                if h is not None:
                    for k in h:
                        d[p + "hist_in." + k], d[p + "hist_out." + k] = h_in[k], h[k].copy()
                idx += 1
    d["n_calls"] = np.int64(idx)
    np.savez_compressed(os.path.join(OUT, "constraints.npz"), **d)
    print("constraints.npz", idx, "calls")


def main_material_point():
    """Load paths of the reference's own model classes under the material-point harness
    (tests/material_point.py): the 100-step uniaxial-stress test and the sine cycle of
    tests/models/test_plasticity.py, relaxation and creep of tests/models/test_viscoelasticity.py --
    whole multi-increment Newton histories of the *reference*, stored as curves."""
    m = import_reference()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests"))
    sys.path.insert(0, root)
    import material_point_cases as cases
    from material_point import HostState, MaterialPoints

    C = m.StressStrainConstraint

    class RefLaw:  # the reference class + the two attributes the harness reads
        def __init__(self, law):
            self.law, self.history_dim, self.constraint = law, law.history_dim, law.constraint

        def evaluate(self, *a):
            self.law.evaluate(*a)

    def build(kind, params, constraint, n):
        c = C[constraint]
        law = {"von_mises_3d": lambda: m.VonMises3D(params),
               "spring_maxwell": lambda: m.SpringMaxwellModel(params, c),
               "spring_kelvin": lambda: m.SpringKelvinModel(params, c),
               "linear_elasticity": lambda: m.LinearElasticityModel(params, c)}[kind]()
        return MaterialPoints(HostState(RefLaw(law), n), constraint, tol=1e-11)

    d = {}
    load, disp = cases.uniaxial_stress_3d(build, "von_mises_3d", n=8)
    d["uniaxial_stress_3d.load"], d["uniaxial_stress_3d.disp"] = load, disp
    load, disp = cases.uniaxial_cyclic_strain_3d(build, n=4)
    d["uniaxial_cyclic_strain_3d.load"], d["uniaxial_cyclic_strain_3d.disp"] = load, disp
    for kind in ("spring_kelvin", "spring_maxwell"):
        for constraint in ("UNIAXIAL_STRESS", "PLANE_STRESS", "FULL"):
            d[f"relaxation.{kind}.{constraint}"] = cases.relaxation(build, kind, constraint, n=5)
        for constraint in ("PLANE_STRESS", "FULL"):
            d[f"creep.{kind}.{constraint}"] = cases.creep(build, kind, constraint, n=5)
    cases.kelvin_vs_maxwell(build)
    for kind in ("spring_kelvin", "spring_maxwell"):
        cases.plane_strain_vs_3d(build, kind)
    cases.elasticity_constraints(build)
    np.savez_compressed(os.path.join(OUT, "material_point.npz"), **d)
    print("material_point.npz", len(d), "curves (every scenario's assertions also held for the reference classes)")


def main_random_parameters():
    """Random MATERIAL PARAMETERS (the other fixtures vary the inputs around two or three parameter sets): eight sets per
    law over many decades, random prestress / history, two increments with a discarded Newton iterate each -- pins the
    oracles' and the kernels' dependence on the parameters (host constants, del_t / tau ratios, hardening laws) to the
    imported reference itself.  Files random_parameters_<law>.npz, a generator of their own: the fixtures above are
    unchanged."""
    m = import_reference()
    FULL = m.StressStrainConstraint.FULL
    rng = np.random.default_rng(20261004)
    n, sets = 32, 8

    def logu(lo, hi):
        return float(10 ** rng.uniform(np.log10(lo), np.log10(hi)))

    # LinearElasticityModel
    r = Recorder()
    for k in range(sets):
        p = {"E": logu(1.0, 1e12), "nu": float(rng.uniform(-0.3, 0.49))}
        law = m.LinearElasticityModel(p, FULL)
        r.call(f"set{k}", law, p, 1.0, rng.normal(scale=logu(1e-6, 1e-1), size=9 * n), rng.normal(scale=1e-3 * p["E"], size=6 * n), None)
    r.save("random_parameters_linear_elasticity.npz")

    # VonMises3D: mixed elastic / plastic points around each set's own yield strain
    r = Recorder()
    k = 0
    while k < sets:
        mu = logu(1e3, 1e11)
        y0 = mu * logu(1e-4, 1e-2)
        p = {"p_ka": mu * float(rng.uniform(0.7, 5.0)), "p_mu": mu, "p_y0": y0, "p_y00": y0 * float(rng.uniform(1.0, 3.0)),
             "p_w": float(rng.uniform(1.0, 500.0))}
        law = m.VonMises3D(p)
        scale = np.repeat((y0 / mu) * 10 ** rng.uniform(-1.5, 1.0, size=n), 9)
        s, h = np.zeros(6 * n), {"eps_n": np.zeros(6 * n), "alpha": rng.uniform(0.0, 0.02, size=n)}
        mark = (len(r.calls), dict(r.d))
        try:
            for step in range(2):
                gk = rng.normal(size=9 * n) * scale
                r.call(f"set{k}_step{step}_iter0", law, p, 1.0, 0.85 * gk, s, h)
                s, h = r.call(f"set{k}_step{step}_iter1", law, p, 1.0, gk, s, h)
        except RuntimeError:  # the reference's Newton iteration gave up on this set: draw another one
            del r.calls[mark[0]:]
            r.d = mark[1]
            continue
        k += 1
    r.save("random_parameters_von_mises_3d.npz")

    # SLS Maxwell / Kelvin: del_t / tau over six decades
    for cls, fname in ((m.SpringMaxwellModel, "random_parameters_spring_maxwell.npz"), (m.SpringKelvinModel, "random_parameters_spring_kelvin.npz")):
        r = Recorder()
        for k in range(sets):
            E0 = logu(1.0, 1e11)
            p = {"E0": E0, "E1": E0 * float(rng.uniform(0.05, 2.0)), "tau": logu(1e-3, 1e3), "nu": float(rng.uniform(0.0, 0.45))}
            law = cls(p, FULL)
            eps = logu(1e-5, 1e-2)
            s = rng.normal(scale=eps * E0, size=6 * n)
            h = {"strain_visco": rng.normal(scale=0.3 * eps, size=6 * n), "strain": rng.normal(scale=eps, size=6 * n)}
            for step in range(2):
                dt = p["tau"] * logu(1e-3, 1e3)
                gk = rng.normal(scale=eps, size=9 * n)
                r.call(f"set{k}_step{step}_iter0", law, p, dt, 1.2 * gk, s, h)
                s, h = r.call(f"set{k}_step{step}_iter1", law, p, dt, gk, s, h)
        r.save(fname)

    # the same three laws under the four non-FULL constraints, four random parameter sets each (format of constraints.npz
    # plus the parameters of every call)
    C = m.StressStrainConstraint
    d, idx = {}, 0
    for c in (C.UNIAXIAL_STRAIN, C.UNIAXIAL_STRESS, C.PLANE_STRAIN, C.PLANE_STRESS):
        for lname in ("le", "maxwell", "kelvin"):
            for k in range(4):
                if lname == "le":
                    p = {"E": logu(1.0, 1e12), "nu": float(rng.uniform(-0.3, 0.49))}
                    law, E = m.LinearElasticityModel(p, c), p["E"]
                else:
                    E = logu(1.0, 1e11)
                    p = {"E0": E, "E1": E * float(rng.uniform(0.05, 2.0)), "tau": logu(1e-3, 1e3), "nu": float(rng.uniform(0.0, 0.45))}
                    law = (m.SpringMaxwellModel if lname == "maxwell" else m.SpringKelvinModel)(p, c)
                gd2, sd = law.geometric_dim**2, law.stress_strain_dim
                eps = logu(1e-5, 1e-2)
                s = rng.normal(scale=eps * E, size=sd * n)
                h = None if lname == "le" else {"strain_visco": rng.normal(scale=0.3 * eps, size=sd * n), "strain": rng.normal(scale=eps, size=sd * n)}
                for step in range(2):
                    dt = 1.0 if lname == "le" else p["tau"] * logu(1e-3, 1e3)
                    g = rng.normal(scale=eps, size=gd2 * n)
                    s_in, h_in = s.copy(), None if h is None else {kk: v.copy() for kk, v in h.items()}
                    tan = np.full(sd * sd * n, np.nan)
                    law.evaluate(0.0, dt, g, s, tan, h)
                    q = f"c{idx}."
                    d[q + "constraint"], d[q + "law"], d[q + "del_t"] = np.array(c.name), np.array(lname), np.float64(dt)
                    d[q + "param_keys"], d[q + "param_vals"] = np.array(list(p)), np.array([float(v) for v in p.values()])
                    d[q + "grad"], d[q + "stress_in"], d[q + "stress_out"], d[q + "tangent_out"] = g, s_in, s.copy(), tan
                    if h is not None:
                        for kk in h:
                            d[q + "hist_in." + kk], d[q + "hist_out." + kk] = h_in[kk], h[kk].copy()
                    idx += 1
    d["n_calls"] = np.int64(idx)
    np.savez_compressed(os.path.join(OUT, "random_parameters_constraints.npz"), **d)
    print("random_parameters_constraints.npz", idx, "calls")


def main_mises_limit():
    """A pin from the imported reference for the PLASTIC branch of comfe-rs MisesPlasticity3D (which has no test of its own
    in the reference): the Python VonMises3D with saturation hardening sigma_y = y0 + (y00 - y0)(1 - exp(-w alpha))
    (mises_plasticity_isotropic_hardening.py:92-95) tends to LINEAR hardening sigma_y = y0 + h alpha, h = (y00 - y0) w, for
    w alpha -> 0 -- the hardening law of the Rust class (mises_plasticity.rs:98-107).  With w = 1e-6 the two yield stresses
    differ by less than 1e-8 h alpha (truncation w alpha / 2 and the cancellation in 1 - exp), so stress and alpha of the Rust
    law must reproduce these outputs to ~1e-9; its plastic strain is sqrt(2/3) times the Python one and its tangent differs
    by a rank-one term (the non-unit flow direction and the sign of the Rust text) -- tests/test_oracle_golden.py states
    both relations.  File mises_linear_hardening_limit.npz."""
    m = import_reference()
    rng = np.random.default_rng(777)
    r = Recorder()
    n = 96
    for k, (mu, ka, y0, h) in enumerate([(80769.0, 175000.0, 1200.0, 200.0), (80769.0, 175000.0, 1200.0, 20000.0), (2.6e10, 5.5e10, 2.4e8, 1.5e9)]):
        w = 1e-6
        p = {"p_ka": ka, "p_mu": mu, "p_y0": y0, "p_y00": y0 + h / w, "p_w": w}
        law = m.VonMises3D(p)
        scale = np.repeat((y0 / mu) * 10 ** rng.uniform(-1.0, 1.0, size=n), 9)
        s, hist = np.zeros(6 * n), {"eps_n": np.zeros(6 * n), "alpha": rng.uniform(0.0, 0.02, size=n)}
        for step in range(3):
            gk = rng.normal(size=9 * n) * scale
            s, hist = r.call(f"set{k}_step{step}", law, p, 1.0, gk, s, hist)
    r.save("mises_linear_hardening_limit.npz")

    # ... and for the general return mapping (plasticity/general.rs:105-266) with the Drucker-Prager surfaces: for b = b_flow = 0
    # they are J2 cylinders without hardening (drucker_prager_classic.rs:88: f = sqrt(J2) - a; hyperbolic.rs:87:
    # f = sqrt(J2 + d^2) - a) with a flow direction along s, i.e. the closest-point projection IS the radial return of the
    # Python VonMises3D with y00 = y0 = sqrt(3) a (hyperbolic: sqrt(3 (a^2 - d^2))).  Three increments, mixed points.
    r = Recorder()
    for k, (mu, ka, y0) in enumerate([(80769.0, 175000.0, 1200.0), (2.6e10, 5.5e10, 2.4e8)]):
        p = {"p_ka": ka, "p_mu": mu, "p_y0": y0, "p_y00": y0, "p_w": 200.0}
        law = m.VonMises3D(p)
        scale = np.repeat((y0 / mu) * 10 ** rng.uniform(-1.0, 0.7, size=n), 9)
        s, hist = np.zeros(6 * n), {"eps_n": np.zeros(6 * n), "alpha": np.zeros(n)}
        for step in range(3):
            gk = rng.normal(size=9 * n) * scale
            s, hist = r.call(f"set{k}_step{step}", law, p, 1.0, gk, s, hist)
    r.save("von_mises_perfect_plasticity.npz")

    # ... and the PRESSURE DEPENDENCE of the yield functions: with b != 0 but b_flow = 0 (non-associated, purely deviatoric flow,
    # drucker_prager_classic.rs:96-103) the return leaves the pressure at its trial value, so every point returns radially onto a
    # J2 cylinder of ITS OWN radius sqrt(J2) = a - b I1_trial (hyperbolic: sqrt((a - b I1_trial)^2 - d^2)).  The Python
    # VonMises3D, called point by point with y00 = y0 = sqrt(3) x that radius, returns the same stress and plastic strain; its
    # tangent lacks only the derivative of the radius with respect to the strain (tests/golden_util.py: dp_pressure_cases).
    d, idx = {}, 0
    mu, ka = 80769.0, 175000.0
    sid = np.array([1.0, 1.0, 1.0, 0.0, 0.0, 0.0])
    for name, a, b, dd in (("classic", 100.0, 0.05, 0.0), ("classic_steep", 40.0, 0.2, 0.0), ("hyperbolic", 100.0, 0.05, 40.0)):
        nn = 64
        s = rng.normal(scale=30.0, size=(nn, 6))
        s[:, :3] -= rng.uniform(300.0, 1500.0, size=nn)[:, None]
        eps_p = np.zeros((nn, 6))
        for step in range(2):
            g = rng.normal(size=(nn, 9)) * (10 ** rng.uniform(-4.0, -2.3, size=nn))[:, None]
            g[:, [0, 4, 8]] -= (0.9 * g[:, [0, 4, 8]].sum(axis=1) / 3.0)[:, None]  # mostly isochoric: the trial pressure stays below the tip
            strain = m.strain_from_grad_u(g.reshape(-1), m.StressStrainConstraint.FULL).reshape(nn, 6)
            i1_tr = s[:, :3].sum(axis=1) + 3.0 * ka * strain[:, :3].sum(axis=1)
            big = a - b * i1_tr
            radius = np.sqrt(big * big - dd * dd)
            s_out, t_out, e_out = np.empty((nn, 6)), np.empty((nn, 36)), np.empty((nn, 6))
            for i in range(nn):  # one reference call per point: its own yield stress
                y = float(np.sqrt(3.0) * radius[i])
                law = m.VonMises3D({"p_ka": ka, "p_mu": mu, "p_y0": y, "p_y00": y, "p_w": 1.0})
                si, ti = s[i].copy(), np.full(36, np.nan)
                hi = {"eps_n": eps_p[i].copy(), "alpha": np.zeros(1)}
                law.evaluate(0.0, 1.0, g[i].copy(), si, ti, hi)
                s_out[i], t_out[i], e_out[i] = si, ti, hi["eps_n"]
            q = f"c{idx}."
            d[q + "name"] = np.array(f"{name}_step{step}")
            d[q + "param_keys"] = np.array(["mu", "kappa", "a", "b", "d", "b_flow"])
            d[q + "param_vals"] = np.array([mu, ka, a, b, dd, 0.0])
            d[q + "grad"], d[q + "stress_in"], d[q + "eps_p_in"] = g.reshape(-1), s.reshape(-1).copy(), eps_p.reshape(-1).copy()
            d[q + "stress_out"], d[q + "tangent_py"], d[q + "eps_p_out"] = s_out.reshape(-1), t_out.reshape(-1), e_out.reshape(-1)
            d[q + "radius"], d[q + "a_minus_b_i1"] = radius, big
            s, eps_p = s_out, e_out
            idx += 1
    d["n_calls"] = np.int64(idx)
    np.savez_compressed(os.path.join(OUT, "drucker_prager_deviatoric_flow.npz"), **d)
    print("drucker_prager_deviatoric_flow.npz", idx, "calls")

    # ... and b_flow != 0 (classic surface; associated flow b_flow = b included): the deviatoric part of the return is still
    # radial -- onto sqrt(J2) = R = sqrt(J2_trial) - mu del_lambda --, the flow's volumetric part b_flow I2 moves the pressure.
    # The radius handed to the reference (point by point, as above) is the one at which the RETURNED state -- the reference's
    # deviatoric stress, the pressure moved by what the Rust flow rule says for the reference's own plastic multiplier --
    # satisfies the Rust yield function; tests/golden_util.py: dp_volumetric_cases checks that and states the relations.
    d, idx = {}, 0
    for name, a, b, bf in (("nonassociated", 100.0, 0.05, 0.02), ("associated", 100.0, 0.05, 0.05), ("steep", 40.0, 0.2, 0.1)):
        nn = 64
        s = rng.normal(scale=30.0, size=(nn, 6))
        s[:, :3] -= rng.uniform(300.0, 1500.0, size=nn)[:, None]
        eps_p = np.zeros((nn, 6))
        for step in range(2):
            g = rng.normal(size=(nn, 9)) * (10 ** rng.uniform(-4.0, -2.3, size=nn))[:, None]
            g[:, [0, 4, 8]] -= (0.9 * g[:, [0, 4, 8]].sum(axis=1) / 3.0)[:, None]
            strain = m.strain_from_grad_u(g.reshape(-1), m.StressStrainConstraint.FULL).reshape(nn, 6)
            i1_tr = s[:, :3].sum(axis=1) + 3.0 * ka * strain[:, :3].sum(axis=1)
            dev = s.copy()
            dev[:, :3] -= s[:, :3].mean(axis=1, keepdims=True)
            e_dev = strain.copy()
            e_dev[:, :3] -= strain[:, :3].mean(axis=1, keepdims=True)
            rj2_tr = np.sqrt(0.5 * ((dev + 2.0 * mu * e_dev) ** 2).sum(axis=1))
            f_tr = rj2_tr + b * i1_tr - a
            radius = np.where(f_tr > 0.0, rj2_tr - mu * f_tr / (mu + 9.0 * ka * b * bf), a - b * i1_tr)
            s_out, t_out, e_out, al = np.empty((nn, 6)), np.empty((nn, 36)), np.empty((nn, 6)), np.empty(nn)
            for i in range(nn):
                y = float(np.sqrt(3.0) * radius[i])
                law = m.VonMises3D({"p_ka": ka, "p_mu": mu, "p_y0": y, "p_y00": y, "p_w": 1.0})
                si, ti = s[i].copy(), np.full(36, np.nan)
                hi = {"eps_n": np.zeros(6), "alpha": np.zeros(1)}
                law.evaluate(0.0, 1.0, g[i].copy(), si, ti, hi)
                s_out[i], t_out[i], e_out[i], al[i] = si, ti, hi["eps_n"], hi["alpha"][0]
            q = f"c{idx}."
            d[q + "name"] = np.array(f"{name}_step{step}")
            d[q + "param_keys"] = np.array(["mu", "kappa", "a", "b", "b_flow"])
            d[q + "param_vals"] = np.array([mu, ka, a, b, bf])
            d[q + "grad"], d[q + "stress_in"], d[q + "eps_p_in"] = g.reshape(-1), s.reshape(-1).copy(), eps_p.reshape(-1).copy()
            d[q + "stress_py"], d[q + "tangent_py"], d[q + "deps_py"], d[q + "alpha_py"] = s_out.reshape(-1), t_out.reshape(-1), e_out.reshape(-1), al
            # carried state for the next increment: what the Rust law returns according to the relations of dp_volumetric_cases
            dl = np.sqrt(3.0) * al
            s = s_out - (3.0 * ka * bf * dl)[:, None] * sid[None, :]
            eps_p = eps_p + e_out + (bf * dl)[:, None] * sid[None, :]
            idx += 1
    # the hyperbolic surface with b_flow != 0, same idea: s_1 (1 + mu del_lambda / Q) = s_trial with Q = sqrt(J2_1 + d^2) =
    # a - b (I1_trial - 9 kappa b_flow del_lambda) is one scalar equation for del_lambda, solved here by bisection ONLY to choose the
    # radius handed to the reference; the checks of dp_volumetric_cases are on the returned state
    for name, a, b, bf, dd in (("hyperbolic_nonassociated", 100.0, 0.05, 0.02, 40.0), ("hyperbolic_associated", 100.0, 0.05, 0.05, 40.0)):
        nn = 64
        s = rng.normal(scale=30.0, size=(nn, 6))
        s[:, :3] -= rng.uniform(300.0, 1500.0, size=nn)[:, None]
        eps_p = np.zeros((nn, 6))
        for step in range(2):
            g = rng.normal(size=(nn, 9)) * (10 ** rng.uniform(-4.0, -2.3, size=nn))[:, None]
            g[:, [0, 4, 8]] -= (0.9 * g[:, [0, 4, 8]].sum(axis=1) / 3.0)[:, None]
            strain = m.strain_from_grad_u(g.reshape(-1), m.StressStrainConstraint.FULL).reshape(nn, 6)
            i1_tr = s[:, :3].sum(axis=1) + 3.0 * ka * strain[:, :3].sum(axis=1)
            dev = s.copy()
            dev[:, :3] -= s[:, :3].mean(axis=1, keepdims=True)
            e_dev = strain.copy()
            e_dev[:, :3] -= strain[:, :3].mean(axis=1, keepdims=True)
            rj2_tr = np.sqrt(0.5 * ((dev + 2.0 * mu * e_dev) ** 2).sum(axis=1))
            big = a - b * i1_tr
            f_tr = np.sqrt(rj2_tr**2 + dd * dd) - big
            radius = np.sqrt(big * big - dd * dd)  # elastic points: any radius they stay inside of
            for i in np.flatnonzero(f_tr > 0.0):
                def h(dl):
                    q_ = big[i] + 9.0 * ka * b * bf * dl
                    return np.sqrt(max(q_ * q_ - dd * dd, 0.0)) * (1.0 + mu * dl / q_) - rj2_tr[i]
                lo, hi = 0.0, rj2_tr[i] / mu
                for _ in range(200):
                    mid = 0.5 * (lo + hi)
                    lo, hi = (mid, hi) if h(mid) < 0.0 else (lo, mid)
                q_ = big[i] + 9.0 * ka * b * bf * 0.5 * (lo + hi)
                radius[i] = np.sqrt(q_ * q_ - dd * dd)
            s_out, t_out, e_out, al = np.empty((nn, 6)), np.empty((nn, 36)), np.empty((nn, 6)), np.empty(nn)
            for i in range(nn):
                y = float(np.sqrt(3.0) * radius[i])
                law = m.VonMises3D({"p_ka": ka, "p_mu": mu, "p_y0": y, "p_y00": y, "p_w": 1.0})
                si, ti = s[i].copy(), np.full(36, np.nan)
                hi_ = {"eps_n": np.zeros(6), "alpha": np.zeros(1)}
                law.evaluate(0.0, 1.0, g[i].copy(), si, ti, hi_)
                s_out[i], t_out[i], e_out[i], al[i] = si, ti, hi_["eps_n"], hi_["alpha"][0]
            q = f"c{idx}."
            d[q + "name"] = np.array(f"{name}_step{step}")
            d[q + "param_keys"] = np.array(["mu", "kappa", "a", "b", "d", "b_flow"])
            d[q + "param_vals"] = np.array([mu, ka, a, b, dd, bf])
            d[q + "grad"], d[q + "stress_in"], d[q + "eps_p_in"] = g.reshape(-1), s.reshape(-1).copy(), eps_p.reshape(-1).copy()
            d[q + "stress_py"], d[q + "tangent_py"], d[q + "deps_py"], d[q + "alpha_py"] = s_out.reshape(-1), t_out.reshape(-1), e_out.reshape(-1), al
            dev1 = s_out.copy()
            dev1[:, :3] -= s_out[:, :3].mean(axis=1, keepdims=True)
            j2_1 = 0.5 * (dev1**2).sum(axis=1)
            dl = np.sqrt(3.0) * al * np.sqrt(j2_1 + dd * dd) / np.sqrt(np.where(j2_1 > 0.0, j2_1, 1.0))
            s = s_out - (3.0 * ka * bf * dl)[:, None] * sid[None, :]
            eps_p = eps_p + e_out + (bf * dl)[:, None] * sid[None, :]
            idx += 1
    d["n_calls"] = np.int64(idx)
    np.savez_compressed(os.path.join(OUT, "drucker_prager_volumetric_flow.npz"), **d)
    print("drucker_prager_volumetric_flow.npz", idx, "calls")


if __name__ == "__main__":
    if "--mises-limit" in sys.argv:
        main_mises_limit()
    elif "--random-parameters" in sys.argv:
        main_random_parameters()
    elif "--material-point" in sys.argv:
        main_material_point()
    elif "--constraints" in sys.argv:
        main_constraints()
    elif "--wrappers" in sys.argv:
        main_wrappers()
    else:
        main()
        main_wrappers()
        main_constraints()
        main_material_point()
        main_random_parameters()
        main_mises_limit()
