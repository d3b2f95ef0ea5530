"""GPU parity of the device-backed 3D->1D/2D wrappers and the component-map kernels
(fcamd_convert_device) against golden vectors captured from the reference's
UniaxialStrainFrom3D / PlaneStrainFrom3D (models/utils.py:211-412)."""

import numpy as np
import pytest
from golden_util import rel_err
from wrappers_util import PARAMS, load_sequences

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import fenics_constitutive_amd as fc  # noqa: E402

FULL = fc.StressStrainConstraint.FULL
TOL = {"le": 1e-10, "maxwell": 1e-10, "vm": 1e-6}


def make(kind, lname):
    law = {"le": lambda: fc.LinearElasticityModel(PARAMS["le"], FULL), "vm": lambda: fc.VonMises3D(PARAMS["vm"]),
           "maxwell": lambda: fc.SpringMaxwellModel(PARAMS["maxwell"], FULL)}[lname]()
    return (fc.PlaneStrainFrom3D if kind == "plane_strain" else fc.UniaxialStrainFrom3D)(law)


SEQS = load_sequences()


@pytest.mark.parametrize("path", ["host", "device"])
@pytest.mark.parametrize("kind,lname,calls", SEQS, ids=[f"{k}-{name}" for k, name, _ in SEQS])
def test_wrapper_sequences(kind, lname, calls, path):
    w = make(kind, lname)
    assert w.constraint.name == kind.upper() and w.history_dim == w.model.history_dim
    for c in calls:
        s = c["stress_in"].copy()
        t = np.full_like(c["tangent_out"], np.nan)
        h = None if c["hist_in"] is None else {k: v.copy() for k, v in c["hist_in"].items()}
        if path == "host":
            w.evaluate(0.0, 2.0, c["grad"], s, t, h)
        else:
            sd, td = torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()
            hd = None if h is None else {k: torch.from_numpy(v).cuda() for k, v in h.items()}
            w.evaluate(0.0, 2.0, torch.from_numpy(c["grad"]).cuda(), sd, td, hd)
            s, t = sd.cpu().numpy(), td.cpu().numpy()
            h = None if hd is None else {k: v.cpu().numpy() for k, v in hd.items()}
        assert rel_err(s, c["stress_out"]) <= TOL[lname], (kind, lname, path)
        assert rel_err(t, c["tangent_out"]) <= TOL[lname]
        if h is not None:
            for k in h:
                assert rel_err(h[k], c["hist_out"][k]) <= TOL[lname], k


def test_plane_strain_equals_constrained_3d():
    """tests/models/test_viscoelasticity.py:664-694 in array form: plane strain == 3-D with the
    out-of-plane gradient components zero."""
    rng = np.random.default_rng(2)
    n = 500
    g2 = rng.normal(scale=1e-3, size=4 * n)
    g3 = np.zeros((n, 9))
    g3[:, [0, 1, 3, 4]] = g2.reshape(-1, 4)
    law3 = fc.LinearElasticityModel(PARAMS["le"], FULL)
    s3, t3 = np.zeros(6 * n), np.zeros(36 * n)
    law3.evaluate(0, 1, g3.reshape(-1), s3, t3, None)
    s2, t2 = np.zeros(4 * n), np.zeros(16 * n)
    fc.PlaneStrainFrom3D(fc.LinearElasticityModel(PARAMS["le"], FULL)).evaluate(0, 1, g2, s2, t2, None)
    assert np.array_equal(s2.reshape(-1, 4), s3.reshape(-1, 6)[:, :4])
    assert np.array_equal(t2.reshape(-1, 4, 4), t3.reshape(-1, 6, 6)[:, :4, :4])


def _mises_law(lname):
    if lname == "le":
        return fc.LinearElasticityModel(PARAMS["le"], FULL)
    if lname == "vm":
        return fc.VonMises3D(PARAMS["vm"])
    if lname == "comfe_mises":
        return fc.MisesPlasticityLinearHardening3D(
            {k: np.array([v]) for k, v in {"mu": 80769.0, "kappa": 175000.0, "y_0": 1200.0, "h": 200.0}.items()})
    p = {"mu": 80769.0, "kappa": 175000.0, "a": 100.0, "b": 0.05, "b_flow": 0.02}
    if lname == "dp_hyper":
        p = {"mu": p["mu"], "kappa": p["kappa"], "a": p["a"], "b": p["b"], "d": 40.0, "b_flow": p["b_flow"]}
        return fc.DruckerPragerHyperbolic3D({k: np.array([v]) for k, v in p.items()})
    return fc.DruckerPrager3D({k: np.array([v]) for k, v in p.items()})


@pytest.mark.parametrize("lname", ["le", "vm", "comfe_mises", "dp", "dp_hyper"])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 4097])
@pytest.mark.parametrize("kind", ["plane_strain", "uniaxial_strain"])
def test_fused_wrapper_equals_map_evaluate_map(kind, n, lname):
    """The plasticity laws: the fused kernel (fcamd_evaluate_device_wrapped) and the generic map -> 3-D evaluate ->
    map sequence give bit-identical stress, tangent, history and cached 3-D stress over several calls with
    growing plastic sets (the cached lateral stresses of the uniaxial wrapper carry over between calls)."""
    rng = np.random.default_rng(n)
    W = fc.PlaneStrainFrom3D if kind == "plane_strain" else fc.UniaxialStrainFrom3D
    a, b = W(_mises_law(lname)), W(_mises_law(lname))
    b.fused = False
    gd2, sd = a.geometric_dim**2, a.stress_strain_dim
    d = lambda x: torch.from_numpy(x.copy()).cuda()  # noqa: E731
    s0 = rng.normal(scale=30.0, size=sd * n)
    dp = lname.startswith("dp")
    if dp:  # compressive prestress: the regime where the reference's Newton iteration converges
        s0.reshape(n, sd)[:, : min(sd, 3)] -= 1000.0 if sd == 4 else 100.0  # (uniaxial: the lateral stresses start at 0)
    if lname == "le":
        h0 = {}
    elif lname == "vm":
        h0 = {"eps_n": rng.normal(scale=1e-3, size=6 * n), "alpha": rng.uniform(0, 0.02, size=n)}
    else:
        hh = rng.normal(scale=1e-3, size=7 * n)
        hh.reshape(-1, 7)[:, 0] = rng.uniform(0, 0.02, size=n)
        h0 = {"history": hh}
    sa, sb = d(s0), d(s0)
    ha, hb = {k: d(v) for k, v in h0.items()}, {k: d(v) for k, v in h0.items()}
    ta, tb = torch.zeros(sd * sd * n, dtype=torch.float64, device="cuda"), torch.zeros(sd * sd * n, dtype=torch.float64, device="cuda")
    for call in range(4):
        hi = ((-2.9 if gd2 == 4 else -3.6) if dp else -2.0) + 0.1 * call
        g = rng.normal(size=gd2 * n) * np.repeat(10 ** rng.uniform(-4, hi, size=n), gd2)
        if dp and gd2 == 4:  # mostly isochoric in-plane increments keep the classic surface off its tip
            gv = g.reshape(n, 4)
            tr = gv[:, 0] + gv[:, 3]
            gv[:, 0] -= 0.475 * tr
            gv[:, 3] -= 0.475 * tr
        a.evaluate(0.0, 1.0, d(g), sa, ta, ha or None)
        b.evaluate(0.0, 1.0, d(g), sb, tb, hb or None)
        assert torch.equal(sa, sb) and torch.equal(ta, tb), (kind, n, call)
        assert torch.equal(a.stress_3d, b.stress_3d)
        for k in ha:
            assert torch.equal(ha[k], hb[k]), k
    assert a.grad_del_u_3d is None and a.tangent_3d is None and b.tangent_3d is not None
    assert lname == "le" or a.model.device_stats().n_plastic > 0 or n < 10


@pytest.mark.parametrize("kind", ["plane_strain", "uniaxial_strain"])
def test_wrapped_3d_law_equals_native_constraint(kind):
    """tests/models/test_elasticity.py:157-298 in array form: LinearElasticityModel(FULL) behind the
    wrapper gives the stresses of LinearElasticityModel(UNIAXIAL_STRAIN / PLANE_STRAIN), the analytical
    uniaxial-strain stress, a non-zero out-of-plane stress under plane strain, and no shear in the
    cached 3-D stress."""
    C = fc.StressStrainConstraint
    E, nu = PARAMS["le"]["E"], PARAMS["le"]["nu"]
    n = 64 * 3 + 11
    rng = np.random.default_rng(4)
    if kind == "uniaxial_strain":
        native, W, gd2, sd = fc.LinearElasticityModel(PARAMS["le"], C.UNIAXIAL_STRAIN), fc.UniaxialStrainFrom3D, 1, 1
        g = np.full(n, 0.01) * np.linspace(0.5, 1.0, n)
    else:
        native, W, gd2, sd = fc.LinearElasticityModel(PARAMS["le"], C.PLANE_STRAIN), fc.PlaneStrainFrom3D, 4, 4
        g = np.zeros((n, 4))
        g[:, 0] = 0.01 * np.linspace(0.5, 1.0, n)
        g[:, 3] = rng.normal(scale=1e-3, size=n)
        g = g.reshape(-1)
    wrapped = W(fc.LinearElasticityModel(PARAMS["le"], FULL))
    s_n, t_n = np.zeros(sd * n), np.zeros(sd * sd * n)
    s_w, t_w = np.zeros(sd * n), np.zeros(sd * sd * n)
    native.evaluate(0.0, 1.0, g, s_n, t_n, None)
    wrapped.evaluate(0.0, 1.0, g, s_w, t_w, None)
    assert rel_err(s_w, s_n) < 1e-10 and rel_err(t_w, t_n) < 1e-10
    if kind == "uniaxial_strain":
        analytical = E * (1.0 - nu) / ((1.0 + nu) * (1.0 - 2.0 * nu)) * g
        assert np.max(np.abs(s_w - analytical)) < 1e-10 / analytical.max()
    else:
        assert np.abs(s_w.reshape(-1, 4)[:, 2]).min() > 1e-3  # sigma_33 is not zero under plane strain
    s3 = wrapped.stress_3d.cpu().numpy().reshape(-1, 6)
    assert np.abs(s3[:, 3 if kind == "uniaxial_strain" else 4:]).max() < 1e-14
    assert np.abs(s3[:, 1:3]).min() > 0  # lateral stresses live in the cached 3-D stress
