"""Parameter-space sweep: random material parameters (not just the reference's test values) for every
law -- the host-side constants of libfcamd (fill_constants) and the kernels against the oracle.
Includes the edges the formulas are sensitive to: nu -> 0 and nu -> 0.5, stiff/soft moduli over ten
decades, hardening nearly flat / very steep, tau from 1e-3 to 1e3 times del_t."""

import numpy as np
import pytest
from golden_util import rel_err
from test_gpu_parity import CLASS, TOL, make_law, oracle_run, run_device, run_host

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def params_for(kind, rng):
    scale = 10 ** rng.uniform(-3, 7)  # modulus scale
    nu = rng.choice([0.0, 1e-6, 0.499, rng.uniform(0.05, 0.45)])
    if kind == "linear_elasticity":
        return {"E": scale, "nu": nu}, scale
    if kind in ("spring_maxwell", "spring_kelvin"):
        return {"E0": scale, "E1": scale * 10 ** rng.uniform(-2, 2), "tau": 10 ** rng.uniform(-3, 3), "nu": nu}, scale
    mu = scale
    kappa = scale * 10 ** rng.uniform(-0.5, 2)
    y0 = mu * 10 ** rng.uniform(-4, -2)
    if kind == "comfe_linear_elasticity":
        return {"mu": mu, "kappa": kappa}, scale
    if kind == "von_mises_3d":
        return {"p_ka": kappa, "p_mu": mu, "p_y0": y0, "p_y00": y0 * rng.uniform(1.0, 3.0), "p_w": 10 ** rng.uniform(0, 4)}, scale
    return {"mu": mu, "kappa": kappa, "y_0": y0, "h": mu * 10 ** rng.uniform(-6, 1)}, scale


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("kind", list(CLASS))
def test_random_parameters(kind, seed):
    rng = np.random.default_rng(1000 * seed + len(kind))
    p, scale = params_for(kind, rng)
    n = 64 * 9 + 17
    # strains around the yield strain of the plasticity laws, 1e-3 otherwise
    eps = 10 ** rng.uniform(-5, -1, size=n) if CLASS[kind] == "pl" else np.full(n, 1e-3)
    g = rng.normal(size=9 * n) * np.repeat(eps, 9)
    s = rng.normal(scale=1e-3 * scale, size=6 * n)
    h = None
    if kind == "von_mises_3d":
        h = {"eps_n": rng.normal(scale=1e-3, size=6 * n), "alpha": rng.uniform(0, 0.02, size=n)}
    elif kind in ("spring_maxwell", "spring_kelvin"):
        h = {"strain_visco": rng.normal(scale=1e-4, size=6 * n), "strain": rng.normal(scale=1e-3, size=6 * n)}
    elif kind == "comfe_mises_plasticity":
        hh = rng.normal(scale=1e-3, size=7 * n)
        hh.reshape(-1, 7)[:, 0] = rng.uniform(0, 0.02, size=n)
        h = {"history": hh}
    del_t = 10 ** rng.uniform(-2, 2)
    ref = oracle_run(kind, p, del_t, g, s, h)
    law = make_law(kind, p)
    tol = TOL[CLASS[kind]]
    for runner in (run_host, run_device):
        hc = None if h is None else {k: v.copy() for k, v in h.items()}
        got = runner(law, del_t, g, s.copy(), np.full(36 * n, np.nan), hc)
        assert rel_err(got[0], ref[0]) <= tol, (kind, p, "stress", rel_err(got[0], ref[0]))
        assert rel_err(got[1], ref[1]) <= tol, (kind, p, "tangent", rel_err(got[1], ref[1]))
        if h is not None:
            for k in h:
                assert rel_err(got[2][k], ref[2][k]) <= tol, (kind, p, k)
