"""The HIP kernels against the self-derived 50-digit vectors of the comfe-rs plasticity updates (oracle/mp_pins.py,
tests/golden/comfe_selfderived_*.npz): the laws the reference holds no known answer for.  Contract tolerance 1e-6
(BASELINE.json north_star, plasticity); the regression bounds are what the kernels measure on MI355X with a margin."""

import os

import numpy as np
import pytest
from golden_util import GOLDEN

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import fenics_constitutive_amd as fc  # noqa: E402

# measured on MI355X (round 3): stress / tangent / alpha <= 4e-16, eps_p <= 4e-15 per point for all three -- the kernels'
# invariant-coordinate Newton iteration reproduces the 50-digit result of the full 8 x 8 system to rounding level
CASES = [("comfe_selfderived_mises.npz", fc.MisesPlasticityLinearHardening3D, 1e-14, 1e-14),
         ("comfe_selfderived_drucker_prager_classic.npz", fc.DruckerPrager3D, 1e-14, 1e-14),
         ("comfe_selfderived_drucker_prager_hyperbolic.npz", fc.DruckerPragerHyperbolic3D, 1e-14, 1e-14)]


def worst_point(a, b, d):
    a, b = a.reshape(-1, d), b.reshape(-1, d)
    return float((np.abs(a - b).max(axis=1) / np.abs(b).max(axis=1)).max())


@pytest.mark.parametrize("fname,cls,tol_s,tol_t", CASES)
def test_hip_kernels_against_the_50_digit_transcription(fname, cls, tol_s, tol_t):
    z = np.load(os.path.join(GOLDEN, fname))
    law = cls({str(k): np.array([float(v)]) for k, v in zip(z["param_keys"], z["param_vals"])})
    s, t, h = z["stress_in"].copy(), np.full(z["tangent_out"].size, np.nan), {"history": z["hist_in"].copy()}
    law.evaluate(0.0, 1.0, z["grad"].copy(), s, t, h)
    hv, ho = h["history"].reshape(-1, 7), z["hist_out"].reshape(-1, 7)
    assert np.array_equal(hv[:, 0] != z["hist_in"].reshape(-1, 7)[:, 0], z["plastic"])  # the same points yield
    errs = {"stress": worst_point(s, z["stress_out"], 6), "tangent": worst_point(t, z["tangent_out"], 36),
            "alpha": float(np.abs(hv[:, 0] - ho[:, 0]).max() / np.abs(ho[:, 0]).max()),
            "eps_p": worst_point(hv[:, 1:].copy(), ho[:, 1:].copy(), 6)}
    print(fname, errs)
    assert max(errs.values()) <= 1e-6, errs  # the contract
    assert errs["stress"] <= tol_s and errs["alpha"] <= tol_s and errs["eps_p"] <= 10 * tol_s and errs["tangent"] <= tol_t, errs


from golden_util import check_mises_limit, mises_limit_cases  # noqa: E402

MISES_LIMIT = mises_limit_cases()


@pytest.mark.parametrize("path", ["host", "device", "resident"])
@pytest.mark.parametrize("case", MISES_LIMIT, ids=[c["name"] for c in MISES_LIMIT])
def test_comfe_mises_kernel_against_the_imported_reference_in_the_linear_hardening_limit(case, path):
    """The one pin the reference itself gives for the plastic branch of comfe-rs MisesPlasticity3D: its Python VonMises3D for
    w -> 0 is the same radial return with linear hardening (golden_util.mises_limit_cases: stress, alpha, plastic strain x
    sqrt(2/3), tangent + the rank-one term of the Rust text; tolerance = the limit's own error)."""
    law = fc.MisesPlasticityLinearHardening3D({k: np.array([v]) for k, v in case["params"].items()})
    n = case["grad"].size // 9
    s, t, h = case["stress_in"].copy(), np.full(36 * n, np.nan), {"history": case["history_in"].copy()}
    if path == "host":
        law.evaluate(0.0, 1.0, case["grad"].copy(), s, t, h)
    elif path == "device":
        sd, td = torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()
        hd = {"history": torch.from_numpy(h["history"]).cuda()}
        law.evaluate(0.0, 1.0, torch.from_numpy(case["grad"]).cuda(), sd, td, hd)
        s, t, h = sd.cpu().numpy(), td.cpu().numpy(), {"history": hd["history"].cpu().numpy()}
    else:  # the device-resident state: split history, sparse protocols
        from fenics_constitutive_amd.resident import ResidentState

        rs = ResidentState(law, n, stress0=s, history0=h)
        rs.evaluate_into(0.0, 1.0, case["grad"].copy(), s, t)
        h = {"history": rs.history["history"].cpu().numpy()}
    check_mises_limit(case, s, t, h["history"])


from golden_util import check_dp_j2, dp_j2_cases, dp_pressure_cases, dp_volumetric_cases  # noqa: E402

# b = b_flow = 0; b != 0 with b_flow = 0; b_flow != 0 on the classic surface (point-by-point reference calls)
DP_J2 = dp_j2_cases() + dp_pressure_cases() + dp_volumetric_cases()


@pytest.mark.parametrize("path", ["host", "device", "resident"])
@pytest.mark.parametrize("case", DP_J2, ids=[c["name"] for c in DP_J2])
def test_drucker_prager_kernels_against_the_imported_reference_for_b_zero(case, path):
    """The general return mapping where it coincides with the Python VonMises3D without hardening: b = b_flow = 0 (J2
    sub-family: stress, plastic strain, consistent tangent from the imported reference, golden_util.dp_j2_cases) and b != 0
    with b_flow = 0 (every point returns onto a cylinder whose radius follows from its trial pressure: point-by-point
    reference calls, tangent + a non-symmetric rank-one term, golden_util.dp_pressure_cases), and the classic surface with
    b_flow != 0 (the reference's deviatoric return + the volumetric part of the Rust flow rule for the reference's own plastic
    multiplier, verified to lie on the Rust yield surface, golden_util.dp_volumetric_cases).  The kernels iterate in
    invariant coordinates with a closed-form inverse; the pin is on the result."""
    cls = fc.DruckerPragerHyperbolic3D if case["hyperbolic"] else fc.DruckerPrager3D
    law = cls({k: np.array([v]) for k, v in case["params"].items()})
    n = case["grad"].size // 9
    s, t, h = case["stress_in"].copy(), np.full(36 * n, np.nan), {"history": case["history_in"].copy()}
    if path == "host":
        law.evaluate(0.0, 1.0, case["grad"].copy(), s, t, h)
    elif path == "device":
        sd, td = torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()
        hd = {"history": torch.from_numpy(h["history"]).cuda()}
        law.evaluate(0.0, 1.0, torch.from_numpy(case["grad"]).cuda(), sd, td, hd)
        law.device_stats()
        s, t, h = sd.cpu().numpy(), td.cpu().numpy(), {"history": hd["history"].cpu().numpy()}
    else:
        from fenics_constitutive_amd.resident import ResidentState

        rs = ResidentState(law, n, stress0=s, history0=h)
        rs.evaluate_into(0.0, 1.0, case["grad"].copy(), s, t)
        h = {"history": rs.history["history"].cpu().numpy()}
    check_dp_j2(case, s, t, h["history"])
