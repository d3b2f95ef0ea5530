"""Non-finite and extreme inputs at a few points must neither hang a kernel (every per-point Newton
loop is bounded) nor leak into other points: the points that were not poisoned are bit-identical to a
clean run, whatever branch their tile-mates take."""

import numpy as np
import pytest
from test_gpu_parity import make_law, random_case

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

KINDS = ["linear_elasticity", "von_mises_3d", "spring_maxwell", "spring_kelvin", "comfe_linear_elasticity",
         "comfe_mises_plasticity"]
POISON = [float("nan"), float("inf"), -float("inf"), 1e300, -1e300, 1e-310, 0.0]


def run(law, g, s, h):
    d = lambda a: torch.from_numpy(a.copy()).cuda()  # noqa: E731
    sd, td = d(s), torch.zeros(6 * s.size, dtype=torch.float64, device="cuda")
    hd = None if h is None else {k: d(v) for k, v in h.items()}
    law.evaluate(0.0, 0.7, d(g), sd, td, hd)
    try:
        law.device_stats()  # synchronises; non-convergence at a poisoned point is allowed to raise
    except RuntimeError:
        pass
    return sd.cpu().numpy(), td.cpu().numpy(), None if hd is None else {k: v.cpu().numpy() for k, v in hd.items()}


@pytest.mark.timeout(120)
@pytest.mark.parametrize("kind", KINDS)
def test_poisoned_points_do_not_leak(kind):
    n = 64 * 5 + 3
    p, g, s, h = random_case(kind, n, seed=9)
    law = make_law(kind, p)
    clean = run(law, g, s, h)
    bad = np.array([3, 64, 130, 191, 200, 301, n - 1])
    gp, sp = g.copy(), s.copy()
    for pt, v in zip(bad, POISON):
        gp.reshape(-1, 9)[pt, pt % 9] = v
    sp.reshape(-1, 6)[bad[0], 2] = float("nan")
    hp = None if h is None else {k: v.copy() for k, v in h.items()}
    if hp is not None:
        first = next(iter(hp))
        hp[first].reshape(n, -1)[bad[1], 0] = float("inf")
    dirty = run(law, gp, sp, hp)
    ok = np.setdiff1d(np.arange(n), bad)
    assert np.array_equal(dirty[0].reshape(-1, 6)[ok], clean[0].reshape(-1, 6)[ok])
    assert np.array_equal(dirty[1].reshape(-1, 36)[ok], clean[1].reshape(-1, 36)[ok])
    if h is not None:
        for k in h:
            assert np.array_equal(dirty[2][k].reshape(n, -1)[ok], clean[2][k].reshape(n, -1)[ok]), k
    # and the law still works afterwards
    again = run(law, g, s, h)
    assert np.array_equal(again[0], clean[0]) and np.array_equal(again[1], clean[1])


@pytest.mark.timeout(120)
@pytest.mark.parametrize("hyper", [False, True])
def test_drucker_prager_poisoned_points(hyper):
    import fenics_constitutive_amd as fc
    from test_oracle_golden import DP_H, DP_P, dp_inputs

    p = DP_H if hyper else DP_P
    law = (fc.DruckerPragerHyperbolic3D if hyper else fc.DruckerPrager3D)({k: np.array([v]) for k, v in p.items()})
    n = 64 * 5 + 3
    g, s, h = dp_inputs(n, 4)
    clean = run(law, g, s, h)
    bad = np.array([3, 64, 130, 191, 200, 301, n - 1])
    gp = g.copy()
    for pt, v in zip(bad, POISON):
        gp.reshape(-1, 9)[pt, pt % 9] = v
    dirty = run(law, gp, s, h)
    ok = np.setdiff1d(np.arange(n), bad)
    assert np.array_equal(dirty[0].reshape(-1, 6)[ok], clean[0].reshape(-1, 6)[ok])
    assert np.array_equal(dirty[1].reshape(-1, 36)[ok], clean[1].reshape(-1, 36)[ok])
    assert np.array_equal(dirty[2]["history"].reshape(n, -1)[ok], clean[2]["history"].reshape(n, -1)[ok])
