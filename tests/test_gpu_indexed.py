"""Submesh-indexed evaluate: gather of the committed stress and scatter of stress /
tangent through a parent-row index, fused into the kernel, must equal the reference sequence
map_to_sub -> evaluate -> map_to_parent (solver/_lawonsubmesh.py:58-95, solver/maps.py:82-123)."""

import numpy as np
import pytest
from golden_util import rel_err
from test_gpu_parity import CLASS, TOL, make_law, oracle_run, random_case

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

FULL_KINDS = ["linear_elasticity", "von_mises_3d", "spring_maxwell", "spring_kelvin", "comfe_linear_elasticity", "comfe_mises_plasticity"]


@pytest.mark.parametrize("order", ["sorted", "random", "cells"])
@pytest.mark.parametrize("n_sub", [1, 63, 64, 65, 1000, 20_011])
@pytest.mark.parametrize("kind", FULL_KINDS)
def test_indexed_equals_gather_evaluate_scatter(kind, n_sub, order):
    rng = np.random.default_rng(n_sub + 3)
    n_parent = 2 * n_sub + 17
    if order == "cells":  # what build_subspace_map yields (maps.py:159-161): ascending cells, 4 consecutive rows each, last one ragged
        n_parent = 8 * ((n_sub + 3) // 4) + 17
        cells = np.sort(rng.choice(n_parent // 4, size=(n_sub + 3) // 4, replace=False))
        rows = (cells[:, None] * 4 + np.arange(4)[None, :]).reshape(-1)[:n_sub]
    else:
        rows = rng.choice(n_parent, size=n_sub, replace=False)
    if order == "sorted":
        rows.sort()
    p, g, s_sub, h = random_case(kind, n_sub, seed=n_sub + 5)
    # parent arrays: committed stress everywhere, this law's rows = s_sub
    s_prev_parent = rng.normal(size=6 * n_parent)
    s_prev_parent.reshape(-1, 6)[rows] = s_sub.reshape(-1, 6)
    s_cur_parent = rng.normal(size=6 * n_parent)      # rows of other laws must survive
    t_parent = rng.normal(size=36 * n_parent)
    # reference sequence on the CPU oracle
    s_ref, t_ref, h_ref = oracle_run(kind, p, 0.7, g, s_sub, h)
    exp_s, exp_t = s_cur_parent.copy(), t_parent.copy()
    exp_s.reshape(-1, 6)[rows] = s_ref.reshape(-1, 6)
    exp_t.reshape(-1, 36)[rows] = t_ref.reshape(-1, 36)
    # fused launch
    law = make_law(kind, p)
    d = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    sp, sc, tp = d(s_prev_parent), d(s_cur_parent), d(t_parent)
    hd = None if h is None else {k: d(v) for k, v in h.items()}
    law.evaluate_indexed(0.0, 0.7, d(g), sp, sc, tp, torch.from_numpy(rows.astype(np.int32)).cuda(), None, hd)
    torch.cuda.synchronize()
    tol = TOL[CLASS[kind]]
    got_s, got_t = sc.cpu().numpy(), tp.cpu().numpy()
    other = np.setdiff1d(np.arange(n_parent), rows)
    assert np.array_equal(got_s.reshape(-1, 6)[other], s_cur_parent.reshape(-1, 6)[other])
    assert np.array_equal(got_t.reshape(-1, 36)[other], t_parent.reshape(-1, 36)[other])
    assert rel_err(got_s.reshape(-1, 6)[rows], s_ref.reshape(-1, 6)) <= tol
    assert rel_err(got_t.reshape(-1, 36)[rows], t_ref.reshape(-1, 36)) <= tol
    assert np.array_equal(sp.cpu().numpy(), s_prev_parent)  # committed parent stress untouched
    if h is not None:
        for k in h:
            assert rel_err(hd[k].cpu().numpy(), h_ref[k]) <= tol, k


def test_two_materials_cover_the_parent():
    """Two laws on complementary halves of one parent mesh (tests/models/test_elasticity.py:90-154 in
    array form): after both launches every parent row is written exactly once."""
    rng = np.random.default_rng(0)
    n_parent = 10_000
    perm = rng.permutation(n_parent)
    rows_a, rows_b = np.sort(perm[: n_parent // 2]), np.sort(perm[n_parent // 2 :])
    sp = torch.zeros(6 * n_parent, dtype=torch.float64, device="cuda")
    sc = torch.full((6 * n_parent,), float("nan"), dtype=torch.float64, device="cuda")
    tp = torch.full((36 * n_parent,), float("nan"), dtype=torch.float64, device="cuda")
    import fenics_constitutive_amd as fc

    FULL = fc.StressStrainConstraint.FULL
    for E, rows in ((42.0, rows_a), (4.2, rows_b)):
        law = fc.LinearElasticityModel({"E": E, "nu": 0.3}, FULL)
        g = torch.from_numpy(rng.normal(scale=1e-3, size=9 * rows.size)).cuda()
        law.evaluate_indexed(0, 1, g, sp, sc, tp, torch.from_numpy(rows.astype(np.int32)).cuda(), None, None)
    assert not torch.isnan(sc).any() and not torch.isnan(tp).any()
    D_a = fc.get_elastic_tangent(42.0, 0.3, FULL).reshape(-1)
    assert np.array_equal(tp.cpu().numpy().reshape(-1, 36)[rows_a[0]], D_a)
