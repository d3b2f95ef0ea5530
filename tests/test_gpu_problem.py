"""ResidentProblemState: several laws on disjoint cell sets of one mesh, state on the GPU, against
the reference protocol replayed on the host with the oracle (gather committed stress -> evaluate
with trial history reset -> scatter stress / tangent; commit by copying; solver/_solver.py:130-159,
solver/_lawonsubmesh.py:58-95)."""

import numpy as np
import pytest
from golden_util import rel_err
from test_gpu_parity import make_law, oracle_run, random_case

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from fenics_constitutive_amd.problem import ResidentProblemState, rows_of_cells  # noqa: E402


def test_rows_of_cells():
    assert rows_of_cells(np.array([2, 0]), 4).tolist() == [8, 9, 10, 11, 0, 1, 2, 3]


@pytest.mark.parametrize("sparse", [True, False])
@pytest.mark.parametrize("layout", ["interleaved_cells", "blocks"])
def test_three_materials_replay_reference_protocol(layout, sparse):
    kinds = ["linear_elasticity", "von_mises_3d", "spring_maxwell"]
    q, n_cells = 4, 1500
    n = q * n_cells
    rng = np.random.default_rng(11)
    owner = rng.integers(0, 3, size=n_cells) if layout == "interleaved_cells" else np.repeat([0, 1, 2], n_cells // 3)
    rows = [rows_of_cells(np.flatnonzero(owner == k), q) for k in range(3)]
    cases = [random_case(kind, r.size, seed=3 + i) for i, (kind, r) in enumerate(zip(kinds, rows))]
    laws = [make_law(kind, c[0]) for kind, c in zip(kinds, cases)]
    del_t = 0.7
    # host reference state
    stress_0 = np.zeros(6 * n)
    for r, c in zip(rows, cases):
        stress_0.reshape(-1, 6)[r] = c[2].reshape(-1, 6)
    hist_0 = [None if c[3] is None else {k: v.copy() for k, v in c[3].items()} for c in cases]
    st = ResidentProblemState(list(zip(laws, rows)), n, del_t=del_t, sparse_history=sparse)
    assert (st._laws[1].mask is not None) == sparse and st._laws[0].mask is None
    st.set_state(stress_0, hist_0)
    for inc in range(3):
        for it in range(2):
            # plastic sets that change from call to call (some calls nearly all elastic)
            hi = -2.0 if (inc + it) % 2 == 0 else -3.2
            grads = [rng.normal(size=9 * r.size) * np.repeat(10 ** rng.uniform(-4, hi, size=r.size), 9) for r in rows]
            # reference protocol on the host
            stress_1, tangent, hist_1 = stress_0.copy(), np.zeros(36 * n), []
            for kind, c, r, g, h0 in zip(kinds, cases, rows, grads, hist_0):
                s_sub = stress_0.reshape(-1, 6)[r].reshape(-1).copy()                      # map_to_sub(previous)
                s_new, t_new, h_new = oracle_run(kind, c[0], del_t, g, s_sub, h0)            # trial history = committed
                stress_1.reshape(-1, 6)[r] = s_new.reshape(-1, 6)                            # map_to_parent
                tangent.reshape(-1, 36)[r] = t_new.reshape(-1, 36)
                hist_1.append(h_new)
            st.evaluate(grads if it == 0 else [torch.from_numpy(g).cuda() for g in grads])
            st.check()
            assert rel_err(st.stress_1.cpu().numpy(), stress_1) <= 1e-9
            assert rel_err(st.tangent.cpu().numpy(), tangent) <= 1e-6
            assert rel_err(st.stress_0.cpu().numpy(), stress_0) <= 1e-9  # committed copy untouched by trial evaluations
            for hk, href in zip(st._history_1, hist_1):
                if href is not None:
                    for k in href:
                        assert rel_err(hk[k].cpu().numpy(), href[k]) <= 1e-6, k
        st.update()
        stress_0, hist_0 = stress_1, hist_1
        assert st._time == pytest.approx((inc + 1) * del_t)
    # the LE and Maxwell rows of the tangent were written once (del_t never changed), VonMises3D's every time
    assert [ls.tangent_key for ls in st._laws] == [0.0, None, del_t]
    s_out, t_out = np.empty(6 * n), np.empty(36 * n)
    st.evaluate(grads)
    st.download(s_out, t_out)
    assert np.array_equal(s_out, st.stress_1.cpu().numpy())


def test_single_law_and_guards():
    p, g, s, h = random_case("von_mises_3d", 777, seed=1)
    law = make_law("von_mises_3d", p)
    st = ResidentProblemState(law, 777, del_t=1.0)
    st.set_state(s, [h])
    with pytest.raises(RuntimeError):
        st.update()
    st.evaluate(g)
    ref = oracle_run("von_mises_3d", p, 1.0, g, s, h)
    assert rel_err(st.stress_1.cpu().numpy(), ref[0]) <= 1e-9 and rel_err(st.tangent.cpu().numpy(), ref[1]) <= 1e-6
    st.update()
    assert np.array_equal(st.stress_0.cpu().numpy(), st._stress[st._c].cpu().numpy())
    with pytest.raises(AssertionError):  # overlapping rows
        ResidentProblemState([(law, np.array([0, 1])), (law, np.array([1, 2]))], 3)


def test_tune_placement_keeps_results():
    """ResidentProblemState.tune_placement swaps the parent tangent array for the fastest of a few
    candidate allocations; stress, tangent (incl. the rows no law owns) and histories are unaffected."""
    n = 6000
    rng = np.random.default_rng(3)
    perm = rng.permutation(n)
    rows = [np.sort(perm[:2500]).astype(np.int32), np.sort(perm[2500:5200]).astype(np.int32)]  # 800 points unowned
    kinds = ["linear_elasticity", "von_mises_3d"]
    cases = [random_case(k, r.size, seed=11 + i) for i, (k, r) in enumerate(zip(kinds, rows))]
    laws = [make_law(k, c[0]) for k, c in zip(kinds, cases)]
    grads = [c[1] for c in cases]
    stress = rng.normal(size=6 * n)
    hist = [c[3] for c in cases]
    a = ResidentProblemState(list(zip(laws, rows)), n, del_t=1.0)
    b = ResidentProblemState(list(zip(laws, rows)), n, del_t=1.0)
    for st in (a, b):
        st.set_state(stress, hist)
    a.evaluate(grads)
    info = b.tune_placement(grads, tries=3)
    assert len(info["candidate_ms"]) == 3
    torch.cuda.synchronize()
    assert torch.equal(a.stress_1, b.stress_1) and torch.equal(a.tangent, b.tangent)
    for ha, hb in zip(a._history_1, b._history_1):
        if ha is not None:
            for k in ha:
                assert torch.equal(ha[k], hb[k])
    b.evaluate(grads)  # the constant LE rows are not rewritten and still there
    torch.cuda.synchronize()
    assert torch.equal(a.tangent, b.tangent)
