"""A trial state with a non-converged point must not be committed.

The reference raises INSIDE evaluate (mises_plasticity_isotropic_hardening.py:141-143; comfe-rs
general.rs:186 / drucker_prager_classic.rs:82), so its solver never reaches ``update()`` with such a
state.  The device launches of the resident states are asynchronous: ``update()`` (and
``evaluate(check=True)`` on device tensors) looks at the launch's counters and raises the same error
before anything is committed."""

import numpy as np
import pytest
import torch

import fenics_constitutive_amd as fc
from fenics_constitutive_amd.problem import ResidentProblemState
from fenics_constitutive_amd.resident import ResidentState
from test_oracle_c import NONCONVERGING, nonconverging_inputs

pytestmark = pytest.mark.gpu


def test_device_evaluate_check_raises_synchronously():
    law = fc.VonMises3D(NONCONVERGING)
    n = 70
    g, s, t, h = nonconverging_inputs(n)
    dev = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    with pytest.raises(RuntimeError, match="did not converge for plastic multiplier"):
        law.evaluate(0, 1.0, dev(g), dev(s), dev(t), {k: dev(v) for k, v in h.items()}, check=True)
    # a converging call with check=True passes
    law.evaluate(0, 1.0, dev(0.0 * g), dev(s), dev(t), {k: dev(v) for k, v in h.items()}, check=True)


@pytest.mark.parametrize("host", [False, True])
def test_resident_state_refuses_to_commit_nonconvergence(host):
    law = fc.VonMises3D(NONCONVERGING)
    n = 200
    g, s, t, h = nonconverging_inputs(n)
    st = ResidentState(law, n)
    committed = st.stress_committed.clone()
    if host:
        with pytest.raises(RuntimeError, match="did not converge"):  # the synchronous pass reports itself
            st.evaluate_into(0.0, 1.0, g, s, t)
    else:
        st.evaluate(0.0, 1.0, g)  # asynchronous: nothing raised yet
    with pytest.raises(RuntimeError, match="did not converge"):
        st.update()
    with pytest.raises(RuntimeError, match="did not converge"):  # and again: still nothing to commit
        st.update()
    assert torch.equal(st.stress_committed, committed)
    # a clean evaluate of the same increment can be committed
    st.evaluate(0.0, 1.0, 0.0 * g)
    st.update()
    assert st.check().n_nonconverged == 0


def test_states_sharing_one_law_keep_their_own_counters():
    """Two resident states on ONE law object: the second state's clean launch must not hide the first
    state's non-convergence (every state owns its counters, fcamd_eval_args.counters)."""
    law = fc.VonMises3D(NONCONVERGING)
    n = 130
    g, _, _, _ = nonconverging_inputs(n)
    bad, good = ResidentState(law, n), ResidentState(law, n)
    bad.evaluate(0.0, 1.0, g)
    good.evaluate(0.0, 1.0, 0.0 * g)
    good.update()
    with pytest.raises(RuntimeError, match="did not converge"):
        bad.update()
    assert bad.law.last_stats.n_nonconverged == n


def test_drucker_prager_messages_and_problem_state():
    """The multi-material state checks every law before the commit; the comfe-rs laws carry the
    messages of the host entries (general.rs:186)."""
    n = 256
    vm = fc.VonMises3D(NONCONVERGING)
    le = fc.LinearElasticityModel({"E": 42.0, "nu": 0.3}, fc.StressStrainConstraint.FULL)
    rows = [np.arange(0, n, 2, dtype=np.int32), np.arange(1, n, 2, dtype=np.int32)]
    ps = ResidentProblemState([(vm, rows[0]), (le, rows[1])], n)
    g_bad, _, _, _ = nonconverging_inputs(n // 2)
    g_le = np.full(9 * (n // 2), 1e-3)
    ps.evaluate([g_bad, g_le])
    with pytest.raises(RuntimeError, match="did not converge for plastic multiplier"):
        ps.update()
    s0 = ps.stress_0.clone()
    ps.evaluate([0.0 * g_bad, g_le])
    ps.update()
    assert not torch.equal(ps.stress_0, s0)  # the clean state was committed
    dp = fc.DruckerPrager3D({k: np.array([v]) for k, v in
                             {"mu": 80769.0, "kappa": 175000.0, "a": 100.0, "b": 0.05, "b_flow": 0.02}.items()})
    st = type("S", (), {"n_domain": 0, "n_nonconverged": 3})()
    with pytest.raises(RuntimeError, match="Plasticity3D: Newton-Raphson did not converge"):
        dp.raise_for_stats(st)
    st.n_domain = 1
    with pytest.raises(RuntimeError, match="non-differentiable tip"):
        dp.raise_for_stats(st)


def test_a_caught_failure_of_one_law_is_not_forgotten_by_the_next_laws_evaluate():
    """law 0's evaluate raises (sync=True), the caller catches it and goes on to law 1 -- update() must still
    refuse to commit; only a clean re-evaluate of law 0 clears the record."""
    n = 256
    vm = fc.VonMises3D(NONCONVERGING)
    le = fc.LinearElasticityModel({"E": 42.0, "nu": 0.3}, fc.StressStrainConstraint.FULL)
    rows = [np.arange(0, n, 2, dtype=np.int32), np.arange(1, n, 2, dtype=np.int32)]
    ps = ResidentProblemState([(vm, rows[0]), (le, rows[1])], n)
    g_bad, _, _, _ = nonconverging_inputs(n // 2)
    g_le = np.full(9 * (n // 2), 1e-3)
    sp, tp = np.zeros(6 * n), np.zeros(36 * n)
    with pytest.raises(RuntimeError, match="did not converge"):
        ps.evaluate_law_into(0, g_bad, sp, tp, sync=True)
    # a clean law in between must not clear law 0's failure: its own synchronising check still reports it ...
    with pytest.raises(RuntimeError, match="did not converge"):
        ps.evaluate_law_into(1, g_le, sp, tp, sync=True)
    assert ps._laws[0].failed is not None and ps._laws[1].failed is None
    with pytest.raises(RuntimeError, match="did not converge"):  # ... and nothing can be committed
        ps.update()
    ps.evaluate_law_into(0, 0.0 * g_bad, sp, tp, sync=True)
    ps.update()


def test_set_state_writes_both_history_copies():
    """a restart history must reach the trial copy too, or elastic points commit stale rows
    under the sparse protocol."""
    vm_p = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}
    law = fc.VonMises3D(vm_p)
    n = 1000
    rng = np.random.default_rng(5)
    st = ResidentState(law, n)
    h = {"eps_n": rng.normal(scale=1e-3, size=6 * n), "alpha": np.abs(rng.normal(scale=1e-3, size=n))}
    s = rng.normal(size=6 * n)
    st.set_state(s, h)
    g = rng.normal(scale=1e-6, size=9 * n)  # all elastic: no history row is rewritten
    st.evaluate(0.0, 1.0, g)
    st.update()
    assert np.array_equal(st.history_committed["eps_n"].cpu().numpy(), h["eps_n"])
    assert np.array_equal(st.history_committed["alpha"].cpu().numpy(), h["alpha"])
    # against a state that was constructed with the same values
    ref = ResidentState(law, n, stress0=s, history0=h)
    ref.evaluate(0.0, 1.0, g)
    ref.update()
    assert torch.equal(ref.stress_committed, st.stress_committed)
