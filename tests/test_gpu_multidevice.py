"""One process, several GPUs (fcamd_multi, include/fcamd.h): every device evaluates its own slice of the caller's
host arrays in place.  A one-GPU box tests it with several contexts on device 0 (``devices=[0, 0, ...]``): the slicing,
the shared page locks, the worker threads and the resident state are the same code with one device or eight.
Everything must be BIT-identical to the single-device entries: the points are independent and the slices start on
tile boundaries."""

import numpy as np
import pytest
from golden_util import load_calls, rel_err

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import fenics_constitutive_amd as fc  # noqa: E402
from fenics_constitutive_amd import _capi  # noqa: E402
from fenics_constitutive_amd.multidevice import MultiDeviceResidentState  # noqa: E402
from fenics_constitutive_amd.resident import ResidentState  # noqa: E402
from test_gpu_resident import _sparse_case  # noqa: E402

FULL = fc.StressStrainConstraint.FULL
VM_P = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}
SLS_P = {"E0": 42.0, "E1": 10.0, "tau": 10.0, "nu": 0.2}
RS_P = {"mu": 80769.0, "kappa": 175000.0, "y_0": 1200.0, "h": 200.0}
DP_P = {"mu": 80769.0, "kappa": 175000.0, "a": 100.0, "b": 0.05, "b_flow": 0.02}


def rs(p):
    return {k: np.array([v]) for k, v in p.items()}


LAWS = {
    "le": lambda c=FULL: fc.LinearElasticityModel({"E": 42.0, "nu": 0.3}, c),
    "vm": lambda: fc.VonMises3D(VM_P),
    "maxwell": lambda c=FULL: fc.SpringMaxwellModel(SLS_P, c),
    "kelvin": lambda c=FULL: fc.SpringKelvinModel(SLS_P, c),
    "comfe_le": lambda: fc.LinearElasticity3D(rs({"mu": 16.0, "kappa": 35.0})),
    "comfe_mises": lambda: fc.MisesPlasticityLinearHardening3D(rs(RS_P)),
    "dp": lambda: fc.DruckerPrager3D(rs(DP_P)),
    "dp_hyper": lambda: fc.DruckerPragerHyperbolic3D(rs({**DP_P, "d": 40.0})),
}


def inputs(law, n, rng, kind):
    gd2, sd = law.geometric_dim**2, law.stress_strain_dim
    scale = 10 ** rng.uniform(-4.5, -2.0, size=n) if kind in ("vm", "comfe_mises", "dp", "dp_hyper") else np.full(n, 1e-3)
    g = (rng.normal(size=(n, gd2)) * scale[:, None])
    s = rng.normal(scale=30.0, size=(n, sd))
    if kind.startswith("dp"):
        s[:, :3] -= 1000.0
        g[:, [0, 4, 8]] -= (0.95 * g[:, [0, 4, 8]].sum(axis=1) / 3.0)[:, None]
    h = None
    if law.history_dim is not None:
        h = {k: np.abs(rng.normal(scale=1e-3, size=d * n)) for k, d in law.history_dim.items()}
    return g.reshape(-1).copy(), s.reshape(-1).copy(), h


def evaluate_copy(law, del_t, g, s, h, tangent=True):
    s2, t2 = s.copy(), (np.full(law.stress_strain_dim**2 * (s.size // law.stress_strain_dim), np.nan) if tangent else None)
    h2 = None if h is None else {k: v.copy() for k, v in h.items()}
    law.evaluate(0.0, del_t, g, s2, t2, h2)
    return s2, t2, h2


def assert_same(a, b, what=""):
    for x, y, name in zip(a[:2], b[:2], ("stress", "tangent")):
        if x is None:
            assert y is None
            continue
        assert not np.isnan(x).any(), f"{what} {name} has unwritten entries"
        assert np.array_equal(x, y), f"{what} {name} differs"
    if a[2] is not None:
        for k in a[2]:
            assert np.array_equal(a[2][k], b[2][k]), f"{what} history[{k}] differs"


@pytest.mark.parametrize("kind", sorted(LAWS))
@pytest.mark.parametrize("n,devices", [(50_003, [0, 0]), (100_037, [0, 0, 0]), (40_000, [0, 0, 0, 0])])
def test_multi_host_equals_single_device(kind, n, devices):
    rng = np.random.default_rng(n)
    single, multi = LAWS[kind](), LAWS[kind]().use_devices(devices)
    g, s, h = inputs(single, n, rng, kind)
    ref = evaluate_copy(single, 2.0, g, s, h)
    got = evaluate_copy(multi, 2.0, g, s, h)
    assert_same(got, ref, kind)
    mode, used = multi._multi().last_host_mode()
    assert used == min(len(devices), n // _capi.MULTI_MIN_POINTS) and used > 1
    assert mode & _capi.HOST_ZERO_COPY_OUT and mode & _capi.HOST_TEMP_LOCK  # one launch per device on the locked caller arrays
    if single.last_stats is not None and kind in ("vm", "comfe_mises", "dp", "dp_hyper"):
        assert multi.last_stats.n_plastic == single.last_stats.n_plastic > 0
        assert multi.last_stats.n_newton_iters == single.last_stats.n_newton_iters
    # without a tangent (comfe-rs/src/interfaces.rs:383-394)
    assert_same(evaluate_copy(multi, 2.0, g, s, h, tangent=False), evaluate_copy(single, 2.0, g, s, h, tangent=False), kind)


@pytest.mark.parametrize("kind", ["le", "maxwell", "kelvin"])
@pytest.mark.parametrize("constraint", ["UNIAXIAL_STRAIN", "UNIAXIAL_STRESS", "PLANE_STRAIN", "PLANE_STRESS"])
def test_multi_host_low_dimensional_constraints(kind, constraint):
    c = fc.StressStrainConstraint[constraint]
    n = 70_001
    single, multi = LAWS[kind](c), LAWS[kind](c).use_devices([0, 0, 0])
    g, s, h = inputs(single, n, np.random.default_rng(5), kind)
    assert_same(evaluate_copy(multi, 0.5, g, s, h), evaluate_copy(single, 0.5, g, s, h), f"{kind}/{constraint}")


GOLDEN = [(f, k, c) for f, k in [("linear_elasticity.npz", "le"), ("von_mises_3d.npz", "vm"),
                                 ("spring_maxwell.npz", "maxwell"), ("spring_kelvin.npz", "kelvin")]
          for c in load_calls(f)]
CTOR = {"le": lambda p: fc.LinearElasticityModel(p, FULL), "vm": fc.VonMises3D,
        "maxwell": lambda p: fc.SpringMaxwellModel(p, FULL), "kelvin": lambda p: fc.SpringKelvinModel(p, FULL)}


@pytest.mark.parametrize("fname,kind,c", GOLDEN, ids=[f"{k}-{c.name}" for _, k, c in GOLDEN])
def test_golden_vectors_over_three_contexts(fname, kind, c):
    """The reference's golden calls (n = 1 .. 257) with the slicing forced down to one tile per device: bit-equal to the
    single-device result and within the parity tolerance of the reference's own output."""
    single, multi = CTOR[kind](c.params), CTOR[kind](c.params).use_devices([0, 0, 0])
    multi._multi().set_option("min_points", 64)
    s, t, h = c.fresh()
    single.evaluate(0.0, c.del_t, c.grad.copy(), s, t, h)
    s2, t2, h2 = c.fresh()
    multi.evaluate(0.0, c.del_t, c.grad.copy(), s2, t2, h2)
    assert_same((s2, t2, h2), (s, t, h), f"{kind}/{c.name}")
    tol = 1e-6 if kind == "vm" else 1e-10
    assert rel_err(s2, c.stress_out) <= tol and rel_err(t2, c.tangent_out) <= tol
    assert multi._multi().last_host_mode()[1] == max(1, min(3, c.n // 64))


def test_small_calls_use_one_device_and_plan_matches_shard_bounds():
    law = LAWS["vm"]().use_devices([0, 0, 0, 0])
    m = law._multi()
    assert m.plan(1000) == 1 and m.plan(8192 * 2 + 5) == 2 and m.plan(10**6) == 4 and m.plan(0) == 1
    n = 1_000_003
    for k in range(4):
        assert m.bounds(n, k) == _capi.shard_bounds(n, 4, k)
    assert m.bounds(20_000, 3) == (20_000, 20_000)  # slot not used by a call of this size
    g, s, h = inputs(law, 1000, np.random.default_rng(1), "vm")
    assert_same(evaluate_copy(law, 1.0, g, s, h), evaluate_copy(LAWS["vm"](), 1.0, g, s, h))
    assert m.last_host_mode()[1] == 1
    # empty call
    law.evaluate(0.0, 1.0, np.zeros(0), np.zeros(0), np.zeros(0), {"eps_n": np.zeros(0), "alpha": np.zeros(0)})


def test_registered_arrays_skip_the_per_call_page_lock():
    n = 60_000
    law = LAWS["vm"]().use_devices([0, 0])
    g, s, h = inputs(law, n, np.random.default_rng(2), "vm")
    ref = evaluate_copy(LAWS["vm"](), 1.0, g, s, h)
    s2, t2, h2 = s.copy(), np.full(36 * n, np.nan), {k: v.copy() for k, v in h.items()}
    law.pin_host_arrays(g, s2, t2, *h2.values())
    try:
        law.evaluate(0.0, 1.0, g, s2, t2, h2)
        mode, used = law._multi().last_host_mode()
        assert used == 2 and mode == (_capi.HOST_ZERO_COPY_IN | _capi.HOST_ZERO_COPY_OUT)  # no HOST_TEMP_LOCK
        assert_same((s2, t2, h2), ref)
    finally:
        law.unpin_arrays()
    # after unpinning the same arrays are page-locked per call again
    s3, t3, h3 = s.copy(), np.full(36 * n, np.nan), {k: v.copy() for k, v in h.items()}
    law.evaluate(0.0, 1.0, g, s3, t3, h3)
    assert law._multi().last_host_mode()[0] & _capi.HOST_TEMP_LOCK
    assert_same((s3, t3, h3), ref)


def test_error_conventions_are_the_single_device_ones():
    n = 30_000
    sls = LAWS["maxwell"]().use_devices([0, 0])
    g, s, h = inputs(sls, n, np.random.default_rng(3), "maxwell")
    with pytest.raises(AssertionError):
        sls.evaluate(0.0, 0.0, g, s.copy(), np.zeros(36 * n), {k: v.copy() for k, v in h.items()})
    with pytest.raises(ValueError):
        sls.evaluate(0.0, 1.0, g, s.copy(), np.zeros(36 * n), None)
    with pytest.raises(AssertionError):
        sls.evaluate(0.0, 1.0, g[:-9], s.copy(), np.zeros(36 * n), h)
    # Newton non-convergence in ONE slice only: the reference's RuntimeError (mises_plasticity_isotropic_hardening.py:141-143),
    # the counters are the sums over the slices
    from test_oracle_c import NONCONVERGING, nonconverging_inputs

    vm = fc.VonMises3D(NONCONVERGING).use_devices([0, 0, 0])
    g, s, t, h = nonconverging_inputs(n)
    g[: 9 * (n - 7)] = 0.0  # only the last 7 points (third slice) fail
    with pytest.raises(RuntimeError, match="did not converge"):
        vm.evaluate(0.0, 1.0, g, s, t, h)
    assert vm._multi().last_host_mode()[1] == 3
    vm.evaluate(0.0, 1.0, 0.0 * g, s, t, h)
    assert vm.last_stats.n_nonconverged == 0


@pytest.mark.parametrize("law_name", ["VonMises3D", "MisesPlasticityLinearHardening3D", "DruckerPrager3D",
                                      "DruckerPragerHyperbolic3D"])
@pytest.mark.parametrize("pinned", [False, True])
def test_multi_device_resident_state_equals_resident_state(law_name, pinned):
    """Increments with Newton re-evaluations, plastic sets that grow, shrink and vanish, commits: the state sliced over
    three contexts must hand the assembler exactly the arrays the single-device ResidentState does, every iteration --
    with the sparse tangent, the split history and (pinned) the one-launch zero-copy pass all on."""
    n = 64 * 400 + 29
    rng = np.random.default_rng(11)
    law, s0, h0, grad = _sparse_case(law_name, n, rng)
    one = ResidentState(law, n, stress0=s0, history0=h0, placement="torch")
    multi = MultiDeviceResidentState(type(law)(_ctor_params(law)), n, devices=[0, 0, 0], stress0=s0, history0=h0)
    assert multi.slices()[0][0] == 0 and multi.slices()[-1][1] == n
    sa, ta, sb, tb = np.zeros(6 * n), np.full(36 * n, np.nan), np.zeros(6 * n), np.full(36 * n, np.nan)
    gbuf = np.zeros(9 * n)
    if pinned:
        multi.pin_host_arrays(gbuf, sb, tb)
    n_plastic = []
    for inc in range(4):
        for it in range(3):
            gbuf[:] = grad(all_elastic=(inc == 2 and it == 1), zoned=(inc % 2 == 1)).cpu().numpy()
            st1 = one.evaluate_into(0.0, 1.0, gbuf, sa, ta)
            st2 = multi.evaluate_into(0.0, 1.0, gbuf, sb, tb)
            n_plastic.append(int(st2.n_plastic))
            assert st1.n_plastic == st2.n_plastic and st1.n_newton_iters == st2.n_newton_iters
            assert np.array_equal(sa, sb), (inc, it)
            assert np.array_equal(ta, tb), (inc, it)
            ht, hc = multi.history, multi.history_committed
            for k in h0:
                assert np.array_equal(ht[k], one.history[k].cpu().numpy()), (inc, it, k)
                assert np.array_equal(hc[k], one.history_committed[k].cpu().numpy()), (inc, it, k)
            assert np.array_equal(multi.stress_committed, one.stress_committed.cpu().numpy())
            mode, used = multi.last_host_mode()
            assert used == 3 and mode & _capi.HOST_ZERO_COPY_OUT and bool(mode & _capi.HOST_TEMP_LOCK) == (not pinned)
        one.update()
        multi.update()
    assert max(n_plastic) > 0.1 * n and min(n_plastic) < 0.75 * max(n_plastic)  # the sets really changed
    multi.close()


def _ctor_params(law):
    """constructor argument that rebuilds ``law`` (a second object: the multi-device state owns its own handles)"""
    name = type(law).__name__
    if name == "VonMises3D":
        return {k: getattr(law, k) for k in ("p_ka", "p_mu", "p_y0", "p_y00", "p_w")}
    keys = {"MisesPlasticityLinearHardening3D": ("mu", "kappa", "y_0", "h"), "DruckerPrager3D": ("mu", "kappa", "a", "b", "b_flow"),
            "DruckerPragerHyperbolic3D": ("mu", "kappa", "a", "b", "d", "b_flow")}[name]
    return {k: np.array([v]) for k, v in zip(keys, law._parameter_vector)}


@pytest.mark.parametrize("fname,cls,steps", [("spring_maxwell.npz", "SpringMaxwellModel", 5), ("spring_kelvin.npz", "SpringKelvinModel", 5),
                                             ("von_mises_3d.npz", "VonMises3D", 4)])
def test_multi_device_resident_state_replays_the_golden_sequences(fname, cls, steps):
    calls = {c.name: c for c in load_calls(fname)}
    vm = cls == "VonMises3D"
    name = (lambda k, it: f"mixed_step{k}_iter{it}") if vm else (lambda k, it: f"step{k}_iter{it}")
    c0 = calls[name(0, 0)]
    law = fc.VonMises3D(c0.params) if vm else getattr(fc, cls)(c0.params, FULL)
    st = MultiDeviceResidentState(law, c0.n, devices=[0, 0], stress0=c0.stress_in, history0=c0.hist_in)
    tol = 1e-6 if vm else 1e-10
    s, t = np.empty(6 * c0.n), np.full(36 * c0.n, np.nan)
    for k in range(steps):
        for it in (0, 1):
            c = calls[name(k, it)]
            st.evaluate_into(0.0, c.del_t, c.grad.copy(), s, t)
            assert rel_err(s, c.stress_out) <= tol and rel_err(t, c.tangent_out) <= tol
            h = st.history
            for key in c.hist_out:
                assert rel_err(h[key], c.hist_out[key]) <= tol
            assert np.array_equal(st.stress_committed, c.stress_in)
        st.update()
    assert np.array_equal(st.stress_committed, s)


def test_multi_device_state_never_commits_a_failed_evaluate():
    from test_oracle_c import NONCONVERGING, nonconverging_inputs

    n = 20_000
    st = MultiDeviceResidentState(fc.VonMises3D(NONCONVERGING), n, devices=[0, 0])
    with pytest.raises(RuntimeError):
        st.update()  # nothing evaluated yet
    bad, s, t, _ = nonconverging_inputs(n)
    bad[: 9 * (n - 3)] = 0.0  # three failing points, all in the second slice
    with pytest.raises(RuntimeError, match="did not converge"):
        st.evaluate_into(0.0, 1.0, bad, s, t)
    with pytest.raises(RuntimeError, match="nothing to commit"):
        st.update()
    committed = st.stress_committed
    assert not committed.any()
    st.evaluate_into(0.0, 1.0, 0.0 * bad, s, t)
    st.update()
    st.close()


def test_use_resident_state_with_devices_equals_the_unpatched_problem():
    """integration.use_resident_state(problem, devices=[0, 0]) on the stand-ins of tests/test_gpu_integration.py: two
    laws on interleaved cells, every law's state sliced over two contexts, bit-identical global arrays and histories."""
    import test_gpu_integration as TI
    from fenics_constitutive_amd.integration import use_resident_state

    n_cells, q = 9000, 4
    a, b = TI.build(n_cells, q, 7), TI.build(n_cells, q, 7)
    states = use_resident_state(b, devices=[0, 0])
    assert len(states) == 2 and all(isinstance(s, MultiDeviceResidentState) for s in states)
    rng = np.random.default_rng(3)
    for inc in range(3):
        for it in range(3):
            for los_a, los_b in zip(a._law_on_submeshs, b._law_on_submeshs):
                m = los_a.stress.x.array.size // 6
                g = rng.normal(size=9 * m) * np.repeat(10 ** rng.uniform(-4, -1.8 if it else -3.5, size=m), 9)
                a.incr_disp.grads[los_a.cells.tobytes()] = g
                b.incr_disp.grads[los_b.cells.tobytes()] = g
            a.form()
            b.form()
            assert np.array_equal(a.stress.current.x.array, b.stress.current.x.array), (inc, it)
            assert np.array_equal(a.tangent.x.array, b.tangent.x.array), (inc, it)
        a.update()
        b.update()
        for los_a, los_b in zip(a._law_on_submeshs, b._law_on_submeshs):
            for k in los_a.history.history_0:
                assert np.array_equal(los_a.history.history_0[k].x.array, los_b.history.history_0[k].x.array), (inc, k)
    assert states[0].last_host_mode()[1] == 2
    for s in states:
        s.close()


@pytest.mark.parametrize("permuted", [False, True])
def test_fused_problem_state_over_device_contexts_equals_unpatched(permuted):
    """use_resident_problem_state(problem, devices=[0, 0, 0]): the fused multi-material host flow (kernels write their rows of
    the GLOBAL stress / tangent host arrays, no submesh maps) with every law's points cut into three device slices --
    bit-identical global arrays and histories against the unpatched reference protocol (stand-ins of test_gpu_integration)."""
    import test_gpu_integration as TI
    from fenics_constitutive_amd.integration import use_resident_problem_state
    from fenics_constitutive_amd.multidevice import MultiDeviceProblemState

    n_cells, q = 1500, 4
    a, b = TI.build(n_cells, q, 13), TI.build(n_cells, q, 13)
    if permuted:
        rng_p = np.random.default_rng(1)
        for pa, pb in zip(a._law_on_submeshs, b._law_on_submeshs):
            perm = rng_p.permutation(pa.submesh_map.parent.size)
            pa.submesh_map.sub, pb.submesh_map.sub = perm, perm.copy()
            for k, f in pa.history.history_0.items():
                d = f.x.array.size // perm.size
                v = f.x.array.reshape(-1, d).copy()
                f.x.array.reshape(-1, d)[perm] = v
                pb.history.history_0[k].x.array[:] = f.x.array
    for los in b._law_on_submeshs:  # no map call may survive the patch
        los.submesh_map.map_to_sub = los.submesh_map.map_to_parent = None
    state = use_resident_problem_state(b, devices=[0, 0, 0])
    assert isinstance(state, MultiDeviceProblemState) and len(state.states) == 3
    rng = np.random.default_rng(8)
    try:
        for inc in range(3):
            for it in range(3):
                for los_a, los_b in zip(a._law_on_submeshs, b._law_on_submeshs):
                    m = los_a.stress.x.array.size // 6
                    g = rng.normal(size=9 * m) * np.repeat(10 ** rng.uniform(-4, -1.8 if it else -3.5, size=m), 9)
                    a.incr_disp.grads[los_a.cells.tobytes()] = g
                    b.incr_disp.grads[los_b.cells.tobytes()] = g
                a.form()
                b.form()
                assert np.array_equal(a.stress.current.x.array, b.stress.current.x.array), (inc, it)
                assert np.array_equal(a.tangent.x.array, b.tangent.x.array), (inc, it)
            a.update()
            b.update()
            for los_a, los_b in zip(a._law_on_submeshs, b._law_on_submeshs):
                for k in los_a.history.history_0:
                    assert np.array_equal(los_a.history.history_0[k].x.array, los_b.history.history_0[k].x.array), (inc, k)
        assert all(s._tangent is None for s in state.states)  # the parent tangent never existed on a device
    finally:
        state.close()


def test_multi_device_problem_state_never_commits_a_failed_law():
    from fenics_constitutive_amd.multidevice import MultiDeviceProblemState
    from test_oracle_c import NONCONVERGING, nonconverging_inputs

    n = 4096
    vm = fc.VonMises3D(NONCONVERGING)
    le = fc.LinearElasticityModel({"E": 42.0, "nu": 0.3}, FULL)
    rows = [np.arange(0, n, 2, dtype=np.int32), np.arange(1, n, 2, dtype=np.int32)]
    ps = MultiDeviceProblemState([(vm, rows[0]), (le, rows[1])], n, devices=[0, 0])
    g_bad, _, _, _ = nonconverging_inputs(n // 2)
    g_bad[: 9 * (n // 2 - 5)] = 0.0  # the failing points lie in the second device's slice only
    g_le = np.full(9 * (n // 2), 1e-3)
    sp, tp, g_ok = np.zeros(6 * n), np.zeros(36 * n), 0.0 * g_bad
    ps.pin_host_arrays(sp, tp, g_bad, g_le, g_ok)  # page-locked: the asynchronous launches of the fused flow
    ps.evaluate_law_into(0, g_bad, sp, tp, sync=False)
    with pytest.raises(RuntimeError, match="did not converge"):
        ps.evaluate_law_into(1, g_le, sp, tp, sync=True)
    with pytest.raises(RuntimeError, match="did not converge"):
        ps.update()
    ps.evaluate_law_into(0, g_ok, sp, tp, sync=True)
    ps.update()
    assert ps._time == 1.0 and all(s._time == 1.0 for s in ps.states)
    ps.close()


def test_a_range_registered_by_one_context_is_entered_by_another():
    """Two threads, a context each (one process driving several GPUs looks the same to the library): the second
    registration of one array enters the first one's page lock instead of locking again (a second hipHostRegister of a
    registered address 'succeeds' on this runtime and its unregister would take the owner's lock away)."""
    import threading

    n = 40_000
    law = LAWS["vm"]()
    g, s, h = inputs(law, n, np.random.default_rng(9), "vm")
    ref = evaluate_copy(LAWS["vm"](), 1.0, g, s, h)
    arrays = {"s": s.copy(), "t": np.full(36 * n, np.nan), "e": h["eps_n"].copy(), "a": h["alpha"].copy()}
    main_ctx = _capi.get_context(0)
    for x in (g, *arrays.values()):
        main_ctx.register_host_buffer(x)
    result = {}

    def other_thread():
        try:
            ctx = _capi.get_context(0)  # thread-local: a context of its own
            assert ctx is not main_ctx
            for x in (g, *arrays.values()):
                ctx.register_host_buffer(x)  # enters the main thread's registration
            law.evaluate(0.0, 1.0, g, arrays["s"], arrays["t"], {"eps_n": arrays["e"], "alpha": arrays["a"]})
            result["mode"] = ctx.last_host_mode()
            for x in (g, *arrays.values()):
                ctx.unregister_host_buffer(x)  # the borrower leaves; the page lock stays
        except Exception as e:  # noqa: BLE001
            result["error"] = e

    t = threading.Thread(target=other_thread)
    t.start()
    t.join()
    assert "error" not in result, result
    assert result["mode"] == (_capi.HOST_ZERO_COPY_IN | _capi.HOST_ZERO_COPY_OUT)  # no per-call lock: the range was registered
    assert_same((arrays["s"], arrays["t"], {"eps_n": arrays["e"], "alpha": arrays["a"]}), ref)
    # the owner's registration is intact: another zero-copy call from this thread, then the real unlock
    s2, t2, h2 = s.copy(), arrays["t"], {"eps_n": h["eps_n"].copy(), "alpha": h["alpha"].copy()}
    arrays["s"][:] = s
    arrays["e"][:], arrays["a"][:] = h["eps_n"], h["alpha"]
    law.evaluate(0.0, 1.0, g, arrays["s"], arrays["t"], {"eps_n": arrays["e"], "alpha": arrays["a"]})
    assert main_ctx.last_host_mode() == (_capi.HOST_ZERO_COPY_IN | _capi.HOST_ZERO_COPY_OUT)
    assert_same((arrays["s"], arrays["t"], {"eps_n": arrays["e"], "alpha": arrays["a"]}), ref)
    for x in (g, *arrays.values()):
        main_ctx.unregister_host_buffer(x)
    del s2, t2, h2


def test_shared_page_lock_outlives_the_context_that_took_it():
    """a range entered by a second context must stay page-locked when the FIRST registrant unregisters it (or
    is destroyed) -- the lock is shared and reference-counted, the last holder unlocks the pages.  Before round 4 the
    second context kept an entry that still claimed 'page-locked, zero copy' and the next launch DMA'd into unpinned memory."""
    import threading

    n = 30_000
    law = LAWS["vm"]()
    g, s, h = inputs(law, n, np.random.default_rng(11), "vm")
    ref = evaluate_copy(LAWS["vm"](), 1.0, g, s, h)
    arrays = {"s": s.copy(), "t": np.full(36 * n, np.nan), "e": h["eps_n"].copy(), "a": h["alpha"].copy()}
    everything = (g, *arrays.values())
    main_ctx = _capi.get_context(0)
    ready, go, result = threading.Event(), threading.Event(), {}

    def first_registrant():  # a context of its own (thread-local), which takes the page locks and then goes away
        try:
            ctx = _capi.get_context(0)
            for x in everything:
                ctx.register_host_buffer(x)
            ready.set()
            go.wait(30)
            for x in everything[:2]:
                ctx.unregister_host_buffer(x)  # two ranges left explicitly ...
            ctx.close()                         # ... the others with the context's destruction
        except Exception as e:  # noqa: BLE001
            result["error"] = e
            ready.set()

    t = threading.Thread(target=first_registrant)
    t.start()
    assert ready.wait(60) and "error" not in result, result
    for x in everything:
        main_ctx.register_host_buffer(x)  # enters the other context's locks
    go.set()
    t.join()
    assert "error" not in result, result
    # the first registrant is gone; this context's ranges are still page-locked: zero copy, right results, twice
    for _ in range(2):
        arrays["s"][:], arrays["e"][:], arrays["a"][:] = s, h["eps_n"], h["alpha"]
        law.evaluate(0.0, 1.0, g, arrays["s"], arrays["t"], {"eps_n": arrays["e"], "alpha": arrays["a"]})
        assert main_ctx.last_host_mode() == (_capi.HOST_ZERO_COPY_IN | _capi.HOST_ZERO_COPY_OUT)
        assert_same((arrays["s"], arrays["t"], {"eps_n": arrays["e"], "alpha": arrays["a"]}), ref)
    for x in everything:
        main_ctx.unregister_host_buffer(x)  # the last holder: the pages are unlocked here
    # unlocked for real: the same call now takes a call-scoped lock (or the scratch path) again
    arrays["s"][:], arrays["e"][:], arrays["a"][:] = s, h["eps_n"], h["alpha"]
    law.evaluate(0.0, 1.0, g, arrays["s"], arrays["t"], {"eps_n": arrays["e"], "alpha": arrays["a"]})
    assert main_ctx.last_host_mode() & (_capi.HOST_TEMP_LOCK | _capi.HOST_BOUNCE)
    assert_same((arrays["s"], arrays["t"], {"eps_n": arrays["e"], "alpha": arrays["a"]}), ref)


@pytest.mark.parametrize("kind,constraint", [("kelvin", "PLANE_STRESS"), ("maxwell", "UNIAXIAL_STRESS"), ("le", "PLANE_STRAIN"), ("kelvin", "UNIAXIAL_STRAIN")])
def test_multi_device_resident_state_low_dimensional_constraints(kind, constraint):
    """the laws the reference implements for all constraints, resident over three device contexts (the chunked pass of
    fcamd_evaluate_resident: 1 / 4 doubles per point instead of 6 / 9) against the single-device resident state"""
    c = fc.StressStrainConstraint[constraint]
    n = 64 * 300 + 11
    law = LAWS[kind](c)
    rng = np.random.default_rng(21)
    gd2, sd = law.geometric_dim**2, law.stress_strain_dim
    one = ResidentState(law, n, placement="torch")
    multi = MultiDeviceResidentState(LAWS[kind](c), n, devices=[0, 0, 0])
    sa, ta, sb, tb = np.zeros(sd * n), np.full(sd * sd * n, np.nan), np.zeros(sd * n), np.full(sd * sd * n, np.nan)
    for inc in range(3):
        for it in range(2):
            g = rng.normal(scale=1e-3, size=gd2 * n)
            one.evaluate_into(float(inc), 0.5, g, sa, ta)
            multi.evaluate_into(float(inc), 0.5, g, sb, tb)
            assert np.array_equal(sa, sb) and np.array_equal(ta, tb), (inc, it)
        one.update()
        multi.update()
        if law.history_dim is not None:
            hc = multi.history_committed
            for k in law.history_dim:
                assert np.array_equal(hc[k], one.history_committed[k].cpu().numpy()), (inc, k)
    multi.close()


def test_multi_handle_error_paths():
    """a device that does not exist, flags of the wrong law, wrong history counts: statuses with messages, no leaks of
    half-built handles (the worker threads of a failed create are joined)"""
    import threading

    before = threading.active_count()  # Python threads only; the C workers are not counted, a hang would be
    with pytest.raises(ValueError, match="out of range"):
        _capi.Multi([0, 99], _capi.VON_MISES_3D, 5, [175000.0, 80769.0, 1200.0, 2500.0, 200.0])
    with pytest.raises(ValueError):
        _capi.Multi([0], _capi.VON_MISES_3D, 5, [1.0, 2.0])  # wrong parameter count
    with pytest.raises(NotImplementedError):
        _capi.Multi([0, 0], _capi.VON_MISES_3D, 3, [175000.0, 80769.0, 1200.0, 2500.0, 200.0])  # PLANE_STRAIN: FULL only
    m = _capi.Multi([0, 0], _capi.VON_MISES_3D, 5, [175000.0, 80769.0, 1200.0, 2500.0, 200.0])
    with pytest.raises(NotImplementedError, match="SPLIT_HISTORY"):
        _capi.MultiState(m, 1000, _capi.EVAL_SPLIT_HISTORY)  # VonMises3D has no 7-double rows
    st = _capi.MultiState(m, 1000)
    a = np.zeros(6000)
    with pytest.raises(AssertionError, match="history fields"):
        st.set(a.ctypes.data, [a.ctypes.data])  # one history array for a law with two fields
    with pytest.raises(ValueError, match="commit before any evaluate"):
        st.commit()
    with pytest.raises(ValueError, match="SPARSE_TANGENT or 0"):
        st.evaluate(0.0, 1.0, np.zeros(9000).ctypes.data, None, None, flags=_capi.EVAL_PACKED_HISTORY)
    m.close()  # destroys the state that is still alive on it
    st.close()  # ... so this is a no-op
    assert threading.active_count() == before
