"""GPU parity tests: the HIP path, called through the C ABI (Python law -> ctypes ->
libfcamd.so), against the CPU oracle on identical inputs and against the committed golden
vectors captured from the imported reference.

Tolerances (BASELINE.json north_star): 1e-10 relative for linear elasticity (and the SLS
laws), 1e-6 relative for the plasticity return mappings.  "relative" = max|a-b| / max|b|
per array.  A stricter regression bound (rounding level) is asserted separately.
"""

import numpy as np
import pytest
from golden_util import load_calls, rel_err

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import fenics_constitutive_amd as fc  # noqa: E402
from oracle import c_oracle as CO  # noqa: E402
from oracle import numpy_oracle as O  # noqa: E402

FULL = fc.StressStrainConstraint.FULL
TOL = {"le": 1e-10, "sls": 1e-10, "pl": 1e-6}
STRICT = {"le": 1e-14, "sls": 1e-14, "pl": 1e-11}

VM_P = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}
SLS_P = {"E0": 42.0, "E1": 10.0, "tau": 10.0, "nu": 0.2}
RS_P = {"mu": 80769.0, "kappa": 175000.0, "y_0": 1200.0, "h": 200.0}
LE_P = {"E": 42.0, "nu": 0.3}


def rs(p):
    return {k: np.array([v]) for k, v in p.items()}


def make_law(kind, p):
    return {
        "linear_elasticity": lambda: fc.LinearElasticityModel(p, FULL),
        "von_mises_3d": lambda: fc.VonMises3D(p),
        "spring_maxwell": lambda: fc.SpringMaxwellModel(p, FULL),
        "spring_kelvin": lambda: fc.SpringKelvinModel(p, FULL),
        "comfe_linear_elasticity": lambda: fc.LinearElasticity3D(rs(p)),
        "comfe_mises_plasticity": lambda: fc.MisesPlasticityLinearHardening3D(rs(p)),
    }[kind]()


CLASS = {"linear_elasticity": "le", "comfe_linear_elasticity": "le", "spring_maxwell": "sls",
         "spring_kelvin": "sls", "von_mises_3d": "pl", "comfe_mises_plasticity": "pl"}


def run_host(law, del_t, g, s, t, h):
    law.evaluate(0.0, del_t, g, s, t, h)
    return s, t, h


def run_device(law, del_t, g, s, t, h):
    gd, sd = torch.from_numpy(g).cuda(), torch.from_numpy(s).cuda()
    td = None if t is None else torch.full((t.size,), float("nan"), dtype=torch.float64, device="cuda")
    hd = None if h is None else {k: torch.from_numpy(v).cuda() for k, v in h.items()}
    law.evaluate(0.0, del_t, gd, sd, td, hd)
    torch.cuda.synchronize()
    return (sd.cpu().numpy(), None if td is None else td.cpu().numpy(),
            None if hd is None else {k: v.cpu().numpy() for k, v in hd.items()})


def compare(got, ref, tol, what=""):
    s, t, h = got
    s_ref, t_ref, h_ref = ref
    assert rel_err(s, s_ref) <= tol, f"{what} stress {rel_err(s, s_ref):.3e}"
    if t is not None:
        assert not np.isnan(t).any(), f"{what} tangent has unwritten entries"
        assert rel_err(t, t_ref) <= tol, f"{what} tangent {rel_err(t, t_ref):.3e}"
    if h_ref is not None:
        for k in h_ref:
            assert rel_err(h[k], h_ref[k]) <= tol, f"{what} history[{k}] {rel_err(h[k], h_ref[k]):.3e}"


# ---------------------------------------------------------------------------------------
# golden vectors captured from the reference
# ---------------------------------------------------------------------------------------
GOLDEN = [(f, k, c) for f, k in [("linear_elasticity.npz", "linear_elasticity"), ("von_mises_3d.npz", "von_mises_3d"),
                                 ("spring_maxwell.npz", "spring_maxwell"), ("spring_kelvin.npz", "spring_kelvin"),
                                 ("random_parameters_linear_elasticity.npz", "linear_elasticity"),
                                 ("random_parameters_von_mises_3d.npz", "von_mises_3d"),
                                 ("random_parameters_spring_maxwell.npz", "spring_maxwell"),
                                 ("random_parameters_spring_kelvin.npz", "spring_kelvin")]
          for c in load_calls(f)]


@pytest.mark.parametrize("path", ["host", "device"])
@pytest.mark.parametrize("fname,kind,c", GOLDEN, ids=[f"{k}-{c.name}" for _, k, c in GOLDEN])
def test_golden(fname, kind, c, path):
    law = make_law(kind, c.params)
    s, t, h = c.fresh()
    got = (run_host if path == "host" else run_device)(law, c.del_t, c.grad.copy(), s, t, h)
    compare(got, (c.stress_out, c.tangent_out, c.hist_out), TOL[CLASS[kind]], f"{kind}/{c.name}/{path}")
    compare(got, (c.stress_out, c.tangent_out, c.hist_out), STRICT[CLASS[kind]], f"strict {kind}/{c.name}/{path}")


def test_le_sls_bit_pattern_report():
    """Bit-pattern goal: fraction of outputs bit-identical to the reference's
    NumPy results.  Reported; the hard gate is 1e-10."""
    for fname, kind in [("linear_elasticity.npz", "linear_elasticity"), ("spring_maxwell.npz", "spring_maxwell"),
                        ("spring_kelvin.npz", "spring_kelvin")]:
        eq = tot = 0
        for c in load_calls(fname):
            s, t, h = c.fresh()
            got = run_device(make_law(kind, c.params), c.del_t, c.grad.copy(), s, t, h)
            eq += int(np.sum(got[0] == c.stress_out)) + int(np.sum(got[1] == c.tangent_out))
            tot += got[0].size + got[1].size
        print(f"{kind}: {eq}/{tot} outputs bit-identical to the reference ({100.0 * eq / tot:.4f} %)")
        assert eq / tot > 0.999


# ---------------------------------------------------------------------------------------
# seeded random inputs against the oracle, ragged and edge sizes
# ---------------------------------------------------------------------------------------
def random_case(kind, n, seed, gscale=None):
    rng = np.random.default_rng(seed)
    p = {"linear_elasticity": LE_P, "von_mises_3d": VM_P, "spring_maxwell": SLS_P, "spring_kelvin": SLS_P,
         "comfe_linear_elasticity": {"mu": 16.0, "kappa": 35.0}, "comfe_mises_plasticity": RS_P}[kind]
    if gscale is None:
        gscale = 10 ** rng.uniform(-4, -2, size=n) if CLASS[kind] == "pl" else np.full(n, 1e-3)
    g = rng.normal(size=9 * n) * np.repeat(gscale, 9)
    s = rng.normal(scale=30.0 if CLASS[kind] == "pl" else 1.0, size=6 * n)
    h = None
    if kind == "von_mises_3d":
        h = {"eps_n": rng.normal(scale=1e-3, size=6 * n), "alpha": rng.uniform(0, 0.02, size=n)}
    elif kind in ("spring_maxwell", "spring_kelvin"):
        h = {"strain_visco": rng.normal(scale=1e-4, size=6 * n), "strain": rng.normal(scale=1e-3, size=6 * n)}
    elif kind == "comfe_mises_plasticity":
        hh = rng.normal(scale=1e-3, size=7 * n)
        hh.reshape(-1, 7)[:, 0] = rng.uniform(0, 0.02, size=n)
        h = {"history": hh}
    return p, g, s, h


def oracle_run(kind, p, del_t, g, s, h, mod=O):
    s, t = s.copy(), np.zeros(36 * (g.size // 9))
    h = None if h is None else {k: v.copy() for k, v in h.items()}
    mod.MODELS[kind](p, 0.0, del_t, g, s, t, h)
    return s, t, h


KINDS = list(CLASS)


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 127, 128, 1000, 4097])
@pytest.mark.parametrize("kind", KINDS)
def test_sizes_host_and_device(kind, n):
    p, g, s, h = random_case(kind, n, seed=n + 17)
    ref = oracle_run(kind, p, 2.0, g, s, h)
    law = make_law(kind, p)
    tol = TOL[CLASS[kind]]
    hc = None if h is None else {k: v.copy() for k, v in h.items()}
    compare(run_host(law, 2.0, g, s.copy(), np.full(36 * n, np.nan), hc), ref, tol, f"{kind} n={n} host")
    compare(run_device(law, 2.0, g, s.copy(), np.full(36 * n, np.nan), h), ref, tol, f"{kind} n={n} device")


@pytest.mark.parametrize("kind", KINDS)
def test_large_random_vs_c_oracle(kind):
    n = 300_001
    p, g, s, h = random_case(kind, n, seed=5)
    ref = oracle_run(kind, p, 0.5, g, s, h, mod=CO)
    law = make_law(kind, p)
    got = run_device(law, 0.5, g, s.copy(), np.full(36 * n, np.nan), h)
    compare(got, ref, TOL[CLASS[kind]], kind)
    compare(got, ref, STRICT[CLASS[kind]], "strict " + kind)
    if CLASS[kind] == "pl":
        st = law.device_stats()
        assert 0 < st.n_plastic < n


@pytest.mark.parametrize("gscale,expect", [(1e-5, "elastic"), (3e-2, "plastic")])
@pytest.mark.parametrize("kind", ["von_mises_3d", "comfe_mises_plasticity"])
def test_all_elastic_and_all_plastic_tiles(kind, gscale, expect):
    n = 64 * 37
    p, g, s, h = random_case(kind, n, seed=9, gscale=np.full(n, gscale))
    s[:] = 0.0
    ref = oracle_run(kind, p, 1.0, g, s, h)
    law = make_law(kind, p)
    got = run_device(law, 1.0, g, s.copy(), np.full(36 * n, np.nan), h)
    compare(got, ref, TOL["pl"], f"{kind} {expect}")
    st = law.device_stats()
    n_pl_ref = O.MODELS[kind](p, 0.0, 1.0, g, s.copy(), np.zeros(36 * n), {k: v.copy() for k, v in h.items()})
    n_pl_ref = n_pl_ref[0] if isinstance(n_pl_ref, tuple) else n_pl_ref
    assert st.n_plastic == n_pl_ref
    assert st.n_plastic == 0 if expect == "elastic" else st.n_plastic > 0.99 * n
    if expect == "elastic":  # history must be untouched bit for bit
        for k in h:
            assert np.array_equal(got[2][k], h[k])
    elif kind == "von_mises_3d":
        assert 3 * st.n_plastic <= st.n_newton_iters <= 12 * st.n_plastic


def test_zero_strain_is_identity_for_stress():
    n = 130
    for kind in ("linear_elasticity", "von_mises_3d", "comfe_linear_elasticity"):
        p, g, s, h = random_case(kind, n, seed=3)
        g[:] = 0.0
        s *= 0.01
        got = run_device(make_law(kind, p), 1.0, g, s.copy(), np.full(36 * n, np.nan), h)
        assert np.array_equal(got[0], s), kind


def test_tangent_none_is_accepted():
    """The Rust entry takes tangent=None (comfe-rs/src/interfaces.rs:383-394)."""
    n = 257
    for kind in ("comfe_linear_elasticity", "comfe_mises_plasticity", "linear_elasticity"):
        p, g, s, h = random_case(kind, n, seed=4)
        ref = oracle_run(kind, p, 1.0, g, s, h)
        got = run_device(make_law(kind, p), 1.0, g, s.copy(), None, h)
        compare(got, (ref[0], None, ref[2]), TOL[CLASS[kind]], kind)


@pytest.mark.parametrize("kind", ["von_mises_3d", "spring_maxwell", "spring_kelvin", "comfe_mises_plasticity", "linear_elasticity"])
def test_out_of_place_equals_in_place(kind):
    """evaluate_from(committed -> trial) == copy + in-place evaluate (the reference protocol,
    solver/_lawonsubmesh.py:58-61, solver/_history.py:64-79); committed state is not modified."""
    n = 64 * 9 + 5
    p, g, s, h = random_case(kind, n, seed=8)
    law = make_law(kind, p)
    ref = run_device(law, 2.0, g, s.copy(), np.zeros(36 * n), None if h is None else {k: v.copy() for k, v in h.items()})
    gd, sp = torch.from_numpy(g).cuda(), torch.from_numpy(s).cuda()
    sc, td = torch.zeros_like(sp), torch.zeros(36 * n, dtype=torch.float64, device="cuda")
    hp = None if h is None else {k: torch.from_numpy(v).cuda() for k, v in h.items()}
    hc = None if h is None else {k: torch.zeros_like(v) for k, v in hp.items()}
    law.evaluate_from(0.0, 2.0, gd, sp, sc, td, hp, hc)
    torch.cuda.synchronize()
    assert np.array_equal(sc.cpu().numpy(), ref[0]) and np.array_equal(td.cpu().numpy(), ref[1])
    assert np.array_equal(sp.cpu().numpy(), s)
    if h is not None:
        for k in h:
            assert np.array_equal(hc[k].cpu().numpy(), ref[2][k]), k
            assert np.array_equal(hp[k].cpu().numpy(), h[k]), k


def test_multistep_protocol_matches_reference_sequence():
    """Replay the committed mixed sequence of the VonMises3D fixture end to end on the GPU."""
    calls = {c.name: c for c in load_calls("von_mises_3d.npz")}
    law = make_law("von_mises_3d", calls["mixed_step0_iter1"].params)
    c0 = calls["mixed_step0_iter0"]
    s = torch.from_numpy(c0.stress_in).cuda()
    h = {k: torch.from_numpy(v).cuda() for k, v in c0.hist_in.items()}
    for k in range(4):
        for it in (0, 1):
            c = calls[f"mixed_step{k}_iter{it}"]
            st, ht = s.clone(), {kk: v.clone() for kk, v in h.items()}  # trial <- committed
            t = torch.empty(36 * c.n, dtype=torch.float64, device="cuda")
            law.evaluate(0.0, c.del_t, torch.from_numpy(c.grad).cuda(), st, t, ht)
            assert rel_err(st.cpu().numpy(), c.stress_out) <= 1e-6
            assert rel_err(t.cpu().numpy(), c.tangent_out) <= 1e-6
        s, h = st, ht  # commit (problem.update())
    assert rel_err(h["alpha"].cpu().numpy(), calls["mixed_step3_iter1"].hist_out["alpha"]) <= 1e-6


def test_strain_from_grad_u():
    # known answer of tests/models/test_conversions.py:29-44 and the golden array
    g = np.arange(1.0, 10.0)
    e = fc.strain_from_grad_u(g, FULL)
    assert np.allclose(e, [1.0, 5.0, 9.0, 0.5 * 6 * 2**0.5, 0.5 * 10 * 2**0.5, 0.5 * 14 * 2**0.5])
    z = np.load(__import__("os").path.join(__import__("golden_util").GOLDEN, "strain_from_grad_u.npz"))
    assert np.array_equal(fc.strain_from_grad_u(z["grad"], FULL), z["strain"])
    big = np.random.default_rng(1).normal(size=9 * 100_003)
    assert np.array_equal(fc.strain_from_grad_u(big, FULL), O.strain_from_grad_u_full(big))


# ---------------------------------------------------------------------------------------
# error behaviour of the reference
# ---------------------------------------------------------------------------------------
def test_error_conventions():
    n = 10
    g, s, t = np.zeros(9 * n), np.zeros(6 * n), np.zeros(36 * n)
    sls = fc.SpringMaxwellModel(SLS_P, FULL)
    h = {"strain_visco": np.zeros(6 * n), "strain": np.zeros(6 * n)}
    with pytest.raises(ValueError, match="history must not be None"):
        sls.evaluate(0, 1.0, g, s, t, None)
    with pytest.raises(AssertionError):
        sls.evaluate(0, 0.0, g, s, t, h)
    with pytest.raises(AssertionError):
        fc.SpringKelvinModel(SLS_P, FULL).evaluate(0, -1.0, g, s, t, h)
    le = fc.LinearElasticityModel(LE_P, FULL)
    with pytest.raises(AssertionError):
        le.evaluate(0, 1.0, g, s[:-6], t, None)
    with pytest.raises(AssertionError):
        le.evaluate(0, 1.0, g, s, t[:-36], None)
    with pytest.raises(TypeError):
        le.evaluate(0, 1.0, g.astype(np.float32), s, t, None)
    with pytest.raises(TypeError):
        le.evaluate(0, 1.0, np.zeros(18 * n)[::2], s, t, None)
    with pytest.raises(ValueError):
        fc.VonMises3D(VM_P).evaluate(0, 1.0, g, s, t, None)


def test_newton_nonconvergence_raises_runtime_error():
    from test_oracle_c import NONCONVERGING, nonconverging_inputs

    law = fc.VonMises3D(NONCONVERGING)
    g, s, t, h = nonconverging_inputs(70)
    with pytest.raises(RuntimeError, match="did not converge"):
        law.evaluate(0, 1.0, g, s, t, h)
    assert law._handle(0).last_stats().n_nonconverged == 70
    # device path: asynchronous, reported by device_stats()
    gd, sd = torch.from_numpy(g).cuda(), torch.zeros(6 * 70, dtype=torch.float64, device="cuda")
    td = torch.zeros(36 * 70, dtype=torch.float64, device="cuda")
    hd = {"eps_n": torch.zeros(6 * 70, dtype=torch.float64, device="cuda"), "alpha": torch.zeros(70, dtype=torch.float64, device="cuda")}
    law.evaluate(0, 1.0, gd, sd, td, hd)
    with pytest.raises(RuntimeError, match="did not converge"):
        law.device_stats()


def test_auto_pin_host_arrays():
    """Opt-in auto_pin page-locks the caller's arrays at first sight and keeps them alive; results are
    unchanged while pinned and after unpin_arrays()."""
    n = 400_000
    p, g, s, h = random_case("von_mises_3d", n, seed=21)
    ref = oracle_run("von_mises_3d", p, 1.0, g, s, h, mod=CO)
    law = make_law("von_mises_3d", p)
    assert law.auto_pin is False
    law.auto_pin = True
    t = np.full(36 * n, np.nan)
    sc, hc = s.copy(), {k: v.copy() for k, v in h.items()}   # stable buffers, refilled per call
    for call in range(3):
        sc[:] = s
        for k in h:
            hc[k][:] = h[k]
        law.evaluate(0.0, 1.0, g, sc, t, hc)
        compare((sc, t, hc), ref, TOL["pl"], f"auto_pin call {call}")
    assert sum(a is not None for a in law._pinned.values()) == 5
    law.unpin_arrays()
    sc[:] = s
    for k in h:
        hc[k][:] = h[k]
    law.auto_pin = False
    law.evaluate(0.0, 1.0, g, sc, t, hc)
    compare((sc, t, hc), ref, TOL["pl"], "after unpin")


@pytest.mark.parametrize("kind", ["linear_elasticity", "spring_maxwell", "spring_kelvin", "von_mises_3d"])
def test_special_values_propagate_like_the_reference(kind):
    """NaN / Inf / signed zero / subnormal / huge inputs: the same points become NaN or Inf as in the
    oracle (the reference has no guards: NaN trial states fall into the elastic branch), every other
    point is unaffected."""
    n = 64 * 3 + 9
    p, g, s, h = random_case(kind, n, seed=77)
    specials = [np.nan, np.inf, -np.inf, -0.0, 5e-324, 1e-310, 1e300, -1e300]
    rng = np.random.default_rng(1)
    for k, v in enumerate(specials):
        g[9 * (7 * k + 3) + rng.integers(0, 9)] = v          # one special per chosen point, in the gradient
        s[6 * (7 * k + 70) + rng.integers(0, 6)] = v         # ... and in the stress of other points
    with np.errstate(all="ignore"):
        ref = oracle_run(kind, p, 1.5, g, s, h)
    got = run_device(make_law(kind, p), 1.5, g, s.copy(), np.full(36 * n, np.nan), h)
    for name, a, b in [("stress", got[0], ref[0]), ("tangent", got[1], ref[1])] + (
            [] if h is None else [(k, got[2][k], ref[2][k]) for k in h]):
        assert np.array_equal(np.isnan(a), np.isnan(b)), f"{kind} {name}: NaN pattern differs"
        assert np.array_equal(np.isinf(a), np.isinf(b)), f"{kind} {name}: Inf pattern differs"
        ok = np.isfinite(b)
        assert rel_err(a[ok], b[ok]) <= TOL[CLASS[kind]], f"{kind} {name}"


@pytest.mark.parametrize("kind", ["von_mises_3d", "spring_kelvin"])
def test_host_path_many_chunks_ragged(kind):
    """ndarray path over several staging chunks (4 slots in flight) with a ragged tail."""
    n = 3 * (1 << 19) + 4 * 64 + 37
    p, g, s, h = random_case(kind, n, seed=31)
    ref = oracle_run(kind, p, 0.9, g, s, h, mod=CO)
    law = make_law(kind, p)
    got = run_host(law, 0.9, g, s.copy(), np.full(36 * n, np.nan), {k: v.copy() for k, v in h.items()})
    compare(got, ref, TOL[CLASS[kind]], kind)
    compare(got, ref, STRICT[CLASS[kind]], "strict " + kind)
    if kind == "von_mises_3d":
        assert law.last_stats.n_plastic == int(np.sum(ref[2]["alpha"] > h["alpha"]))


def test_strain_from_grad_u_low_dimensional():
    """tests/models/test_conversions.py:14-28 and the oracle, NumPy in / out and device tensors."""
    C = fc.StressStrainConstraint
    assert np.array_equal(fc.strain_from_grad_u(np.array([[1.0]]), C.UNIAXIAL_STRAIN), [1.0])
    assert np.array_equal(fc.strain_from_grad_u(np.array([[1.0]]), C.UNIAXIAL_STRESS), [1.0])
    g = np.array([[1.0, 2.0], [3.0, 4.0]])
    for c in (C.PLANE_STRAIN, C.PLANE_STRESS):
        assert np.allclose(fc.strain_from_grad_u(g, c), [1.0, 4.0, 0.0, 0.5 * 5.0 * 2**0.5], rtol=1e-15)
    rng = np.random.default_rng(3)
    for c in C:
        if c.name == "FULL":
            continue
        gg = rng.normal(size=c.geometric_dim**2 * 1003)
        ref = O.strain_from_grad_u(gg, c.name)
        assert np.array_equal(fc.strain_from_grad_u(gg, c), ref)
        dev = fc.strain_from_grad_u(torch.from_numpy(gg).cuda(), c)
        assert dev.is_cuda and np.array_equal(dev.cpu().numpy(), ref)


def test_one_law_object_from_two_threads():
    """Contexts (streams, staging buffers, registry of page-locked ranges) are per thread; a law object used
    from two Python threads at once gets one C handle per thread and both get the right numbers."""
    import threading

    n = 200_000
    p, g, s, h = random_case("von_mises_3d", n, seed=6)
    ref = oracle_run("von_mises_3d", p, 1.0, g, s, h, mod=CO)
    law = make_law("von_mises_3d", p)
    results, errors = {}, []

    def work(tag):
        try:
            for _ in range(3):
                sc, t, hc = s.copy(), np.full(36 * n, np.nan), {k: v.copy() for k, v in h.items()}
                law.evaluate(0.0, 1.0, g, sc, t, hc)
            results[tag] = (sc, t, hc)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    assert law.n_handles_created == 2  # one C handle (and context) per thread, released with the thread
    for tag in (0, 1):
        compare(results[tag], ref, STRICT["pl"], f"thread {tag}")
