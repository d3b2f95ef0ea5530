"""Device-resident increment state replays the reference protocol -- several
Newton re-evaluations per increment, commit, next increment -- and must reproduce the golden
multi-step sequences captured from the reference."""

import numpy as np
import pytest
from golden_util import load_calls, rel_err

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import fenics_constitutive_amd as fc  # noqa: E402
from fenics_constitutive_amd.resident import ResidentState  # noqa: E402

FULL = fc.StressStrainConstraint.FULL


def test_von_mises_mixed_sequence():
    calls = {c.name: c for c in load_calls("von_mises_3d.npz")}
    c0 = calls["mixed_step0_iter0"]
    law = fc.VonMises3D(c0.params)
    st = ResidentState(law, c0.n, stress0=c0.stress_in, history0=c0.hist_in)
    for k in range(4):
        for it in (0, 1):
            c = calls[f"mixed_step{k}_iter{it}"]
            st.evaluate(0.0, c.del_t, c.grad)  # NumPy grad: the only PCIe upload
            s, t = np.empty(6 * c.n), np.empty(36 * c.n)
            h = {"eps_n": np.empty(6 * c.n), "alpha": np.empty(c.n)}
            st.download(s, t, h)
            assert rel_err(s, c.stress_out) <= 1e-6 and rel_err(t, c.tangent_out) <= 1e-6
            assert rel_err(s, c.stress_out) <= 1e-11
            for key in h:
                assert rel_err(h[key], c.hist_out[key]) <= 1e-6
            # the committed copy is untouched by trial evaluations
            assert np.array_equal(st.stress_committed.cpu().numpy(), c.stress_in)
        st.check()
        st.update()
    assert np.array_equal(st.stress_committed.cpu().numpy(), s)


@pytest.mark.parametrize("fname,cls", [("spring_maxwell.npz", "SpringMaxwellModel"), ("spring_kelvin.npz", "SpringKelvinModel")])
def test_sls_sequence(fname, cls):
    calls = {c.name: c for c in load_calls(fname)}
    c0 = calls["step0_iter0"]
    law = getattr(fc, cls)(c0.params, FULL)
    st = ResidentState(law, c0.n, stress0=c0.stress_in, history0=c0.hist_in)
    for k in range(5):
        for it in (0, 1):
            c = calls[f"step{k}_iter{it}"]
            st.evaluate(0.0, c.del_t, torch.from_numpy(c.grad).cuda())  # device grad: zero copies
            assert rel_err(st.stress.cpu().numpy(), c.stress_out) <= 1e-10
            assert rel_err(st.tangent.cpu().numpy(), c.tangent_out) <= 1e-10
            for key in c.hist_out:
                assert rel_err(st.history[key].cpu().numpy(), c.hist_out[key]) <= 1e-10
        st.update()


def test_update_without_evaluate_raises():
    st = ResidentState(fc.LinearElasticityModel({"E": 1.0, "nu": 0.3}, FULL), 10)
    with pytest.raises(RuntimeError):
        st.update()
    st.evaluate(0, 1, np.zeros(90))
    st.update()


def _sparse_case(law_name, n, rng):
    """(law, stress0, history0, grad generator) with plastic sets that vary from call to call."""
    if law_name == "VonMises3D":
        law = fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0})
        s0 = rng.normal(scale=30.0, size=6 * n)
        h0 = {"eps_n": rng.normal(scale=1e-3, size=6 * n), "alpha": rng.uniform(0, 0.02, size=n)}
    else:
        h = rng.normal(scale=1e-3, size=7 * n)
        h.reshape(-1, 7)[:, 0] = rng.uniform(0, 0.02, size=n)
        h0 = {"history": h}
        if law_name == "MisesPlasticityLinearHardening3D":
            law = fc.MisesPlasticityLinearHardening3D(
                {k: np.array([v]) for k, v in {"mu": 80769.0, "kappa": 175000.0, "y_0": 1200.0, "h": 200.0}.items()})
            s0 = rng.normal(scale=30.0, size=6 * n)
        else:
            p = {"mu": 80769.0, "kappa": 175000.0, "a": 100.0, "b": 0.05, "b_flow": 0.02}
            if law_name == "DruckerPragerHyperbolic3D":
                p = {"mu": p["mu"], "kappa": p["kappa"], "a": p["a"], "b": p["b"], "d": 40.0, "b_flow": p["b_flow"]}
            law = getattr(fc, law_name)({k: np.array([v]) for k, v in p.items()})
            s0 = rng.normal(scale=30.0, size=6 * n)
            s0.reshape(-1, 6)[:, :3] -= 1000.0  # compressive prestress keeps the classic surface off its tip

    # plastic-strain rows: zones of 200 points that have never been plastic (+0.0 rows: what the packed layout of a resident
    # state leaves out), zones in which every point has been, and mixed zones -- EVER masks from empty to full
    zone = (np.arange(n) // 200) % 4
    virgin = (zone == 0) | ((zone >= 2) & (rng.random(n) < 0.5))
    key, w = ("eps_n", 6) if law_name == "VonMises3D" else ("history", 7)
    h0[key].reshape(-1, w)[virgin, w - 6:] = 0.0

    hi = -2.3 if law_name.startswith("Drucker") else -1.6  # log10 of the largest strain scale

    def grad(all_elastic, zoned):
        # random per-point scale (every tile mixed) or plastic zones of 256 points (many tiles all elastic)
        if zoned:
            scale = np.repeat(10 ** rng.uniform(hi - 3.2, hi, size=(n + 255) // 256), 256)[:n]
        else:
            scale = 10 ** rng.uniform(hi - 2.7, hi, size=n)
        g = (rng.normal(size=9 * n) * np.repeat(scale * (0.0 if all_elastic else 1.0), 9)).reshape(-1, 9)
        if law_name.startswith("Drucker"):
            g[:, [0, 4, 8]] -= (0.95 * g[:, [0, 4, 8]].sum(axis=1) / 3.0)[:, None]  # mostly isochoric
        return torch.from_numpy(g.reshape(-1).copy()).cuda()

    return law, s0, h0, grad


@pytest.mark.parametrize("law_name", ["VonMises3D", "MisesPlasticityLinearHardening3D", "DruckerPrager3D",
                                      "DruckerPragerHyperbolic3D"])
@pytest.mark.parametrize("n", [64 * 40 + 17, 5000])
@pytest.mark.parametrize("split", [True, False])
def test_sparse_history_equals_full_history(n, law_name, split):
    """The sparse trial-history protocol (VonMises3D: only plastic / formerly plastic points touch
    eps_n; comfe-rs laws: only tiles with such points are written) must give the same trial state as
    the full out-of-place evaluate at every Newton iteration of every increment, with plastic sets
    that grow, shrink and move, across pointer-swap commits.  ``split``: the comfe-rs laws with their history kept as
    [scalar (n), eps_p rows (6 n)] inside the state (FCAMD_EVAL_SPLIT_HISTORY, the default) or as the reference's
    7-double rows; one host-assembler pass (fcamd_evaluate_resident) in between."""
    if law_name == "VonMises3D" and not split:
        pytest.skip("VonMises3D has no 7-double history rows")
    rng = np.random.default_rng(n)
    law, s0, h0, grad = _sparse_case(law_name, n, rng)
    sp = ResidentState(law, n, stress0=s0, history0=h0, sparse_history=True, split_history=split)
    fu = ResidentState(law, n, stress0=s0, history0=h0, sparse_history=False)
    assert sp._mask is not None and fu._mask is None
    assert sp._split == (split and law_name != "VonMises3D") and not fu._split
    assert sp._sparse_tangent and not fu._sparse_tangent  # sp also runs the sparse-tangent protocol
    n_plastic = []
    sh, th = np.empty(6 * n), np.empty(36 * n)
    for inc in range(5):
        for it in range(3):
            g = grad(all_elastic=(inc == 2 and it == 1), zoned=(inc % 2 == 1))
            if (inc, it) in ((1, 2), (3, 0)):
                sp.evaluate_into(0.0, 1.0, g.cpu().numpy(), sh, th)
                fu.evaluate(0.0, 1.0, g)
                n_plastic.append(int(fu.check().n_plastic))
                assert np.array_equal(sh, fu.stress.cpu().numpy()) and np.array_equal(th, fu.tangent.cpu().numpy()), (inc, it)
                assert torch.equal(sp.stress, fu.stress)
                for k in h0:
                    assert torch.equal(sp.history[k], fu.history[k]), (inc, it, k)
                continue
            sp.evaluate(0.0, 1.0, g)
            fu.evaluate(0.0, 1.0, g)
            n_plastic.append(int(fu.check().n_plastic))
            assert torch.equal(sp.stress, fu.stress) and torch.equal(sp.tangent, fu.tangent)
            for k in h0:
                assert torch.equal(sp.history[k], fu.history[k]), (inc, it, k)
                assert torch.equal(sp.history_committed[k], fu.history_committed[k]), (inc, it, k)
        sp.check()
        sp.update()
        fu.update()
    assert max(n_plastic) > 0.1 * n and min(n_plastic) < 0.5 * max(n_plastic)  # the sets really changed


@pytest.mark.parametrize("path", ["scratch", "lock", "chunks"])
def test_evaluate_into_replays_golden_sequence(path):
    """fcamd_evaluate_resident: NumPy grad in, NumPy stress/tangent out, state on the device, on each data path of the
    pageable host arrays: through the page-locked scratch (the default of small calls), page-locked for the call with one
    launch on them ("bounce_max" = 0), and page-locked with the chunked DMA pipeline ("zero_copy" = 0, 128-point chunks:
    the 1000+ points go through many chunks and all four slots)."""
    from fenics_constitutive_amd import _capi

    ctx = _capi.get_context(_capi.default_device())
    options = {"scratch": {"bounce_max": 1 << 30}, "lock": {"bounce_max": 0}, "chunks": {"bounce_max": 0, "zero_copy": 0, "host_chunk": 128}}[path]
    saved = {k: ctx.get_option(k) for k in options}
    for k, v in options.items():
        ctx.set_option(k, v)
    try:
        calls = {c.name: c for c in load_calls("von_mises_3d.npz")}
        c0 = calls["mixed_step0_iter0"]
        law = fc.VonMises3D(c0.params)
        st = ResidentState(law, c0.n, stress0=c0.stress_in, history0=c0.hist_in)
        s, t = np.empty(6 * c0.n), np.empty(36 * c0.n)
        expected = {"scratch": _capi.HOST_BOUNCE, "lock": _capi.HOST_TEMP_LOCK | _capi.HOST_ZERO_COPY_IN | _capi.HOST_ZERO_COPY_OUT,
                    "chunks": _capi.HOST_TEMP_LOCK}[path]
        for k in range(4):
            for it in (0, 1):
                c = calls[f"mixed_step{k}_iter{it}"]
                stats = st.evaluate_into(0.0, c.del_t, c.grad, s, t)
                assert ctx.last_host_mode() == expected
                assert rel_err(s, c.stress_out) <= 1e-11 and rel_err(t, c.tangent_out) <= 1e-6
                for key in c.hist_out:
                    assert rel_err(st.history[key].cpu().numpy(), c.hist_out[key]) <= 1e-6
                assert np.array_equal(st.stress_committed.cpu().numpy(), c.stress_in)
                assert stats.n_plastic > 0
            st.update()
    finally:
        for k, v in saved.items():
            ctx.set_option(k, v)
    assert st._grad is None and st._tangent is None  # no n-sized gradient / tangent on the device


def test_evaluate_into_equals_device_evaluate_sls():
    calls = {c.name: c for c in load_calls("spring_kelvin.npz")}
    c0 = calls["step0_iter0"]
    law = fc.SpringKelvinModel(c0.params, FULL)
    a = ResidentState(law, c0.n, stress0=c0.stress_in, history0=c0.hist_in)
    b = ResidentState(law, c0.n, stress0=c0.stress_in, history0=c0.hist_in)
    s, t = np.empty(6 * c0.n), np.empty(36 * c0.n)
    for k in range(5):
        c = calls[f"step{k}_iter1"]
        a.evaluate(0.0, c.del_t, c.grad)
        b.evaluate_into(0.0, c.del_t, c.grad, s, t)
        assert np.array_equal(a.stress.cpu().numpy(), s) and np.array_equal(a.tangent.cpu().numpy(), t)
        for key in a.history:
            assert np.array_equal(a.history[key].cpu().numpy(), b.history[key].cpu().numpy())
        a.update()
        b.update()
    with pytest.raises(AssertionError):
        b.evaluate_into(0.0, 1.0, c.grad[:-9], s, t)
    with pytest.raises(AssertionError):  # SLS: del_t must be positive
        b.evaluate_into(0.0, 0.0, c.grad, s, t)


@pytest.mark.parametrize("cls", ["LinearElasticityModel", "SpringKelvinModel", "SpringMaxwellModel"])
def test_constant_tangent_is_written_once_per_del_t(cls):
    """LE / SLS: the device tangent array keeps the (point-independent) tangent of the current del_t
    and is not rewritten; results are bit-identical to the state that rewrites it every call."""
    n = 64 * 9 + 5
    rng = np.random.default_rng(5)
    params = {"E": 42.0, "nu": 0.3} if cls == "LinearElasticityModel" else {"E0": 42.0, "E1": 10.0, "tau": 10.0, "nu": 0.2}
    law = getattr(fc, cls)(params, FULL)
    a = ResidentState(law, n)
    b = ResidentState(law, n, reuse_constant_tangent=False)
    assert a._const_tangent and not b._const_tangent
    for k, dt in enumerate([2.0, 2.0, 2.0, 0.5, 0.5, 2.0]):
        g = rng.normal(scale=1e-3, size=9 * n)
        a.evaluate(0.0, dt, g)
        b.evaluate(0.0, dt, g)
        assert torch.equal(a.stress, b.stress) and torch.equal(a.tangent, b.tangent), (k, dt)
        if k == 1:  # the second call with the same del_t must not have touched the array
            a.tangent[:36].fill_(-1.0)
            a.evaluate(0.0, dt, g)
            assert bool((a.tangent[:36] == -1.0).all())
            a._tangent_key = None  # invalidate: the next call rewrites it
        a.update()
        b.update()
    # the plasticity laws never skip
    vm = ResidentState(fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}), n)
    assert not vm._const_tangent


@pytest.mark.parametrize("n", [0, 1, 63, 65])
def test_evaluate_into_tiny_sizes(n):
    from test_gpu_parity import oracle_run, random_case

    p, g, s0, h0 = random_case("von_mises_3d", n, seed=n + 1)
    law = fc.VonMises3D(p)
    st = ResidentState(law, n, stress0=s0, history0=h0)
    s, t = np.full(6 * n, np.nan), np.full(36 * n, np.nan)
    st.evaluate_into(0.0, 1.0, g, s, t)
    ref = oracle_run("von_mises_3d", p, 1.0, g, s0, h0)
    if n:
        assert rel_err(s, ref[0]) <= 1e-9 and rel_err(t, ref[1]) <= 1e-6
        st.update()
        assert rel_err(st.history_committed["alpha"].cpu().numpy(), ref[2]["alpha"]) <= 1e-6


def test_tune_placement_keeps_results():
    """ResidentState.tune_placement picks the tangent allocation by timing; numbers are unaffected."""
    from fenics_constitutive_amd.placement import fastest_allocation
    from oracle import c_oracle as CO
    from test_gpu_parity import LE_P, make_law, oracle_run, random_case

    n = 200_000
    p, g, s, h = random_case("von_mises_3d", n, seed=77)
    ref = oracle_run("von_mises_3d", p, 1.0, g, s, h, mod=CO)
    for kind, pp in (("von_mises_3d", p), ("linear_elasticity", LE_P)):
        law = make_law(kind, pp)
        hh = h if kind == "von_mises_3d" else None
        a = ResidentState(law, n, stress0=s, history0=hh)
        b = ResidentState(law, n, stress0=s, history0=hh)
        a.evaluate(0.0, 1.0, g)
        info = b.tune_placement(0.0, 1.0, g, tries=3)
        assert len(info["candidate_ms"]) == 3 and 0 <= info["chosen"] < 3 and min(info["candidate_ms"]) > 0
        torch.cuda.synchronize()
        assert torch.equal(a.stress, b.stress) and torch.equal(a.tangent, b.tangent)
        if hh is not None:
            for k in hh:
                assert torch.equal(a.history[k], b.history[k])
            assert rel_err(b.tangent.cpu().numpy(), ref[1]) <= 1e-11
        b.evaluate(0.0, 1.0, g)  # constant-tangent laws: the chosen array holds the tangent
        torch.cuda.synchronize()
        assert torch.equal(a.tangent, b.tangent)
    # the helper alone
    t, info = fastest_allocation(1 << 20, lambda x: x.fill_(1.0), tries=2)
    assert t.numel() == 1 << 20 and t.dtype == torch.float64 and len(info["candidate_ms"]) == 2


def test_evaluate_into_downloads_a_constant_tangent_once():
    """LE / SLS host assembler: the caller's tangent array is written when it is first seen and again only
    when del_t (SLS) or the array changes; reuse_constant_tangent=False rewrites it on every call."""
    from test_gpu_parity import LE_P, SLS_P, make_law, random_case

    n = 3000
    _, g, s, _ = random_case("linear_elasticity", n, seed=2)
    for kind, p in (("linear_elasticity", LE_P), ("spring_maxwell", SLS_P)):
        law = make_law(kind, p)
        h = None if kind == "linear_elasticity" else {"strain_visco": np.zeros(6 * n), "strain": np.zeros(6 * n)}
        st = ResidentState(law, n, stress0=s, history0=h)
        so, t = np.empty(6 * n), np.full(36 * n, np.nan)
        st.evaluate_into(0.0, 1.0, g, so, t)
        ref = t.copy()
        assert not np.isnan(ref).any()
        t[:] = -7.0                                   # sentinel: a second call must not touch the array
        st.evaluate_into(0.0, 1.0, g, so, t)
        assert np.all(t == -7.0)
        st.evaluate_into(0.0, 0.5, g, so, t)          # new del_t: SLS tangent changes, LE's does not
        if kind == "linear_elasticity":
            assert np.all(t == -7.0)
        else:
            assert not np.any(t == -7.0) and not np.array_equal(t, ref)
        t2 = np.full(36 * n, np.nan)                  # another array: written
        st.evaluate_into(0.0, 1.0, g, so, t2)
        assert np.array_equal(t2, ref)
        st2 = ResidentState(law, n, stress0=s, history0=h, reuse_constant_tangent=False)
        st2.evaluate_into(0.0, 1.0, g, so, t)
        t[:] = -7.0
        st2.evaluate_into(0.0, 1.0, g, so, t)
        assert np.array_equal(t, ref)


@pytest.mark.parametrize("numpy_grad", [False, True])
def test_vmm_placement_moves_the_state_and_keeps_results(numpy_grad):
    """placement="vmm" (the default for large device-assembler states): on the first evaluate the state moves
    both stress / history copies, the tangent (and the gradient staging buffer) into one interleaved VMM
    working set; every number equals the torch-allocated state's over iterations and commits, and views
    handed out stay valid after the state is gone."""
    import gc

    n = 1_000_003  # 288 MB of tangent: above ResidentState.AUTO_TUNE_MIN_BYTES
    rng = np.random.default_rng(3)
    law, s0, h0, grad = _sparse_case("VonMises3D", n, rng)
    a = ResidentState(law, n, stress0=s0, history0=h0, placement="torch")
    b = ResidentState(law, n, stress0=s0, history0=h0, placement="vmm")
    assert b.placement is None and b._vmm is None
    for inc in range(2):
        for it in range(2):
            g = grad(all_elastic=False, zoned=(it == 1))
            gb = g.cpu().numpy() if numpy_grad else g
            a.evaluate(0.0, 1.0, g)
            b.evaluate(0.0, 1.0, gb)
            assert torch.equal(a.stress, b.stress) and torch.equal(a.tangent, b.tangent), (inc, it)
            for k in h0:
                assert torch.equal(a.history[k], b.history[k]) and torch.equal(a.history_committed[k], b.history_committed[k])
        a.update()
        b.update()
    assert b.placement["mode"] == "vmm_interleaved" and ("grad" in b.placement["arrays"]) == numpy_grad
    assert a.placement is None
    assert b.tangent.data_ptr() % (2 << 20) == 0 and b.stress.data_ptr() % (2 << 20) == 0
    keep, ref = b.stress_committed, a.stress_committed.clone()
    del b
    gc.collect()
    torch.cuda.synchronize()
    assert torch.equal(keep, ref)  # the view keeps the working set's memory alive


@pytest.mark.parametrize("law_name", ["VonMises3D", "MisesPlasticityLinearHardening3D", "DruckerPragerHyperbolic3D"])
@pytest.mark.parametrize("n", [64 * 40 + 5, 64 * 64])
def test_packed_rows_inside_long_runs(n, law_name):
    """Every point has yielded once (hardly a +0.0 row: every tile's run is about 64 rows long) and only a scattered few yield now:
    the packed state moves the touched rows INSIDE the run (PackedRows::load_rows) while the trial run has the committed layout, the
    whole run otherwise -- plastic sets of 1-2 points per tile, a burst of 40 %, an iterate without a strain increment, commits in
    between.  Bit for bit the plain protocol, after every iterate."""
    rng = np.random.default_rng(n + 1)
    law, s0, h0, _ = _sparse_case(law_name, n, rng)
    key, w = ("eps_n", 6) if law_name == "VonMises3D" else ("history", 7)
    rows = h0[key].reshape(-1, w)[:, w - 6:]
    rows[:] = rng.normal(scale=1e-3, size=rows.shape)
    rows[rng.random(n) < 0.03] = 0.0   # a few virgin rows: points that can still GROW a run
    big = 5e-3 if law_name.startswith("Drucker") else 2e-2
    p = ResidentState(law, n, stress0=s0, history0=h0)
    f = ResidentState(law, n, stress0=s0, history0=h0, sparse_history=False, sparse_tangent=False, packed_history=False,
                      **({} if law_name == "VonMises3D" else {"split_history": False}))
    assert p._packed
    fractions = []
    for inc, plan in enumerate(([0.02, 0.03, 0.02], [0.02, 0.4, 0.03], [0.03, 0.0, 0.02], [0.01, 0.01, 0.01])):
        for it, frac in enumerate(plan):
            scale = np.where(rng.random(n) < frac, big, 1e-6)  # the chosen points yield, nearly all others stay elastic
            gn = rng.normal(size=(n, 9)) * scale[:, None]
            if law_name.startswith("Drucker"):
                gn[:, [0, 4, 8]] -= (0.95 * gn[:, [0, 4, 8]].sum(axis=1) / 3.0)[:, None]  # mostly isochoric
            g = torch.from_numpy(gn.reshape(-1)).cuda()
            p.evaluate(0.0, 1.0, g)
            f.evaluate(0.0, 1.0, g)
            fractions.append(int(f.check().n_plastic) / n)
            assert int(p.check().n_plastic) == int(f.check().n_plastic)
            assert torch.equal(p.stress, f.stress) and torch.equal(p.tangent, f.tangent), (inc, it)
            for k in h0:
                assert torch.equal(p.history[k], f.history[k]), (inc, it, k)
                assert torch.equal(p.history_committed[k], f.history_committed[k]), (inc, it, k)
        p.update(), f.update()
        for k in h0:
            assert torch.equal(p.history_committed[k], f.history_committed[k]), (inc, k)
    assert max(fractions) > 0.3 and min(fractions) < 0.1, fractions


@pytest.mark.parametrize("law_name", ["VonMises3D", "MisesPlasticityLinearHardening3D", "DruckerPrager3D", "DruckerPragerHyperbolic3D"])
@pytest.mark.parametrize("n", [64 * 30 + 11, 4000])
def test_packed_history_equals_the_plain_protocols(n, law_name):
    """Packed plastic-strain history (FCAMD_EVAL_PACKED_HISTORY, the default of ResidentState; VonMises3D's eps_n and the
    eps_p rows of the comfe-rs laws under the split layout): both copies of the array hold the rows of the ever-plastic
    points of every tile as one run, the commit stays a pointer swap.  Stress, tangent, the scalar history, the unpacked
    trial plastic strain and every committed state must equal the sparse protocol on the reference's layout and the
    full protocol bit for bit, over growing, shrinking and vanishing plastic sets, EVER masks from empty to full, device
    and host-assembler calls."""
    rng = np.random.default_rng(n)
    law, s0, h0, grad = _sparse_case(law_name, n, rng)
    p = ResidentState(law, n, stress0=s0, history0=h0)                               # the default: packed plastic-strain arrays, commit = pointer swap
    u = ResidentState(law, n, stress0=s0, history0=h0, packed_history=False)         # sparse protocol on the reference's layout
    f = ResidentState(law, n, stress0=s0, history0=h0, sparse_history=False, sparse_tangent=False)
    assert p._packed and not (u._packed or f._packed)
    key = "eps_n" if law_name == "VonMises3D" else "history"
    for st in (p, u):  # what went in comes out, before anything is evaluated
        assert torch.equal(st.history_committed[key], f.history_committed[key]) and torch.equal(st.history[key], f.history[key])
    shp, thp = np.empty(6 * n), np.empty(36 * n)
    n_plastic, ever_counts = [], []
    for inc in range(5):
        for it in range(3):
            g = grad(all_elastic=(inc == 2 and it == 1), zoned=(inc % 2 == 1))
            host_pass = (inc, it) == (3, 1)  # one host-assembler pass in between (fcamd_evaluate_resident with the flag)
            if host_pass:
                p.evaluate_into(0.0, 1.0, g.cpu().numpy(), shp, thp)
            else:
                p.evaluate(0.0, 1.0, g)
            u.evaluate(0.0, 1.0, g)
            f.evaluate(0.0, 1.0, g)
            n_plastic.append(int(f.check().n_plastic))
            assert int(u.check().n_plastic) == n_plastic[-1]
            if not host_pass:  # (the synchronous host pass reports through its return value)
                assert int(p.check().n_plastic) == n_plastic[-1]
            assert torch.equal(p.stress, f.stress) and torch.equal(u.stress, f.stress), (inc, it)
            if host_pass:
                assert np.array_equal(thp, f.tangent.cpu().numpy()) and np.array_equal(shp, f.stress.cpu().numpy())
            else:
                assert torch.equal(p.tangent, f.tangent), (inc, it)
            assert torch.equal(u.tangent, f.tangent), (inc, it)
            for k in h0:
                for st in (p, u):
                    assert torch.equal(st.history[k], f.history[k]), (inc, it, k, st._packed)
                    assert torch.equal(st.history_committed[k], f.history_committed[k]), (inc, it, k, st._packed)
        p.update(), u.update(), f.update()
        ever_counts.append(int(sum(bin(w & 0xFFFFFFFFFFFFFFFF).count("1") for w in p._ever[p._c].tolist())))
        for k in h0:
            assert torch.equal(p.history_committed[k], f.history_committed[k]), (inc, k)
            assert torch.equal(u.history_committed[k], f.history_committed[k]), (inc, k)
            assert torch.equal(p.history[k], f.history_committed[k]) or k != key  # nothing evaluated yet: trial view = committed
    assert max(n_plastic) > 0.1 * n
    # the EVER masks are exactly the rows that are not all +0.0, and they only grow
    rows = f.history_committed[key].view(n, -1)[:, -6:]
    assert ever_counts[-1] == int((rows.contiguous().view(torch.int64) != 0).any(dim=1).sum()) and ever_counts == sorted(ever_counts)
    # raw entry: the flag needs the sparse protocol's mask, both EVER-mask arrays and trial arrays of its own; 0.3's delta flag is gone
    mask = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
    g = grad(all_elastic=False, zoned=False)
    m = law._handle(0)
    hp = [f.history_committed[k].data_ptr() for k in h0]
    with pytest.raises(NotImplementedError, match="removed in 0.4"):
        m.evaluate_device_ex(0.0, 1.0, n, g.data_ptr(), f.stress_committed.data_ptr(), f.stress.data_ptr(), None, hp, hp, mask_ptr=mask.data_ptr(), flags=2)
    if law_name == "VonMises3D":
        ht = [f.history[k].data_ptr() for k in h0]
        with pytest.raises(ValueError, match="packed_mask_prev and packed_mask"):
            m.evaluate_device_ex(0.0, 1.0, n, g.data_ptr(), f.stress_committed.data_ptr(), f.stress.data_ptr(), None, hp, ht,
                                 mask_ptr=mask.data_ptr(), flags=8)
        with pytest.raises(ValueError, match="of their own"):
            m.evaluate_device_ex(0.0, 1.0, n, g.data_ptr(), f.stress_committed.data_ptr(), f.stress.data_ptr(), None, hp, hp,
                                 mask_ptr=mask.data_ptr(), flags=8, packed_mask_ptrs=(mask.data_ptr(), mask.data_ptr()))
    else:
        # 7-double rows (no split layout): the scalar sits inside the row, there is no array that only accumulates -- refused
        st = ResidentState(law, n, stress0=s0, history0=h0, split_history=False)
        assert not st._packed
        with pytest.raises(NotImplementedError, match="PACKED_HISTORY"):
            m.evaluate_device_ex(0.0, 1.0, n, g.data_ptr(), st.stress_committed.data_ptr(), st.stress.data_ptr(), None,
                                 [st.history_committed["history"].data_ptr()], [st.history["history"].data_ptr()],
                                 mask_ptr=mask.data_ptr(), flags=8, packed_mask_ptrs=(mask.data_ptr(), mask.clone().data_ptr()))


def test_packed_history_views_do_not_change_meaning():
    """a tensor taken from ``history`` must mean the same thing for the whole run.  Under the packed layout the
    plastic-strain array of ``history`` / ``history_committed`` is a COPY in the reference's layout from the first call on
    (nothing switches mid-run, ``generation`` stays put); the scalar history is the live tensor."""
    n = 64 * 20 + 5
    rng = np.random.default_rng(3)
    law, s0, h0, grad = _sparse_case("VonMises3D", n, rng)
    st = ResidentState(law, n, stress0=s0, history0=h0, placement="torch")
    f = ResidentState(law, n, stress0=s0, history0=h0, sparse_history=False, sparse_tangent=False, placement="torch")
    gen0 = st.generation
    held_eps, held_alpha = st.history["eps_n"], st.history["alpha"]
    before = held_eps.clone()
    for inc in range(3):
        for it in range(4):
            g = grad(all_elastic=False, zoned=False)
            st.evaluate(0.0, 1.0, g), f.evaluate(0.0, 1.0, g)
        assert torch.equal(held_eps, before)                      # a copy: never written behind the holder's back
        assert torch.equal(st.history["eps_n"], f.history["eps_n"])  # the view, asked again, is current
        st.update(), f.update()
    assert st.generation == gen0
    assert held_alpha.data_ptr() in (st._hist[0]["alpha"].data_ptr(), st._hist[1]["alpha"].data_ptr())  # a live tensor of the state


@pytest.mark.parametrize("law_name", ["MisesPlasticityLinearHardening3D", "DruckerPrager3D", "DruckerPragerHyperbolic3D"])
def test_split_history_in_place_and_out_of_place_without_mask(law_name):
    """FCAMD_EVAL_SPLIT_HISTORY through the raw device entry (fcamd_evaluate_device_ex), without the sparse protocol: in
    place (history == history_prev: the reference's contract on the split arrays) and out of place (every row copied)
    against the law's plain evaluate on the reference's 7-double rows.  Also: laws without such rows refuse the flag."""
    from fenics_constitutive_amd.device import join_history_rows, split_history_rows

    n = 64 * 25 + 13
    rng = np.random.default_rng(5)
    law, s0, h0, grad = _sparse_case(law_name, n, rng)
    g = grad(all_elastic=False, zoned=False)
    f = dict(dtype=torch.float64, device="cuda")
    s_ref, t_ref = torch.from_numpy(s0).cuda(), torch.empty(36 * n, **f)
    h_ref = {"history": torch.from_numpy(h0["history"]).cuda()}
    law.evaluate(0.0, 1.0, g, s_ref, t_ref, h_ref)
    torch.cuda.synchronize()
    assert law.device_stats().n_plastic > 0
    # in place on the split arrays
    s1, t1 = torch.from_numpy(s0).cuda(), torch.empty(36 * n, **f)
    h1 = split_history_rows(torch.from_numpy(h0["history"]).cuda())
    law.evaluate_from(0.0, 1.0, g, s1, s1, t1, h1, h1, split_history=True)
    torch.cuda.synchronize()
    assert torch.equal(s1, s_ref) and torch.equal(t1, t_ref) and torch.equal(join_history_rows(h1), h_ref["history"])
    # out of place, no mask: every row of the trial arrays is written
    sp, hp = torch.from_numpy(s0).cuda(), split_history_rows(torch.from_numpy(h0["history"]).cuda())
    s2, t2 = torch.full((6 * n,), float("nan"), **f), torch.empty(36 * n, **f)
    h2 = {k: torch.full_like(v, float("nan")) for k, v in hp.items()}
    law.evaluate_from(0.0, 1.0, g, sp, s2, t2, hp, h2, split_history=True)
    torch.cuda.synchronize()
    assert torch.equal(s2, s_ref) and torch.equal(t2, t_ref) and torch.equal(join_history_rows(h2), h_ref["history"])
    vm = fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0})
    hv = {"scalar": torch.zeros(n, **f), "rows": torch.zeros(6 * n, **f)}
    with pytest.raises(NotImplementedError, match="SPLIT_HISTORY"):
        vm.evaluate_from(0.0, 1.0, g, s1, s2, t2, hv, hv, split_history=True)


@pytest.mark.parametrize("kind,split", [("von_mises_3d", True), ("comfe_mises_plasticity", True), ("comfe_mises_plasticity", False)])
def test_row_masked_access_leaves_identical_bits(kind, split):
    """Row-masked history access (tiles with at most ``masked_max`` touched rows move only the chunks of those rows) against
    the dense tile access (``masked_max`` = 0) and the always-masked one (64): the same BITS in every array -- in place (the
    reference contract) and under the sparse protocol of a resident state, with plastic sets that shrink and grow."""
    from fenics_constitutive_amd import _capi
    from test_gpu_parity import make_law, random_case

    n = 64 * 40 + 13
    p, g0, s0, h0 = random_case(kind, n, seed=21)
    law = make_law(kind, p)
    ctx = law._handle(_capi.default_device()).ctx
    results = {}
    try:
        for fill in (0, 20, 64):
            ctx.set_option("masked_max", fill)
            out = []
            # in place, device tensors: two calls, the second with a smaller plastic set
            s, t = torch.from_numpy(s0).cuda(), torch.full((36 * n,), float("nan"), dtype=torch.float64, device="cuda")
            h = {k: torch.from_numpy(v).cuda() for k, v in h0.items()}
            for scale in (1.0, 0.3):
                law.evaluate(0.0, 1.0, torch.from_numpy(g0 * scale).cuda(), s, t, h)
                law.device_stats()
                out += [s.clone(), t.clone()] + [v.clone() for v in h.values()]
            # sparse protocol: resident state, plastic set shrinks, grows, commit, shrinks
            rs = ResidentState(law, n, stress0=s0, history0=h0, split_history=split, placement="torch")
            for k, scale in enumerate((1.0, 0.2, 1.5, 0.4)):
                rs.evaluate(0.0, 1.0, g0 * scale)
                rs.check()
                out += [rs.stress.clone(), rs.tangent.clone()] + [v.clone() for v in rs.history.values()]
                if k == 2:
                    rs.update()
                    out += [v.clone() for v in rs.history_committed.values()]
            results[fill] = out
    finally:
        ctx.set_option("masked_max", -1)
    for fill in (20, 64):
        assert len(results[fill]) == len(results[0])
        for a, b in zip(results[fill], results[0]):
            assert torch.equal(a.view(torch.int64), b.view(torch.int64)), fill


def test_synthetic_twin_reproduces_the_request_stream_bookkeeping():
    """Context option "twin_masks" (csrc/fcamd_kernels.hip: evaluate_twin_kernel): the measurement device behind the bench row's
    mem_floor_ms.  The twin of a packed sparse-protocol VonMises3D launch reads every tile's plastic ballot from a recording and must
    leave the protocol words exactly as the real launch does: the same sparse-history mask, the same EVER masks of the trial copy."""
    import fenics_constitutive_amd as fc
    from fenics_constitutive_amd import _capi
    from fenics_constitutive_amd.resident import ResidentState

    n = 64 * 40_000  # above the batch kernel's size: the launch is the law's own kernel (the twin exists for that one)
    gen = torch.Generator(device="cuda").manual_seed(5)
    f = dict(dtype=torch.float64, device="cuda")

    def grad():
        g = torch.randn(9 * n, generator=gen, **f)
        g.view(n, 9).mul_(torch.pow(10.0, torch.rand(n, generator=gen, **f) * 2.0 - 4.0)[:, None])
        return g

    law = fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0})
    st = ResidentState(law, n, history0={"eps_n": torch.zeros(6 * n, **f), "alpha": torch.rand(n, generator=gen, **f) * 0.02}, placement="torch")
    st.evaluate(0.0, 1.0, grad())
    st.update()
    ga, gb = grad(), grad()
    st.evaluate(0.0, 1.0, ga)
    st.evaluate(0.0, 1.0, gb)
    torch.cuda.synchronize()
    mask_b, ever_b = st._mask.clone(), st._ever[1 - st._c].clone()
    st.evaluate(0.0, 1.0, ga)
    torch.cuda.synchronize()
    assert not torch.equal(st._mask, mask_b)
    ctx = law._handle(_capi.default_device()).ctx
    ctx.set_option("twin_masks", mask_b.data_ptr())
    try:
        st.evaluate(0.0, 1.0, gb)  # the twin of iterate B after iterate A
        torch.cuda.synchronize()
    finally:
        ctx.set_option("twin_masks", 0)
    assert torch.equal(st._mask, mask_b) and torch.equal(st._ever[1 - st._c], ever_b)
    st.evaluate(0.0, 1.0, gb)  # real launches again (what the twin wrote into the trial arrays is overwritten where the protocol touches)
    torch.cuda.synchronize()
    assert torch.equal(st._mask, mask_b)
