"""Host (ndarray) data paths: with every array of a call inside page-locked, GPU-mapped ranges
(``fcamd_register_host_buffer``) ``fcamd_evaluate_host`` launches the kernel directly on the caller's
NumPy arrays (zero copy); pageable arrays are page-locked for the duration of the call (large calls) or moved
through the context's page-locked scratch (small calls, arrays that cannot be locked, arrays off the 16-byte
grid) -- never handed to the HIP runtime's pageable-copy path.  All paths must be bit-identical (same kernel,
same inputs) and therefore within the parity tolerances of the oracle."""

import numpy as np
import pytest
from golden_util import rel_err

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import fenics_constitutive_amd as fc  # noqa: E402
from fenics_constitutive_amd import _capi  # noqa: E402
from fenics_constitutive_amd.resident import ResidentState  # noqa: E402
from oracle import c_oracle as CO  # noqa: E402
from test_gpu_parity import CLASS, KINDS, STRICT, TOL, compare, make_law, oracle_run, random_case  # noqa: E402

ZC = _capi.HOST_ZERO_COPY_IN | _capi.HOST_ZERO_COPY_OUT
BOUNCE, TEMP = _capi.HOST_BOUNCE, _capi.HOST_TEMP_LOCK
BOUNCE_MAX = 256 << 10  # default of the "bounce_max" option


def data_path(ctx):
    """FCAMD_HOST_* flags of the context's last host call without FCAMD_HOST_TANGENT_CPU (16): who wrote the tangent ROWS -- the kernel
    over the link, or host threads from 8 doubles per plastic point, for calls of 65536 points and more (tests/test_gpu_host_tangent.py)
    -- does not change how the other arrays travel, which is what these tests pin down"""
    return ctx.last_host_mode() & ~16


def pageable_mode(nbytes):
    """data path of a call that moves ``nbytes`` of pageable caller memory"""
    return BOUNCE if nbytes <= BOUNCE_MAX else (ZC | TEMP)


class Pinned:
    """Registers arrays with the law's context for the duration of a ``with`` block."""

    def __init__(self, law, arrays):
        self.ctx, self.arrays = law._handle(_capi.default_device()).ctx, [a for a in arrays if a is not None and a.size]

    def __enter__(self):
        for a in self.arrays:
            self.ctx.register_host_buffer(a)
        return self.ctx

    def __exit__(self, *exc):
        for a in self.arrays:
            self.ctx.unregister_host_buffer(a)


def own(a):
    """Copy of ``a`` in a buffer of its own pages, 16-byte aligned."""
    import mmap

    if a is None:
        return None
    m = mmap.mmap(-1, max(a.nbytes, 8))
    out = np.frombuffer(m, dtype=np.float64, count=a.size)
    out[:] = a
    return out


@pytest.mark.parametrize("n", [1, 63, 65, 1000, 70_001])
@pytest.mark.parametrize("kind", KINDS)
def test_zero_copy_equals_staged_and_oracle(kind, n):
    p, g, s, h = random_case(kind, n, seed=100 + n)
    ref = oracle_run(kind, p, 1.3, g, s, h, mod=CO)
    law = make_law(kind, p)
    # staged
    s1, t1 = s.copy(), np.full(36 * n, np.nan)
    h1 = None if h is None else {k: v.copy() for k, v in h.items()}
    law.evaluate(0.0, 1.3, g, s1, t1, h1)
    ctx = law._handle(_capi.default_device()).ctx
    moved = g.nbytes + s1.nbytes + t1.nbytes + (0 if h1 is None else sum(v.nbytes for v in h1.values()))
    assert data_path(ctx) == (pageable_mode(moved) if n else 0)
    # zero copy: every array in its own page-locked mapping
    g2, s2, t2 = own(g), own(s), own(np.full(36 * n, np.nan))
    h2 = None if h is None else {k: own(v) for k, v in h.items()}
    with Pinned(law, [g2, s2, t2] + ([] if h2 is None else list(h2.values()))):
        law.evaluate(0.0, 1.3, g2, s2, t2, h2)
        assert data_path(ctx) == ZC
        assert law.last_stats is not None
    assert np.array_equal(s1, s2) and np.array_equal(t1, t2), f"{kind} n={n}: zero copy differs from the pageable path"
    if h is not None:
        for k in h:
            assert np.array_equal(h1[k], h2[k]), k
    compare((s2, t2, h2), ref, TOL[CLASS[kind]], f"{kind} n={n} zero copy")
    compare((s2, t2, h2), ref, STRICT[CLASS[kind]], f"strict {kind} n={n} zero copy")
    assert np.array_equal(g2, g), "the gradient is read-only"


def test_zero_copy_without_tangent_and_n_zero():
    law = make_law("comfe_mises_plasticity", {"mu": 80769.0, "kappa": 175000.0, "y_0": 1200.0, "h": 200.0})
    n = 5000
    p, g, s, h = random_case("comfe_mises_plasticity", n, seed=4)
    ref = oracle_run("comfe_mises_plasticity", p, 1.0, g, s, h, mod=CO)
    g2, s2, hh = own(g), own(s), own(h["history"])
    with Pinned(law, [g2, s2, hh]) as ctx:
        law.evaluate(0.0, 1.0, g2, s2, None, {"history": hh})  # Rust binding: tangent=None skips it
        assert data_path(ctx) == ZC
        law.evaluate(0.0, 1.0, g2[:0], s2[:0], None, {"history": hh[:0]})  # n = 0: nothing to launch
    assert rel_err(s2, ref[0]) <= 1e-11 and rel_err(hh, ref[2]["history"]) <= 1e-11


def test_sub_ranges_of_one_registration_take_the_zero_copy_path():
    """dolfinx hands over views; any sub-range of a registered range qualifies (also several arrays
    carved from one registered slab)."""
    n = 3000
    p, g, s, h = random_case("von_mises_3d", n, seed=8)
    ref = oracle_run("von_mises_3d", p, 1.0, g, s, h, mod=CO)
    law = make_law("von_mises_3d", p)
    slab = own(np.zeros(2 + 9 * n + 6 * n + 36 * n + 6 * n + n + 64))
    o = 2  # 16-byte aligned offset into the slab
    views = []
    for k in (9 * n, 6 * n, 36 * n, 6 * n, n):
        views.append(slab[o : o + k])
        o += k + (k % 2)
    gv, sv, tv, ev, av = views
    gv[:], sv[:], ev[:], av[:] = g, s, h["eps_n"], h["alpha"]
    with Pinned(law, [slab]) as ctx:
        law.evaluate(0.0, 1.0, gv, sv, tv, {"eps_n": ev, "alpha": av})
        assert data_path(ctx) == ZC
    compare((sv, tv, {"eps_n": ev, "alpha": av}), ref, STRICT["pl"], "slab views")


def test_fallbacks_to_the_scratch_path():
    n = 2000
    p, g, s, h = random_case("spring_maxwell", n, seed=12)
    ref = oracle_run("spring_maxwell", p, 0.5, g, s, h, mod=CO)
    law = make_law("spring_maxwell", p)
    ctx = law._handle(_capi.default_device()).ctx

    def fresh():
        return own(g), own(s), own(np.full(36 * n, np.nan)), {k: own(v) for k, v in h.items()}

    # (1) one array of the call is not registered: it is page-locked for the call
    g2, s2, t2, h2 = fresh()
    with Pinned(law, [g2, s2, t2, h2["strain"]]):
        law.evaluate(0.0, 0.5, g2, s2, t2, h2)
        assert data_path(ctx) == (ZC | TEMP)
    compare((s2, t2, h2), ref, STRICT["sls"], "partly registered")
    # (2) registered but 8 bytes off the 16-byte grid: the kernel's vector accesses need alignment -- DMA through the
    # device chunk buffers instead
    slab = own(np.zeros(6 * n + 1))
    s3 = slab[1:]
    s3[:] = s
    g2, _, t2, h2 = fresh()
    with Pinned(law, [g2, slab, t2] + list(h2.values())):
        law.evaluate(0.0, 0.5, g2, s3, t2, h2)
        assert data_path(ctx) == 0
    compare((s3, t2, h2), ref, STRICT["sls"], "misaligned")
    # (3) a view that reaches beyond its registered range is never handed to the kernel or to the DMA engines
    # (a partly page-locked range): the CPU moves it through the scratch
    both = own(np.zeros(12 * n))
    g2, _, t2, h2 = fresh()
    s4 = both[: 6 * n]
    s4[:] = s
    half = both[: 3 * n]
    with Pinned(law, [g2, half, t2] + list(h2.values())):
        law.evaluate(0.0, 0.5, g2, s4, t2, h2)
        assert data_path(ctx) == BOUNCE
    compare((s4, t2, h2), ref, STRICT["sls"], "beyond the registered range")
    # ... also when the call is too large for one pass through the scratch (several chunks)
    ctx.set_option("bounce_max", 64 * 1024)
    try:
        g2, _, t2, h2 = fresh()
        s4[:] = s
        with Pinned(law, [g2, half, t2] + list(h2.values())):
            law.evaluate(0.0, 0.5, g2, s4, t2, h2)
            assert data_path(ctx) == BOUNCE
        compare((s4, t2, h2), ref, STRICT["sls"], "beyond the registered range, chunked scratch")
        # pageable arrays above the threshold: page-locked for the call, kernel directly on them
        g2, s2, t2, h2 = fresh()
        law.evaluate(0.0, 0.5, g2, s2, t2, h2)
        assert data_path(ctx) == (ZC | TEMP)
        compare((s2, t2, h2), ref, STRICT["sls"], "temporarily locked")
        # ... or, with zero copy switched off, DMA between them and the device chunk buffers
        ctx.set_option("zero_copy", 0)
        for chunk in (0, 256):  # one chunk; eight chunks over the four slots
            ctx.set_option("host_chunk", chunk)
            g2, s2, t2, h2 = fresh()
            law.evaluate(0.0, 0.5, g2, s2, t2, h2)
            assert data_path(ctx) == TEMP
            compare((s2, t2, h2), ref, STRICT["sls"], f"temporarily locked, chunked DMA ({chunk})")
    finally:
        ctx.set_option("bounce_max", BOUNCE_MAX), ctx.set_option("zero_copy", 1), ctx.set_option("host_chunk", 0)
    # (4) after unregistering everything the same arrays are pageable again
    g2, s2, t2, h2 = fresh()
    with Pinned(law, [g2, s2, t2] + list(h2.values())):
        pass
    law.evaluate(0.0, 0.5, g2, s2, t2, h2)
    assert data_path(ctx) == (ZC | TEMP)
    compare((s2, t2, h2), ref, STRICT["sls"], "unregistered again")


def test_scratch_path_in_chunks():
    """Arrays that cannot be page-locked (here: a tangent array that lies only half inside a registered range) and
    are too large for one pass through the scratch: several chunks, each copied in, evaluated, copied out."""
    n = 200_003  # linear elasticity moves 408 B/pt: 64 MiB chunks hold 164 480 points
    p, g, s, h = random_case("linear_elasticity", n, seed=3)
    ref = oracle_run("linear_elasticity", p, 1.0, g, s, h, mod=CO)
    law = make_law("linear_elasticity", p)
    t = own(np.full(36 * n, np.nan))
    s1 = s.copy()
    with Pinned(law, [t[: 18 * n]]) as ctx:
        ctx.set_option("host_tangent_threads", 0)  # the kernel writes the tangent: the array has to be reachable from the GPU
        try:
            law.evaluate(0.0, 1.0, g, s1, t, None)
            assert data_path(ctx) == BOUNCE
        finally:
            ctx.set_option("host_tangent_threads", -1)
        compare((s1, t, None), ref, STRICT["le"], "chunked scratch")
        # with the tangent rows written by host threads (the default at this size) the array is never handed to the GPU: nothing
        # to lock, the other arrays are page-locked for the call and the kernel works on them in place
        s3, t[:] = s.copy(), np.nan
        law.evaluate(0.0, 1.0, g, s3, t, None)
        assert ctx.last_host_mode() == (ZC | TEMP | 16)
        assert np.array_equal(s1, s3)
    compare((s3, t, None), ref, STRICT["le"], "host-written tangent in a half-registered array")
    # the same call on pageable arrays: page-locked for the call
    s2, t2 = s.copy(), np.full(36 * n, np.nan)
    law.evaluate(0.0, 1.0, g, s2, t2, None)
    assert data_path(ctx) == (ZC | TEMP)
    assert np.array_equal(s1, s2) and np.array_equal(t, t2)


@pytest.mark.parametrize("cname", ["UNIAXIAL_STRESS", "PLANE_STRAIN"])
@pytest.mark.parametrize("n", [10_007, 150_001])
def test_zero_copy_low_dimensional_constraints(cname, n):
    from oracle import numpy_oracle as O
    from wrappers_util import CPARAMS

    c = fc.StressStrainConstraint[cname]
    rng = np.random.default_rng(5)
    gdim, sd = O.DIMS[cname]
    g = rng.normal(scale=1e-3, size=gdim * gdim * n)
    s0 = rng.normal(size=sd * n)
    h0 = {"strain_visco": rng.normal(scale=1e-4, size=sd * n), "strain": rng.normal(scale=1e-3, size=sd * n)}
    s_ref, t_ref, h_ref = s0.copy(), np.zeros(sd * sd * n), {k: v.copy() for k, v in h0.items()}
    O.MODELS_C["kelvin"](CPARAMS["kelvin"], cname, 0.0, 0.7, g, s_ref, t_ref, h_ref)
    law = fc.SpringKelvinModel(CPARAMS["kelvin"], c)
    g2, s2, t2 = own(g), own(s0), own(np.full(sd * sd * n, np.nan))
    h2 = {k: own(v) for k, v in h0.items()}
    with Pinned(law, [g2, s2, t2] + list(h2.values())) as ctx:
        law.evaluate(0.0, 0.7, g2, s2, t2, h2)
        assert data_path(ctx) == ZC
    assert rel_err(s2, s_ref) <= 1e-14 and rel_err(t2, t_ref) <= 1e-14
    for k in h2:
        assert rel_err(h2[k], h_ref[k]) <= 1e-14
    # the same call on pageable arrays (scratch or page-locked for the call, by size): bit-identical
    s3, t3, h3 = s0.copy(), np.full(sd * sd * n, np.nan), {k: v.copy() for k, v in h0.items()}
    law.evaluate(0.0, 0.7, g, s3, t3, h3)
    moved = g.nbytes + s3.nbytes + t3.nbytes + sum(v.nbytes for v in h3.values())
    assert data_path(ctx) == pageable_mode(moved)
    assert np.array_equal(s3, s2) and np.array_equal(t3, t2) and all(np.array_equal(h3[k], h2[k]) for k in h2)


def test_nonconvergence_is_reported_on_the_zero_copy_path():
    from test_oracle_c import NONCONVERGING, nonconverging_inputs

    law = fc.VonMises3D(NONCONVERGING)
    g, s, t, h = nonconverging_inputs(70)
    g, s, t, h = own(g), own(s), own(t), {k: own(v) for k, v in h.items()}
    with Pinned(law, [g, s, t] + list(h.values())) as ctx:
        with pytest.raises(RuntimeError, match="did not converge"):
            law.evaluate(0, 1.0, g, s, t, h)
        assert data_path(ctx) == ZC


@pytest.mark.parametrize("kind", ["von_mises_3d", "linear_elasticity"])
def test_resident_evaluate_into_zero_copy(kind):
    """ResidentState.evaluate_into with page-locked gradient / tangent arrays: the kernel reads the
    gradient from and writes the tangent to the caller's arrays; same numbers as with pageable ones."""
    n = 200_003
    p, g, s, h = random_case(kind, n, seed=41)
    law = make_law(kind, p)
    out = {}
    for mode in ("staged", "zero_copy"):
        st = ResidentState(law, n, stress0=s, history0=h)
        gg, so, to = own(g), own(np.zeros(6 * n)), own(np.full(36 * n, np.nan))
        ctx = law._handle(_capi.default_device()).ctx
        if mode == "zero_copy":
            with Pinned(law, [gg, so, to]):
                st.evaluate_into(0.0, 1.0, gg, so, to)
                assert data_path(ctx) == ZC
        else:  # pageable, 200 003 points: page-locked for the call
            st.evaluate_into(0.0, 1.0, gg, so, to)
            assert data_path(ctx) == (ZC | TEMP)
        hist = None if st.history is None else {k: v.cpu().numpy() for k, v in st.history.items()}
        out[mode] = (so.copy(), to.copy(), hist, st.stress.cpu().numpy())
    a, b = out["staged"], out["zero_copy"]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3])
    assert np.array_equal(b[0], b[3]), "host stress == device trial stress"
    if a[2] is not None:
        for k in a[2]:
            assert np.array_equal(a[2][k], b[2][k])
    ref = oracle_run(kind, p, 1.0, g, s, h, mod=CO)
    compare((b[0], b[1], b[2]), ref, TOL[CLASS[kind]], f"resident zero copy {kind}")


@pytest.mark.parametrize("law_name", ["von_mises_3d", "comfe_mises_plasticity"])
def test_sparse_tangent_into_page_locked_host_array(law_name):
    """evaluate_into with a page-locked tangent array: from the second call on only the rows of plastic /
    formerly plastic points cross PCIe (FCAMD_EVAL_SPARSE_TANGENT).  The array must equal, at every
    Newton iteration, what a state that downloads every row produces -- with plastic sets that grow,
    shrink and vanish, across commits, and when device-side evaluates come in between."""
    n = 50_000 + 37
    p, g0, s, h = random_case(law_name, n, seed=5)
    law = make_law(law_name, p)
    rng = np.random.default_rng(9)
    sp = ResidentState(law, n, stress0=s, history0=h)                          # sparse tangent (default)
    fu = ResidentState(law, n, stress0=s, history0=h, sparse_tangent=False)    # every row, every call
    assert sp._sparse_tangent and not fu._sparse_tangent
    gg, s_sp, t_sp = own(g0), own(np.zeros(6 * n)), own(np.full(36 * n, np.nan))
    s_fu, t_fu = np.zeros(6 * n), np.full(36 * n, np.nan)
    ctx = law._handle(_capi.default_device()).ctx
    scales = [1.0, 1.4, 0.02, 0.7, 0.0, 1.2, 1.2, 0.3]   # 0.02 / 0.0: (nearly) everything elastic again
    with Pinned(law, [gg, s_sp, t_sp]):
        for k, sc in enumerate(scales):
            gg[:] = g0 * sc * (1.0 + 0.2 * rng.standard_normal(1)[0])
            if k == 5:  # a device-side evaluate in between: the host array misses one update ...
                sp.evaluate(0.0, 1.0, gg)
                fu.evaluate(0.0, 1.0, gg)
                assert torch.equal(sp.tangent, fu.tangent)
            sp.evaluate_into(0.0, 1.0, gg, s_sp, t_sp)   # ... so this call has to write every row again
            assert data_path(ctx) == ZC
            fu.evaluate_into(0.0, 1.0, gg, s_fu, t_fu)
            assert np.array_equal(t_sp, t_fu), f"call {k}: sparse tangent differs"
            assert np.array_equal(s_sp, s_fu)
            if k in (2, 6):
                sp.update()
                fu.update()
    assert 0 < law.last_stats.n_plastic < n


@pytest.mark.parametrize("kind", ["von_mises_3d", "spring_maxwell"])
def test_resident_scratch_path_in_chunks(kind, request):
    """fcamd_evaluate_resident with host arrays that cannot be page-locked (the tangent array lies half inside a
    registered range) and do not fit one pass through the scratch: several chunks, the sparse-history mask words and the
    state rows of every chunk at the right offsets -- same numbers as the one-launch pass on pageable arrays."""
    n = 200_003
    p, g, s, h = random_case(kind, n, seed=17)
    law = make_law(kind, p)
    a = ResidentState(law, n, stress0=s, history0=h)
    b = ResidentState(law, n, stress0=s, history0=h)
    ctx = law._handle(_capi.default_device()).ctx
    sa, ta = np.zeros(6 * n), own(np.full(36 * n, np.nan))
    sb, tb = np.zeros(6 * n), np.full(36 * n, np.nan)
    # (the kernel writes the tangent here -- "host_tangent_threads" = 0: with the rows written by host threads a tangent array that
    # cannot be page-locked is no obstacle at all, tests/test_gpu_host_tangent.py)
    ctx.set_option("host_tangent_threads", 0)
    request.addfinalizer(lambda: ctx.set_option("host_tangent_threads", -1))
    with Pinned(law, [ta[: 18 * n]]):
        for it, scale in enumerate((1.0, 0.3, 1.5)):
            gi = g * scale
            a.evaluate_into(0.0, 1.0, gi, sa, ta)
            if kind == "von_mises_3d" or it == 0:  # SLS: the constant tangent is written once, later calls do not pass the array
                assert data_path(ctx) == BOUNCE
            b.evaluate_into(0.0, 1.0, gi, sb, tb)
            assert data_path(ctx) & TEMP
            assert np.array_equal(sa, sb) and np.array_equal(ta, tb), it
            assert torch.equal(a.stress, b.stress)
            for k in (h or {}):
                assert torch.equal(a.history[k], b.history[k]), (it, k)
            if it == 1:
                a.update(), b.update()


@pytest.mark.parametrize("kind", ["von_mises_3d", "linear_elasticity"])
def test_resident_chunked_dma_pipeline(kind):
    """Option "zero_copy" = 0: fcamd_evaluate_resident page-locks the pageable host arrays and moves them chunk by chunk
    by DMA through the four slots (here 13 chunks of 16 384 points) -- same numbers as the one-launch pass."""
    n = 200_003
    p, g, s, h = random_case(kind, n, seed=29)
    law = make_law(kind, p)
    a = ResidentState(law, n, stress0=s, history0=h)
    b = ResidentState(law, n, stress0=s, history0=h)
    ctx = law._handle(_capi.default_device()).ctx
    sa, ta, sb, tb = np.zeros(6 * n), np.full(36 * n, np.nan), np.zeros(6 * n), np.full(36 * n, np.nan)
    for it, scale in enumerate((1.0, 0.4)):
        gi = g * scale
        ctx.set_option("zero_copy", 0), ctx.set_option("host_chunk", 16384)
        try:
            a.evaluate_into(0.0, 1.0, gi, sa, ta)
            assert data_path(ctx) == TEMP
        finally:
            ctx.set_option("zero_copy", 1), ctx.set_option("host_chunk", 0)
        b.evaluate_into(0.0, 1.0, gi, sb, tb)
        # (linear elasticity: the constant tangent is written by the first call only, later ones do not pass the array)
        assert data_path(ctx) == ((ZC | TEMP) if kind == "von_mises_3d" or it == 0 else (_capi.HOST_ZERO_COPY_IN | TEMP))
        assert np.array_equal(sa, sb), it
        if kind == "von_mises_3d" or it == 0:
            assert np.array_equal(ta, tb), it
        assert torch.equal(a.stress, b.stress)
        a.update(), b.update()


def test_registering_an_array_while_another_thread_evaluates_on_it():
    """Two threads, a context each.  One evaluates in place on pageable arrays again and again (every call page-locks them
    for its duration); the other registers the very same tangent array with ITS context in the middle of that.  The
    registration is either refused (a call holds a call-scoped lock on the range right now) or lands between two calls, in
    which case the evaluating thread's later calls use the registered range as it is -- never a second page lock on a locked
    range, whose release would pull the pages from under the other holder (this runtime: abort in hipHostUnregister)."""
    import threading
    import time

    import fenics_constitutive_amd as fc
    from fenics_constitutive_amd import _capi

    n = 1_500_000
    rng = np.random.default_rng(12)
    g = rng.standard_normal(9 * n) * 1e-3
    law = fc.LinearElasticityModel({"E": 42.0, "nu": 0.3}, fc.StressStrainConstraint.FULL)
    s_ref, t_ref = np.zeros(6 * n), np.zeros(36 * n)
    law.evaluate(0.0, 1.0, g, s_ref, t_ref, None)
    t = np.full(36 * n, np.nan)
    stop, errors, calls = threading.Event(), [], [0]

    def work():
        try:
            while not stop.is_set():
                s = np.zeros(6 * n)
                law.evaluate(0.0, 1.0, g, s, t, None)
                assert np.array_equal(s, s_ref)
                calls[0] += 1
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = threading.Thread(target=work)
    th.start()
    ctx = _capi.get_context(_capi.default_device())
    refused, registered = 0, False
    t_end = time.time() + 20.0
    while calls[0] < 2 and time.time() < t_end:
        time.sleep(0.001)
    while not registered and time.time() < t_end:
        try:
            ctx.register_host_buffer(t)
            registered = True
        except (ValueError, RuntimeError) as e:
            assert "in progress" in str(e), e
            refused += 1
    before = calls[0]
    while calls[0] < before + 3 and time.time() < t_end and not errors:
        time.sleep(0.005)
    stop.set()
    th.join()
    assert not errors, errors
    assert registered and calls[0] >= before + 3
    assert np.array_equal(t, t_ref)
    ctx.unregister_host_buffer(t)
    law.evaluate(0.0, 1.0, g, np.zeros(6 * n), t, None)  # pageable again
    assert np.array_equal(t, t_ref)
    print(f"registration refused {refused} times while a call was in progress, {calls[0]} calls")
