"""The tangent of the host (ndarray) entries rebuilt on the CPU (context option "host_tangent_threads",
csrc/fcamd_hosttangent.cpp): the reference builds the tangent on the host too -- np.tile(D.flatten(), n)
(linear_elasticity_model.py:45, spring_maxwell_model.py:84-86, spring_kelvin_model.py:85-86) and
ka xioi + B xpp + C N (x) N (mises_plasticity_isotropic_hardening.py:170-175) -- so the kernel sends no tangent (constant
laws) or the 8 doubles per point its own tangent writer starts from (VonMises3D, comfe-rs Mises).

The bar: the caller's arrays hold BIT FOR BIT what the kernel's own tangent stores leave there ("host_tangent_threads" = 0),
on the reference's golden vectors, on ragged and large random inputs, for the host and the resident entry.
"""

import numpy as np
import pytest
from golden_util import load_calls

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import fenics_constitutive_amd as fc  # noqa: E402
from fenics_constitutive_amd import _capi  # noqa: E402
from test_gpu_parity import CLASS, GOLDEN, STRICT, TOL, compare, make_law, random_case, run_host  # noqa: E402

HOST_TANGENT_CPU = 16  # include/fcamd.h: FCAMD_HOST_TANGENT_CPU
SC = fc.StressStrainConstraint


@pytest.fixture
def ctx():
    c = _capi.get_context(_capi.default_device())
    saved = {k: c.get_option(k) for k in ("host_tangent_min_points", "host_tangent_chunk", "bounce_max")}
    auto = c.get_option("host_tangent_threads")
    yield c
    for k, v in saved.items():
        c.set_option(k, v)
    c.set_option("host_tangent_threads", -1)
    assert c.get_option("host_tangent_threads") == auto


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


def both_ways(ctx, law, del_t, g, s0, h0, threads=3, tangent_offset=0):
    """the same in-place host call with the kernel's tangent stores and with the CPU's: (stress, tangent, history, mode) each"""
    out = []
    n = g.size // law.geometric_dim**2
    td = law.stress_strain_dim**2
    for th in (0, threads):
        ctx.set_option("host_tangent_threads", th)
        s = s0.copy()
        buf = np.full(td * n + 2, np.nan)
        t = buf[tangent_offset: tangent_offset + td * n]
        h = None if h0 is None else {k: v.copy() for k, v in h0.items()}
        law.evaluate(0.0, del_t, g, s, t, h)
        out.append((s, t.copy(), h, ctx.last_host_mode(), law.last_stats))
    return out


def assert_identical(kernel, cpu, what):
    s0, t0, h0, mode0, st0 = kernel
    s1, t1, h1, mode1, st1 = cpu
    assert not (mode0 & HOST_TANGENT_CPU), f"{what}: threads = 0 must keep the kernel's tangent stores"
    assert mode1 & HOST_TANGENT_CPU, f"{what}: the CPU path did not run (mode {mode1})"
    assert not np.isnan(t1).any(), f"{what}: unwritten tangent entries"
    assert np.array_equal(bits(t0), bits(t1)), f"{what}: tangent differs in {int(np.sum(bits(t0) != bits(t1)))} entries"
    assert np.array_equal(bits(s0), bits(s1)), f"{what}: stress"
    for k in (h0 or {}):
        assert np.array_equal(bits(h0[k]), bits(h1[k])), f"{what}: history[{k}]"
    assert st0.n_plastic == st1.n_plastic and st0.n_newton_iters == st1.n_newton_iters, what


@pytest.mark.parametrize("fname,kind,c", GOLDEN, ids=[f"{k}-{c.name}" for _, k, c in GOLDEN])
def test_golden_vectors_through_the_cpu_tangent(ctx, fname, kind, c):
    """the reference's own outputs, with every call (64 points and more) taking the new path"""
    ctx.set_option("host_tangent_min_points", 0)
    ctx.set_option("bounce_max", 0)  # small calls would go through the scratch otherwise
    ctx.set_option("host_tangent_threads", 2)
    law = make_law(kind, c.params)
    s, t, h = c.fresh()
    t[:] = np.nan
    got = run_host(law, c.del_t, c.grad.copy(), s, t, h)
    n = c.grad.size // 9
    assert bool(ctx.last_host_mode() & HOST_TANGENT_CPU) == (n >= 64), (n, ctx.last_host_mode())
    compare(got, (c.stress_out, c.tangent_out, c.hist_out), TOL[CLASS[kind]], f"{kind}/{c.name}")
    compare(got, (c.stress_out, c.tangent_out, c.hist_out), STRICT[CLASS[kind]], f"strict {kind}/{c.name}")


KINDS = ["linear_elasticity", "von_mises_3d", "spring_maxwell", "spring_kelvin", "comfe_linear_elasticity", "comfe_mises_plasticity"]


@pytest.mark.parametrize("n", [64, 1000, 70_001, 300_000])
@pytest.mark.parametrize("kind", KINDS)
def test_bit_identical_to_the_kernels_tangent(ctx, kind, n):
    ctx.set_option("host_tangent_min_points", 0)
    ctx.set_option("bounce_max", 0)
    ctx.set_option("host_tangent_chunk", 4096 if n < 100_000 else 0)  # several chunks and a ragged last one at every size
    p, g, s, h = random_case(kind, n, seed=n % 97 + len(kind))
    kernel, cpu = both_ways(ctx, make_law(kind, p), 2.0, g, s, h)
    assert_identical(kernel, cpu, f"{kind}/{n}")
    if CLASS[kind] == "pl":
        assert 0 < cpu[4].n_plastic < n, "the case must mix elastic and plastic points"


def test_unaligned_tangent_array(ctx):
    """the CPU writes the caller's tangent: an array off the 16-byte grid (a NumPy view at an odd element) needs no staged path"""
    ctx.set_option("host_tangent_min_points", 0)
    ctx.set_option("bounce_max", 0)
    for kind in ("linear_elasticity", "von_mises_3d"):
        p, g, s, h = random_case(kind, 20_000, seed=5)
        kernel, cpu = both_ways(ctx, make_law(kind, p), 1.0, g, s, h, tangent_offset=1)
        assert_identical(kernel, cpu, f"{kind}/unaligned")


@pytest.mark.parametrize("constraint", [SC.PLANE_STRAIN, SC.PLANE_STRESS, SC.UNIAXIAL_STRAIN, SC.UNIAXIAL_STRESS])
@pytest.mark.parametrize("cls", ["LinearElasticityModel", "SpringMaxwellModel", "SpringKelvinModel"])
def test_low_dimensional_constant_tangents(ctx, cls, constraint):
    ctx.set_option("host_tangent_min_points", 0)
    ctx.set_option("bounce_max", 0)
    p = {"E": 42.0, "nu": 0.3} if cls == "LinearElasticityModel" else {"E0": 42.0, "E1": 10.0, "tau": 10.0, "nu": 0.2}
    law = getattr(fc, cls)(p, constraint)
    n = 33_333
    rng = np.random.default_rng(11)
    gd2, sd = law.geometric_dim**2, law.stress_strain_dim
    g = rng.normal(scale=1e-3, size=gd2 * n)
    s = rng.normal(size=sd * n)
    h = None if law.history_dim is None else {k: rng.normal(scale=1e-3, size=d * n) for k, d in law.history_dim.items()}
    kernel, cpu = both_ways(ctx, law, 0.5, g, s, h)
    assert_identical(kernel, cpu, f"{cls}/{constraint}")


def test_small_calls_keep_the_kernels_stores(ctx):
    """below "host_tangent_min_points" (65536 by default) nothing changes"""
    assert ctx.get_option("host_tangent_min_points") == 1 << 16
    ctx.set_option("host_tangent_threads", 2)
    p, g, s, h = random_case("von_mises_3d", 5000, seed=1)
    t = np.full(36 * 5000, np.nan)
    make_law("von_mises_3d", p).evaluate(0.0, 1.0, g, s, t, h)
    assert not (ctx.last_host_mode() & HOST_TANGENT_CPU)
    assert ctx.get_option("last_host_tangent_threads") == 0 and not np.isnan(t).any()


@pytest.mark.parametrize("kind", ["von_mises_3d", "comfe_mises_plasticity", "linear_elasticity", "spring_maxwell"])
def test_resident_entry(ctx, kind):
    """ResidentState.evaluate_into (fcamd_evaluate_resident) without the sparse tangent: two Newton iterations and a commit"""
    from fenics_constitutive_amd.resident import ResidentState

    ctx.set_option("host_tangent_min_points", 0)
    ctx.set_option("bounce_max", 0)
    n = 100_003
    p, g, s, h = random_case(kind, n, seed=3)
    g2 = random_case(kind, n, seed=4)[1]
    results = []
    for th in (0, 3):
        ctx.set_option("host_tangent_threads", th)
        law = make_law(kind, p)
        st = ResidentState(law, n, stress0=s, history0=h, sparse_tangent=False, placement="torch")
        outs = []
        for step, grad in enumerate((g, g2, g)):
            so, to = np.full(6 * n, np.nan), np.full(36 * n, np.nan)
            st.evaluate_into(0.0, 2.0, grad, so, to)
            outs.append((so, to, ctx.last_host_mode()))
            if step == 1:
                st.update()
        results.append(outs)
    for (s0, t0, m0), (s1, t1, m1) in zip(*results):
        assert not (m0 & HOST_TANGENT_CPU) and (m1 & HOST_TANGENT_CPU), (m0, m1)
        assert np.array_equal(bits(t0), bits(t1)) and np.array_equal(bits(s0), bits(s1)), kind


def test_sparse_tangent_keeps_its_own_protocol(ctx):
    from fenics_constitutive_amd.resident import ResidentState

    ctx.set_option("host_tangent_min_points", 0)
    ctx.set_option("host_tangent_threads", 3)
    n = 70_000
    p, g, s, h = random_case("von_mises_3d", n, seed=8)
    st = ResidentState(make_law("von_mises_3d", p), n, stress0=s, history0=h, sparse_tangent=True, placement="torch")
    so, to = np.zeros(6 * n), np.zeros(36 * n)
    st.evaluate_into(0.0, 1.0, g, so, to)
    st.evaluate_into(0.0, 1.0, g, so, to)
    assert not (ctx.last_host_mode() & HOST_TANGENT_CPU)


@pytest.mark.parametrize("kind", ["von_mises_3d", "linear_elasticity"])
def test_ten_million_points(ctx, kind):
    """1e7 random points, default options (automatic thread count, automatic chunks): identical to the kernel's tangent, and the
    expansion's CPU time is reported"""
    n = 10_000_000
    p, g, s, h = random_case(kind, n, seed=21)
    threads = ctx.get_option("host_tangent_threads") or 4  # (0 = a box that grants this process fewer than 7 CPUs: the automatic choice is "off")
    law = make_law(kind, p)
    kernel, cpu = both_ways(ctx, law, 1.0, g, s, h, threads=threads)
    assert_identical(kernel, cpu, f"{kind}/1e7")
    assert ctx.get_option("last_host_tangent_threads") == threads
    assert ctx.get_option("last_host_tangent_cpu_us") > 0


def test_two_threads_two_contexts_expand_at_the_same_time(ctx):  # (ctx: the main thread's options are restored afterwards)
    """Every thread has a context of its own (thread-local), hence a pool and a parameter ring of its own: two threads inside their
    host entries at the same time -- the same gradient array read by both -- give what each gives alone."""
    import threading

    n = 150_001
    p, g, s, h = random_case("von_mises_3d", n, seed=77)
    ref = {}

    def run(tag, rounds):
        ctx = _capi.get_context(_capi.default_device())  # this thread's
        ctx.set_option("host_tangent_min_points", 0)
        ctx.set_option("host_tangent_threads", 3)
        ctx.set_option("host_tangent_chunk", 8192)
        law = make_law("von_mises_3d", p)
        out = None
        for _ in range(rounds):
            s1, t1 = s.copy(), np.full(36 * n, np.nan)
            h1 = {k: v.copy() for k, v in h.items()}
            law.evaluate(0.0, 1.0, g, s1, t1, h1)
            assert ctx.last_host_mode() & HOST_TANGENT_CPU
            if out is not None:
                assert np.array_equal(bits(out[1]), bits(t1))
            out = (s1, t1, h1)
        ref[tag] = out

    run("serial", 1)
    threads = [threading.Thread(target=run, args=(f"t{k}", 6)) for k in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
        assert not th.is_alive(), "a host entry with the tangent rebuilt on the host hangs under concurrency"
    for k in range(2):
        got = ref[f"t{k}"]
        assert np.array_equal(bits(got[0]), bits(ref["serial"][0])) and np.array_equal(bits(got[1]), bits(ref["serial"][1]))
        for name in h:
            assert np.array_equal(bits(got[2][name]), bits(ref["serial"][2][name]))


def test_one_process_several_contexts_share_the_cpus(ctx):
    """law.use_devices([...]) (fcamd_multi: one process, several device contexts, here on one GPU): every context expands the tangent of
    its own slice with its share of the host's threads; the result is the single-context result bit for bit."""
    n = 400_003
    p, g, s, h = random_case("von_mises_3d", n, seed=12)
    single = make_law("von_mises_3d", p)
    s1, t1, h1 = s.copy(), np.full(36 * n, np.nan), {k: v.copy() for k, v in h.items()}
    ctx.set_option("host_tangent_threads", 3)
    single.evaluate(0.0, 1.0, g, s1, t1, h1)
    assert ctx.last_host_mode() & HOST_TANGENT_CPU
    multi = make_law("von_mises_3d", p).use_devices([0, 0, 0])
    auto = multi._multi().get_option("host_tangent_threads")  # the usable CPUs shared by three contexts: 0 (too few to keep up) or 6 .. 16 each
    assert auto == 0 or 6 <= auto <= 16, auto
    multi._multi().set_option("host_tangent_threads", 3)
    s2, t2, h2 = s.copy(), np.full(36 * n, np.nan), {k: v.copy() for k, v in h.items()}
    multi.evaluate(0.0, 1.0, g, s2, t2, h2)
    mode, used = multi._multi().last_host_mode()
    assert used == 3 and (mode & HOST_TANGENT_CPU), (mode, used)
    assert np.array_equal(bits(t1), bits(t2)) and np.array_equal(bits(s1), bits(s2))
    for k in h:
        assert np.array_equal(bits(h1[k]), bits(h2[k]))


@pytest.mark.parametrize("n", [70_001, 300_000])
@pytest.mark.parametrize("hyper", [False, True])
def test_drucker_prager_laws(ctx, hyper, n):
    """The Drucker-Prager laws (comfe-rs general return mapping): the kernel sends 12 doubles per plastic point -- the five coefficients of
    the isotropic tangent form, the flag, rho s_tr -- and the tile's ballot; elastic points get elastic_tangent() itself."""
    from test_gpu_drucker_prager import make
    from test_oracle_golden import dp_inputs

    ctx.set_option("host_tangent_min_points", 0)
    ctx.set_option("bounce_max", 0)
    ctx.set_option("host_tangent_chunk", 8192 if n < 100_000 else 0)
    law, _ = make(hyper)
    g, s0, h0 = dp_inputs(n, n + 3)
    kernel, cpu = both_ways(ctx, law, 1.0, g, s0, h0)
    assert_identical(kernel, cpu, f"drucker_prager hyper={hyper} n={n}")
    assert 0 < cpu[4].n_plastic < n


def test_drucker_prager_resident_entry(ctx):
    from fenics_constitutive_amd.resident import ResidentState
    from test_gpu_drucker_prager import make
    from test_oracle_golden import dp_inputs

    ctx.set_option("host_tangent_min_points", 0)
    ctx.set_option("bounce_max", 0)
    n = 90_005
    g, s0, h0 = dp_inputs(n, 11)
    outs = []
    for th in (0, 3):
        ctx.set_option("host_tangent_threads", th)
        law, _ = make(True)
        st = ResidentState(law, n, stress0=s0, history0=h0, sparse_tangent=False, placement="torch")
        so, to = np.full(6 * n, np.nan), np.full(36 * n, np.nan)
        st.evaluate_into(0.0, 1.0, g, so, to)
        outs.append((so, to, ctx.last_host_mode()))
    assert not (outs[0][2] & HOST_TANGENT_CPU) and (outs[1][2] & HOST_TANGENT_CPU)
    assert np.array_equal(bits(outs[0][0]), bits(outs[1][0])) and np.array_equal(bits(outs[0][1]), bits(outs[1][1]))


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_sizes_chunks_threads(ctx, seed):
    """random law, size (ragged tails, fewer chunks than ring slots, more chunks than slots), chunk size (0 = automatic with its tapered
    tail), thread count and stream count: always bit for bit the kernel's tangent"""
    from test_gpu_drucker_prager import make
    from test_oracle_golden import dp_inputs

    rng = np.random.default_rng(1000 + seed)
    ctx.set_option("host_tangent_min_points", 0)
    ctx.set_option("bounce_max", 0)
    kind = ["linear_elasticity", "von_mises_3d", "spring_maxwell", "spring_kelvin", "comfe_linear_elasticity", "comfe_mises_plasticity",
            "dp_classic", "dp_hyperbolic"][int(rng.integers(0, 8))]
    n = int(rng.choice([64, 65, 127, 4096, 4097, int(rng.integers(64, 20_000)), int(rng.integers(20_000, 400_000))]))
    chunk = int(rng.choice([0, 64, 128, 1024, 4096 + 64 * int(rng.integers(0, 64)), 65536, 131072]))
    ctx.set_option("host_tangent_chunk", chunk)
    ctx.set_option("host_tangent_streams", int(rng.integers(1, 4)))
    try:
        if kind.startswith("dp_"):
            law, _ = make(kind == "dp_hyperbolic")
            g, s, h = dp_inputs(n, seed)
        else:
            p, g, s, h = random_case(kind, n, seed=seed)
            law = make_law(kind, p)
        kernel, cpu = both_ways(ctx, law, 1.5, g, s, h, threads=int(rng.integers(1, 6)))
        assert_identical(kernel, cpu, f"fuzz {seed}: {kind} n={n} chunk={chunk}")
    finally:
        ctx.set_option("host_tangent_streams", 1)


def test_nonconvergence_is_reported_and_leaves_no_work_behind(ctx):
    """A Newton iteration that does not converge (reference: RuntimeError inside evaluate, mises_plasticity_isotropic_hardening.py:141-143)
    on the pipelined path: the error surfaces after the last chunk, the pool is idle again, and the next call works."""
    from test_oracle_c import NONCONVERGING, nonconverging_inputs

    ctx.set_option("host_tangent_min_points", 0)
    ctx.set_option("host_tangent_threads", 3)
    ctx.set_option("host_tangent_chunk", 8192)
    n = 70_017
    bad = fc.VonMises3D(NONCONVERGING)
    g, s, t, h = nonconverging_inputs(n)
    with pytest.raises(RuntimeError, match="did not converge"):
        bad.evaluate(0, 1.0, g, s, t, h)
    assert ctx.last_host_mode() & HOST_TANGENT_CPU
    p, g, s, h = random_case("von_mises_3d", n, seed=2)
    kernel, cpu = both_ways(ctx, make_law("von_mises_3d", p), 1.0, g, s, h)
    assert_identical(kernel, cpu, "after a failed call")


@pytest.mark.parametrize("kind", ["von_mises_3d", "comfe_mises_plasticity", "linear_elasticity"])
def test_non_finite_points(ctx, kind):
    """NaN / Inf / huge inputs at a few points (tests/test_gpu_special_values.py's poison): the other points' rows are bit for bit the
    kernel's, the poisoned points' rows are non-finite where the kernel's are (the bit pattern of a NaN is the arithmetic unit's: the
    host's SSE unit and the GPU propagate payloads differently) and bit-identical where finite."""
    from test_gpu_special_values import POISON

    ctx.set_option("host_tangent_min_points", 0)
    ctx.set_option("bounce_max", 0)
    ctx.set_option("host_tangent_chunk", 4096)
    n = 30_011
    p, g, s, h = random_case(kind, n, seed=19)
    bad = np.array([3, 64, 130, 4191, 8200, 20_301, n - 1])
    for pt, v in zip(bad, POISON):
        g.reshape(-1, 9)[pt, pt % 9] = v
    s.reshape(-1, 6)[bad[0], 2] = float("nan")
    outs = []
    for th in (0, 3):
        ctx.set_option("host_tangent_threads", th)
        s1, t1 = s.copy(), np.full(36 * n, -1.0)
        h1 = None if h is None else {k: v.copy() for k, v in h.items()}
        try:
            make_law(kind, p).evaluate(0.0, 0.7, g, s1, t1, h1)
        except RuntimeError:  # non-convergence at a poisoned point is allowed to raise -- after the whole call has been written
            pass
        outs.append((s1, t1, ctx.last_host_mode()))
    (s0, t0, m0), (s1, t1, m1) = outs
    assert not (m0 & HOST_TANGENT_CPU) and (m1 & HOST_TANGENT_CPU)
    ok = np.setdiff1d(np.arange(n), bad)
    assert np.array_equal(bits(t0.reshape(-1, 36)[ok]), bits(t1.reshape(-1, 36)[ok])), "clean points differ"
    assert np.array_equal(np.isnan(t0), np.isnan(t1)) and np.array_equal(np.isinf(t0), np.isinf(t1))
    fin = np.isfinite(t0)
    assert np.array_equal(bits(t0[fin]), bits(t1[fin]))
    assert np.array_equal(bits(s0.reshape(-1, 6)[ok]), bits(s1.reshape(-1, 6)[ok]))
