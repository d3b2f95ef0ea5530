"""fcamd_evaluate_batch: the laws of one form() in one call (the reference calls them back to back, solver/_solver.py:143-144).
Bit for bit the same state as one fcamd_evaluate_device_ex per law; a refused call launches nothing."""

import ctypes as C

import numpy as np
import pytest
from test_gpu_parity import make_law, random_case

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from fenics_constitutive_amd import _capi  # noqa: E402
from fenics_constitutive_amd.problem import ResidentProblemState, rows_of_cells  # noqa: E402

KINDS = ["linear_elasticity", "von_mises_3d", "spring_maxwell", "spring_kelvin", "comfe_mises_plasticity", "linear_elasticity",
         "von_mises_3d", "comfe_linear_elasticity"]


def build(n_cells, q, n_laws, seed, batch, big_first=False):
    rng = np.random.default_rng(seed)
    if big_first:  # one law that fills the device (context stream) next to small ones (side streams)
        owner = np.where(rng.random(n_cells) < 0.85, 0, rng.integers(1, n_laws, size=n_cells))
    else:
        owner = rng.integers(0, n_laws, size=n_cells)
    rows = [rows_of_cells(np.flatnonzero(owner == k), q) for k in range(n_laws)]
    kinds = KINDS[:n_laws]
    cases = [random_case(kind, r.size, seed=seed + 7 * i) for i, (kind, r) in enumerate(zip(kinds, rows))]
    laws = [make_law(kind, c[0]) for kind, c in zip(kinds, cases)]
    n = q * n_cells
    st = ResidentProblemState(list(zip(laws, rows)), n, del_t=0.7, batch_launches=batch, placement="torch")
    stress_0 = np.zeros(6 * n)
    for r, c in zip(rows, cases):
        stress_0.reshape(-1, 6)[r] = c[2].reshape(-1, 6)
    st.set_state(stress_0, [None if c[3] is None else {k: v.copy() for k, v in c[3].items()} for c in cases])
    return st, rows, rng


@pytest.mark.parametrize("n_laws,n_cells,big_first", [(2, 700, False), (8, 3001, False), (8, 20000, False), (4, 400000, True)])
def test_batched_form_equals_law_by_law(n_laws, n_cells, big_first):
    q = 4
    a, rows, rng = build(n_cells, q, n_laws, 5, True, big_first)
    b, _, _ = build(n_cells, q, n_laws, 5, False, big_first)
    assert a.batch_launches and not b.batch_launches
    for inc in range(2):
        for it in range(3):
            hi = -2.0 if (inc + it) % 2 == 0 else -3.0
            grads = [torch.from_numpy(rng.normal(size=9 * r.size) * np.repeat(10 ** rng.uniform(-4, hi, size=r.size), 9)).cuda() for r in rows]
            a.evaluate(grads), b.evaluate(grads)
            a.check(), b.check()
            assert torch.equal(a.stress_1, b.stress_1) and torch.equal(a.tangent, b.tangent)
            for ha, hb in zip(a._history_1, b._history_1):
                for k in (ha or {}):
                    assert torch.equal(ha[k], hb[k]), k
        a.update(), b.update()
        assert torch.equal(a.stress_0, b.stress_0)


def test_batch_entry_raw_checks_everything_before_it_launches_anything():
    lib = _capi.load()
    n = 64 * 50 + 7
    kinds = ["linear_elasticity", "spring_maxwell", "von_mises_3d"]
    cases = [random_case(k, n, seed=2 + i) for i, k in enumerate(kinds)]
    laws = [make_law(k, c[0]) for k, c in zip(kinds, cases)]
    f = dict(dtype=torch.float64, device="cuda")
    models = [law._handle(0) for law in laws]
    stream = torch.cuda.current_stream().cuda_stream
    models[0].ctx.set_stream(stream)
    assert all(m.ctx is models[0].ctx for m in models)
    keep, args, ref = [], [], []
    for law, m, c in zip(laws, models, cases):
        g = torch.from_numpy(c[1]).cuda()
        s0, s1, t = torch.from_numpy(c[2]).cuda(), torch.zeros(6 * n, **f), torch.zeros(36 * n, **f)
        h0 = [] if c[3] is None else [torch.from_numpy(c[3][name]).cuda() for name, _ in m.history_fields]
        h1 = [torch.zeros_like(h) for h in h0]
        a0, a1 = (C.c_void_p * max(1, len(h0)))(*[h.data_ptr() for h in h0]), (C.c_void_p * max(1, len(h1)))(*[h.data_ptr() for h in h1])
        args.append(_capi.EvalArgs(g.data_ptr(), s0.data_ptr(), s1.data_ptr(), t.data_ptr(), a0, a1, len(h0), None, None, 0, None, None, None, None, 0, None))
        keep.append((g, s0, s1, t, h0, h1, a0, a1))
        # the same call on its own, into arrays of its own
        s1r, tr, h1r = torch.zeros(6 * n, **f), torch.zeros(36 * n, **f), [torch.zeros_like(h) for h in h0]
        m.evaluate_device_ex(0.0, 0.7, n, g.data_ptr(), s0.data_ptr(), s1r.data_ptr(), tr.data_ptr(), [h.data_ptr() for h in h0], [h.data_ptr() for h in h1r])
        ref.append((s1r, tr, h1r))
    mh = (C.c_void_p * 3)(*[m.handle for m in models])
    ns = (C.c_int64 * 3)(n, n, n)
    xs = (_capi.EvalArgs * 3)(*args)
    # the third call is refused (del_t <= 0 is checked per call; here: a NULL stress): nothing may have been written
    bad = (_capi.EvalArgs * 3)(*args)
    bad[2].stress = None
    rc = lib.fcamd_evaluate_batch(3, mh, ns, bad, 0.0, 0.7)
    assert rc != _capi.OK
    torch.cuda.synchronize()
    assert all(float(k[2].abs().max()) == 0.0 and float(k[3].abs().max()) == 0.0 for k in keep)
    with pytest.raises(AssertionError):  # the reference's exception type for del_t <= 0, through the same status mapping
        _capi.check(lib.fcamd_evaluate_batch(3, mh, ns, xs, 0.0, 0.0))
    _capi.check(lib.fcamd_evaluate_batch(3, mh, ns, xs, 0.0, 0.7))
    _capi.check(lib.fcamd_evaluate_batch(0, None, None, None, 0.0, 0.7))  # an empty form()
    torch.cuda.synchronize()
    for k, (s1r, tr, h1r) in zip(keep, ref):
        assert torch.equal(k[2], s1r) and torch.equal(k[3], tr)
        for h, hr in zip(k[5], h1r):
            assert torch.equal(h, hr)
    st = laws[2].device_stats(0)
    assert st.n_plastic > 0  # the batch resets and fills the law's own counters like the single call


def test_timed_context_launches_the_laws_one_by_one():
    """context option "timing": fcamd_model_last_stats must keep reporting each law's own kernel time"""
    a, rows, rng = build(600, 4, 2, 9, True)
    ctx = a._laws[0].law._handle(0).ctx
    grads = [torch.from_numpy(rng.normal(size=9 * r.size) * 1e-3).cuda() for r in rows]
    ctx.set_timing(True)
    try:
        a.evaluate(grads)
        a.check()
        for ls in a._laws:
            assert ls.law._handle(0).last_kernel_ms() > 0.0
    finally:
        ctx.set_timing(False)
    b, _, _ = build(600, 4, 2, 9, False)
    b.evaluate(grads)
    assert torch.equal(a.stress_1, b.stress_1) and torch.equal(a.tangent, b.tangent)


def test_resident_states_join_a_callers_batch():
    """two ResidentStates (two single-law problems) evaluated inside the caller's own batched_launches(): one fcamd_evaluate_batch
    for both, same trial states as evaluated one after the other"""
    from fenics_constitutive_amd.resident import ResidentState

    def states():
        out = []
        for i, kind in enumerate(("von_mises_3d", "spring_maxwell")):
            p, g, s, h = random_case(kind, 64 * 90 + 11, seed=21 + i)
            st = ResidentState(make_law(kind, p), s.size // 6, stress0=s, history0=h, placement="torch")
            out.append((st, torch.from_numpy(g).cuda()))
        return out

    a, b = states(), states()
    for it in range(3):
        grads = [g * (1.0 + 0.1 * it) for _, g in a]  # recorded calls launch when the block ends: their arrays stay alive until then
        with _capi.batched_launches() as batch:
            for (st, _), g in zip(a, grads):
                st.evaluate(0.0, 0.7, g)
        assert len(batch.prepared) == 1 and batch.prepared[0].count == 2
        for (st, _), g in zip(b, grads):
            st.evaluate(0.0, 0.7, g)
        for (sa, _), (sb, _) in zip(a, b):
            sa.check(), sb.check()
            assert torch.equal(sa.stress, sb.stress) and torch.equal(sa.tangent, sb.tangent)
    for (sa, _), (sb, _) in zip(a, b):
        sa.update(), sb.update()
        assert torch.equal(sa.stress_committed, sb.stress_committed)


def test_many_states_on_one_thread_keep_their_tables():
    """twelve resident states evaluated in turn, again and again (one context: more distinct batch tables than a handful of slots):
    every state keeps computing what a state with the batch kernel off computes"""
    from fenics_constitutive_amd.resident import ResidentState

    p, g, s, h = random_case("von_mises_3d", 64 * 40 + 5, seed=33)
    law = make_law("von_mises_3d", p)
    ctx = law._handle(0).ctx
    sts = [ResidentState(law, s.size // 6, stress0=s, history0=h, placement="torch") for _ in range(12)]
    ref = ResidentState(law, s.size // 6, stress0=s, history0=h, placement="torch")
    gd = torch.from_numpy(g).cuda()
    for it in range(4):
        gi = gd * (1.0 + 0.05 * it)
        for st in sts:
            st.evaluate(0.0, 1.0, gi)
        ctx.set_option("batch_kernel", 0)
        try:
            ref._launch_cache.clear()
            ref.evaluate(0.0, 1.0, gi)
        finally:
            ctx.set_option("batch_kernel", 1)
        ref.check()
        for st in sts:
            st.check()
            assert torch.equal(st.stress, ref.stress) and torch.equal(st.tangent, ref.tangent)
        if it == 1:
            ref.update()
            for st in sts:
                st.update()


def test_batch_table_uploaded_on_one_stream_launched_from_another():
    """the first evaluate uploads the batch table on stream A; the next iteration, the same call, comes from stream B: it waits for the upload"""
    a, rows, rng = build(900, 4, 3, 17, True)
    b, _, _ = build(900, 4, 3, 17, False)
    grads = [torch.from_numpy(rng.normal(size=9 * r.size) * np.repeat(10 ** rng.uniform(-4, -2, size=r.size), 9)).cuda() for r in rows]
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(sa):
        a.evaluate(grads)
    with torch.cuda.stream(sb):
        sb.wait_stream(sa)
        a.evaluate(grads)
        a.check()
    b.evaluate(grads), b.evaluate(grads)
    b.check()
    assert torch.equal(a.stress_1, b.stress_1) and torch.equal(a.tangent, b.tangent)
