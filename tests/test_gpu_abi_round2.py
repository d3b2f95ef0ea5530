"""C-ABI additions of round 2, through ctypes on the GPU: the getters of the reference's native model
classes (bindings/src/lib.rs:137-148), timing on every entry (the counterpart of the reference's
Timer("constitutive-law-evaluation"), solver/_lawonsubmesh.py:86), context options, thread-local
contexts, VMM working sets."""

import ctypes as C
import gc
import threading

import numpy as np
import pytest
import torch

import fenics_constitutive_amd as fc
from fenics_constitutive_amd import _capi

pytestmark = pytest.mark.gpu

FULL = fc.StressStrainConstraint.FULL
VM_P = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}


def test_model_getters_match_the_python_properties():
    cases = [fc.VonMises3D(VM_P), fc.LinearElasticityModel({"E": 42.0, "nu": 0.3}, fc.StressStrainConstraint.PLANE_STRESS),
             fc.SpringMaxwellModel({"E0": 42.0, "E1": 10.0, "tau": 10.0, "nu": 0.2}, fc.StressStrainConstraint.UNIAXIAL_STRAIN),
             fc.MisesPlasticityLinearHardening3D({k: np.array([v]) for k, v in
                                                  {"mu": 1.0, "kappa": 2.0, "y_0": 3.0, "h": 4.0}.items()})]
    for law in cases:
        m = law._handle(0)
        assert m.constraint == law.constraint.value
        assert m.dims == (law.stress_strain_dim, law.geometric_dim)
        assert dict(m.history_fields) == (law.history_dim or {})


def _vm_arrays(n, seed=0):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    f = dict(dtype=torch.float64, device="cuda")
    g = torch.randn(9 * n, generator=gen, **f) * 3e-3
    return g, torch.zeros(6 * n, **f), torch.zeros(6 * n, **f), torch.empty(36 * n, **f), \
        {"eps_n": torch.zeros(6 * n, **f), "alpha": torch.zeros(n, **f)}, \
        {"eps_n": torch.zeros(6 * n, **f), "alpha": torch.zeros(n, **f)}


def test_timing_applies_to_every_device_entry_and_to_the_host_entries():
    law = fc.VonMises3D(VM_P)
    n = 1 << 20
    g, s0, s1, t, h0, h1 = _vm_arrays(n)
    m = law._handle(0)
    with pytest.raises(ValueError):  # timing off: nothing to report
        law.evaluate(0, 1.0, g, s1, t, h1)
        m.last_kernel_ms()
    m.ctx.set_timing(True)
    try:
        mask = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
        rows = torch.arange(n, dtype=torch.int32, device="cuda")
        entries = {
            "device": lambda: law.evaluate(0, 1.0, g, s1, t, h1),
            "from": lambda: law.evaluate_from(0, 1.0, g, s0, s1, t, h0, h1),
            "from_sparse": lambda: law.evaluate_from(0, 1.0, g, s0, s1, t, h0, h1, history_mask=mask),
            "ex": lambda: law.evaluate_from(0, 1.0, g, s0, s1, t, h0, h1, history_mask=mask, sparse_tangent=True),
            "indexed": lambda: law.evaluate_indexed(0, 1.0, g, s0, s1, t, rows, h0, h1),
        }
        for name, call in entries.items():
            call()
            ms = m.last_kernel_ms()
            assert 0.01 < ms < 50.0, (name, ms)
        # fused wrapper entry
        w = fc.PlaneStrainFrom3D(law)
        f = dict(dtype=torch.float64, device="cuda")
        w.evaluate(0, 1.0, torch.randn(4 * n, **f) * 1e-3, torch.zeros(4 * n, **f), torch.empty(16 * n, **f),
                   {"eps_n": torch.zeros(6 * n, **f), "alpha": torch.zeros(n, **f)})
        assert 0.005 < m.last_kernel_ms() < 50.0
        # host entry: the wall-clock of the synchronous call
        gh = g.cpu().numpy()
        law.evaluate(0, 1.0, gh, np.zeros(6 * n), np.empty(36 * n), {"eps_n": np.zeros(6 * n), "alpha": np.zeros(n)})
        assert m.last_kernel_ms() > 1.0
    finally:
        m.ctx.set_timing(False)


def test_context_options_replace_the_environment_knobs():
    law = fc.VonMises3D(VM_P)
    ctx = law._handle(0).ctx
    assert ctx.get_option("masked_max") == -1
    with pytest.raises(ValueError, match="unknown option"):
        ctx.set_option("no_such_knob", 1)
    n = 64 * 300 + 5
    g, s0, s1, t, h0, h1 = _vm_arrays(n, seed=3)
    law.evaluate_from(0, 1.0, g, s0, s1, t, h0, h1)
    ref = (s1.clone(), t.clone(), h1["eps_n"].clone())
    try:
        for name, value in (("masked_max", 0), ("masked_max", 64)):
            ctx.set_option(name, value)
            assert ctx.get_option(name) == value
            s1.zero_(), t.zero_(), h1["eps_n"].zero_()
            law.evaluate_from(0, 1.0, g, s0, s1, t, h0, h1)
            assert torch.equal(s1, ref[0]) and torch.equal(t, ref[1]) and torch.equal(h1["eps_n"], ref[2]), (name, value)
    finally:
        ctx.set_option("masked_max", -1)
    ctx.trim()  # staging buffers released; the next pageable host call allocates them again
    gh = g.cpu().numpy()
    sh, th = np.zeros(6 * n), np.empty(36 * n)
    law.evaluate(0, 1.0, gh, sh, th, {"eps_n": np.zeros(6 * n), "alpha": np.zeros(n)})
    assert np.array_equal(sh, ref[0].cpu().numpy())


def test_contexts_die_with_their_thread():
    """a worker thread's context (streams, ~1 GB of staging buffers) is released when
    the thread ends instead of staying in a dict keyed by a recyclable thread ident."""
    law = fc.LinearElasticityModel({"E": 42.0, "nu": 0.3}, FULL)
    n = 3000
    g = np.random.default_rng(0).normal(size=9 * n)
    seen = []

    def work():
        s, t = np.zeros(6 * n), np.empty(36 * n)
        law.evaluate(0.0, 1.0, g, s, t, None)
        ctx = _capi.get_context(_capi.default_device())
        seen.append((ctx, s))

    for _ in range(3):
        th = threading.Thread(target=work)
        th.start()
        th.join()
    assert law.n_handles_created == 3
    import weakref

    assert len({id(pair[0]) for pair in seen}) == 3  # one context per thread, none inherited
    assert all(np.array_equal(pair[1], seen[0][1]) for pair in seen)
    refs = [weakref.ref(pair[0]) for pair in seen]
    seen.clear()
    gc.collect()
    assert all(r() is None for r in refs)  # nothing keeps a dead thread's context alive
    assert _capi.get_context(_capi.default_device()) is _capi.get_context(_capi.default_device())


@pytest.mark.parametrize("interleaved", [False, True])
def test_vmm_working_set_is_usable_memory(interleaved):
    """fcamd_device_alloc_set: arrays placed through hipMemAddressReserve / hipMemCreate / hipMemMap take
    part in an evaluate like any other device memory (wrapped as torch tensors without a copy)."""
    from fenics_constitutive_amd.placement import VmmArraySet

    law = fc.VonMises3D(VM_P)
    n = 64 * 1000 + 9
    g, s0, s1, t, h0, h1 = _vm_arrays(n, seed=11)
    law.evaluate_from(0, 1.0, g, s0, s1, t, h0, h1)
    ctx = law._handle(0).ctx
    aset = VmmArraySet(ctx, {"tangent": 36 * n, "stress": 6 * n, "eps_n": 6 * n}, interleaved=interleaved)
    try:
        tv, sv, ev = aset["tangent"], aset["stress"], aset["eps_n"]
        assert tv.numel() == 36 * n and tv.is_cuda and tv.dtype == torch.float64
        assert tv.data_ptr() % (2 << 20) == 0
        tv.zero_(), sv.zero_(), ev.zero_()
        law.evaluate_from(0, 1.0, g, s0, sv, tv, h0, {"eps_n": ev, "alpha": h1["alpha"]})
        torch.cuda.synchronize()
        assert torch.equal(tv, t) and torch.equal(sv, s1) and torch.equal(ev, h1["eps_n"])
    finally:
        del tv, sv, ev
        aset.free()
    with pytest.raises(ValueError):
        ctx.free(12345 * 4096)  # not one of ours


def test_raw_ctypes_new_entries():
    lib = _capi.load()
    lo, hi = C.c_int64(), C.c_int64()
    slot = C.c_int64()
    assert lib.fcamd_shard_bounds(10**8 * 8, 8, 7, C.byref(lo), C.byref(hi), C.byref(slot)) == 0
    assert (lo.value, hi.value, slot.value) == (7 * 10**8, 8 * 10**8, 10**8)
    assert lib.fcamd_shard_bounds(10, 0, 0, C.byref(lo), C.byref(hi), None) == _capi.ERR_BAD_ARG
    ctx = _capi.get_context(0)
    v = C.c_longlong()
    assert lib.fcamd_context_get_option(ctx.handle, b"host_slots", C.byref(v)) == 0 and v.value == 4
    assert lib.fcamd_model_get_info(None, None) == _capi.ERR_BAD_ARG
    assert lib.fcamd_context_get_option(ctx.handle, b"last_host_mode", C.byref(v)) == 0
    assert lib.fcamd_context_set_option(ctx.handle, b"last_host_mode", 1) == _capi.ERR_BAD_ARG  # read-only
    assert lib.fcamd_copy(ctx.handle, None, None, 8, 99) == _capi.ERR_BAD_ARG
