"""Host <-> device copies and pageable caller arrays (csrc/fcamd_hostpath.cpp: CallerArrays; hostio.py): the package never
hands pageable memory to the HIP runtime's copy path, whose cache of on-the-fly page locks goes stale when memory is
freed and allocated again at the same address.  The reference's counterpart of these copies are plain
NumPy assignments (solver/_history.py:64-79, solver/_lawonsubmesh.py:58-61)."""

import ctypes as C
import mmap

import numpy as np
import pytest
import torch

import fenics_constitutive_amd as fc
from fenics_constitutive_amd import _capi, hostio

pytestmark = pytest.mark.gpu

VM_P = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}
BOUNCE, TEMP, ZC = _capi.HOST_BOUNCE, _capi.HOST_TEMP_LOCK, _capi.HOST_ZERO_COPY_IN | _capi.HOST_ZERO_COPY_OUT


@pytest.mark.parametrize("numel", [0, 1, 1000, (256 << 10) // 8, (256 << 10) // 8 + 1, 3_000_001])
def test_upload_download_round_trip(numel):
    """sizes on both sides of the scratch threshold (256 KiB), ragged, empty"""
    rng = np.random.default_rng(numel)
    a = rng.normal(size=numel)
    d = hostio.to_device(a, "cuda")
    assert d.dtype == torch.float64 and d.numel() == numel
    assert np.array_equal(hostio.to_host(d), a)
    b = np.full(numel, np.nan)
    hostio.download(b, d * 2.0)
    assert np.array_equal(b, 2.0 * a)
    e = torch.zeros(numel, dtype=torch.float64, device="cuda")
    hostio.upload(e, b)
    assert torch.equal(e, d * 2.0)


def test_other_dtypes_views_and_registered_arrays():
    rows = np.arange(50_000, dtype=np.int64)[::-1]                      # non-contiguous source, converted on the host
    r = hostio.to_device(rows, "cuda", np.int32)
    assert r.dtype == torch.int32 and np.array_equal(hostio.to_host(r), rows.astype(np.int32))
    t = torch.arange(12, dtype=torch.float64, device="cuda")
    m = np.zeros((4, 6))[:, :3]                                         # non-contiguous destination: through a temporary
    hostio.assign(m, t)
    assert np.array_equal(m, np.arange(12.0).reshape(4, 3))
    # a registered range is used as it is (no lock, no scratch)
    ctx = _capi.get_context(_capi.default_device())
    buf = np.frombuffer(mmap.mmap(-1, 8 * 400_000), dtype=np.float64)
    ctx.register_host_buffer(buf)
    try:
        buf[:] = np.arange(400_000)
        d = hostio.to_device(buf[100:300_000], "cuda")
        assert np.array_equal(hostio.to_host(d), buf[100:300_000])
        hostio.download(buf[:299_900], d + 1.0)
        assert np.array_equal(buf[:299_900], np.arange(100, 300_000) + 1.0)
    finally:
        ctx.unregister_host_buffer(buf)
    with pytest.raises(AssertionError):
        hostio.upload(torch.zeros(3, dtype=torch.float64, device="cuda"), np.zeros(4))


def _libc():
    libc = C.CDLL(None, use_errno=True)
    libc.mmap.restype = C.c_void_p
    libc.mmap.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_long]
    libc.munmap.argtypes = [C.c_void_p, C.c_size_t]
    return libc


def test_new_memory_at_an_old_address_is_locked_again():
    """The scenario behind the GPU memory faults of round 2: arrays are evaluated, FREED, and new arrays appear at the
    same addresses (what malloc does all the time).  A cache of page locks keyed by address calls the new memory locked;
    the host entries lock what is there at the time of the call instead.  munmap + mmap(MAP_FIXED) makes the reuse certain."""
    libc = _libc()
    PROT_RW, MAP_PRIVATE, MAP_ANON, MAP_FIXED = 3, 2, 0x20, 0x10
    law = fc.VonMises3D(VM_P)
    ctx = law._handle(_capi.default_device()).ctx
    n = 20_000  # 9.3 MB per call: above the scratch threshold
    sizes = [9 * n, 6 * n, 36 * n, 6 * n, n]
    nbytes = [((8 * k + 4095) // 4096) * 4096 for k in sizes]
    total = sum(nbytes)
    rng = np.random.default_rng(1)
    g = rng.normal(size=9 * n) * 3e-3
    ref = None
    base = libc.mmap(None, total, PROT_RW, MAP_PRIVATE | MAP_ANON, -1, 0)
    assert base not in (None, C.c_void_p(-1).value)
    try:
        for round_ in range(3):
            arrs, off = [], 0
            for k, nb in zip(sizes, nbytes):
                arrs.append(np.frombuffer((C.c_double * k).from_address(base + off), dtype=np.float64))
                off += nb
            arrs[0][:] = g
            arrs[2][:] = np.nan
            law.evaluate(0.0, 1.0, arrs[0], arrs[1], arrs[2], {"eps_n": arrs[3], "alpha": arrs[4]})
            assert ctx.last_host_mode() == (ZC | TEMP)
            out = (arrs[1].copy(), arrs[2].copy(), arrs[3].copy(), arrs[4].copy())
            if ref is None:
                ref = out
            assert all(np.array_equal(a, b) for a, b in zip(out, ref)), round_
            # the same through the copy entries
            d = hostio.to_device(arrs[2], "cuda")
            assert np.array_equal(hostio.to_host(d), ref[1])
            del arrs, d
            # free the memory and put NEW memory at the same address
            assert libc.munmap(base, total) == 0
            again = libc.mmap(base, total, PROT_RW, MAP_PRIVATE | MAP_ANON | MAP_FIXED, -1, 0)
            assert again == base
    finally:
        libc.munmap(base, total)


def test_resident_state_uploads_are_synchronous():
    """ResidentState / ResidentProblemState copy a caller's NumPy gradient before they return: the caller may overwrite
    or free it right away (round 1 queued an asynchronous copy from the caller's pageable memory)."""
    from fenics_constitutive_amd.problem import ResidentProblemState
    from fenics_constitutive_amd.resident import ResidentState

    law = fc.VonMises3D(VM_P)
    n = 150_000  # 10.8 MB gradient
    rng = np.random.default_rng(2)
    g = rng.normal(size=9 * n) * 3e-3
    a, b = ResidentState(law, n), ResidentState(law, n)
    pa, pb = ResidentProblemState(law, n, del_t=1.0), ResidentProblemState(law, n, del_t=1.0)
    g_dev = hostio.to_device(g, "cuda")
    scratch = g.copy()
    a.evaluate(0.0, 1.0, scratch)
    pa.evaluate([scratch])
    scratch[:] = 1e9  # overwritten immediately after the call returns
    del scratch
    b.evaluate(0.0, 1.0, g_dev)
    pb.evaluate([g_dev])
    torch.cuda.synchronize()
    assert torch.equal(a.stress, b.stress) and torch.equal(a.tangent, b.tangent)
    assert torch.equal(pa.stress_1, pb.stress_1) and torch.equal(pa.tangent, pb.tangent)


@pytest.mark.parametrize("numel", [1, 2, 1023, 1024 * 1024 + 3, 20_000_001])
def test_device_copy_kernel(numel):
    """fcamd_copy_device (the evaluate kernels' access pattern with nothing but the copy): every byte, ragged tails, a
    destination that is not 16-byte aligned (falls back to torch's copy), nothing outside the range."""
    src = torch.arange(numel + 4, dtype=torch.float64, device="cuda")
    dst = torch.full((numel + 4,), -1.0, dtype=torch.float64, device="cuda")
    hostio.copy_device(dst[2 : 2 + numel], src[2 : 2 + numel])     # both 16-byte aligned (offset 16 bytes)
    torch.cuda.synchronize()
    assert torch.equal(dst[2 : 2 + numel], src[2 : 2 + numel])
    assert (dst[:2] == -1.0).all() and (dst[2 + numel :] == -1.0).all()
    dst.fill_(-1.0)
    hostio.copy_device(dst[1 : 1 + numel], src[2 : 2 + numel])     # destination 8 bytes off the grid
    torch.cuda.synchronize()
    assert torch.equal(dst[1 : 1 + numel], src[2 : 2 + numel]) and dst[0] == -1.0 and (dst[1 + numel :] == -1.0).all()
    ctx = _capi.get_context(_capi.default_device())
    with pytest.raises(ValueError):
        ctx.copy_device(dst.data_ptr() + 8, src.data_ptr(), 64)
