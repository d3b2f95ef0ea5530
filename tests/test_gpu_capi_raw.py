"""The C ABI used directly through ctypes, exactly as INTEGRATION.md section 2 shows a maintainer
binding it (no fenics_constitutive_amd Python classes on the call path): context / model
lifecycle, history introspection, host evaluate, status codes and error strings."""

import ctypes as C
import os

import numpy as np
import pytest
from golden_util import load_calls, rel_err

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")  # one HIP runtime per process: torch first

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fenics-constitutive_amd", "lib", "libfcamd.so")


@pytest.fixture(scope="module")
def lib():
    from fenics_constitutive_amd import _build

    _build.build_library()
    l = C.CDLL(LIB)
    l.fcamd_last_error.restype = C.c_char_p
    return l


def test_integration_stub_von_mises(lib):
    ctx, mdl = C.c_void_p(), C.c_void_p()
    assert lib.fcamd_context_create(0, None, C.byref(ctx)) == 0
    p = (C.c_double * 5)(175000.0, 80769.0, 1200.0, 2500.0, 200.0)
    assert lib.fcamd_model_create(ctx, 2, 5, p, 5, C.byref(mdl)) == 0  # FCAMD_VON_MISES_3D, FCAMD_FULL
    class Info(C.Structure):  # fcamd_model_info
        _fields_ = [("model_id", C.c_int), ("constraint", C.c_int), ("stress_strain_dim", C.c_int), ("geometric_dim", C.c_int),
                    ("n_history", C.c_int), ("history_name", C.c_char_p * 2), ("history_dim", C.c_int * 2)]

    info = Info()
    assert lib.fcamd_model_get_info(mdl, C.byref(info)) == 0 and info.n_history == 2
    names = [(info.history_name[k].decode(), info.history_dim[k]) for k in range(2)]
    assert names == [("eps_n", 6), ("alpha", 1)]  # history_dim of the reference (:184-186)
    assert (info.model_id, info.constraint, info.stress_strain_dim, info.geometric_dim) == (2, 5, 6, 3)

    class Stats(C.Structure):  # fcamd_stats
        _fields_ = [("nonconv", C.c_uint64), ("plastic", C.c_uint64), ("iters", C.c_uint64), ("domain", C.c_uint64), ("kernel_ms", C.c_double)]

    for c in load_calls("von_mises_3d.npz")[:6]:
        s, t, h = c.fresh()
        g = c.grad.copy()
        hp = (C.c_void_p * 2)(h["eps_n"].ctypes.data, h["alpha"].ctypes.data)
        st = Stats()
        rc = lib.fcamd_evaluate_host(mdl, C.c_double(0.0), C.c_double(c.del_t), C.c_int64(c.n), C.c_void_p(g.ctypes.data),
                                     C.c_void_p(s.ctypes.data), C.c_void_p(t.ctypes.data), hp, 2, C.byref(st))
        assert rc == 0, lib.fcamd_last_error()
        assert rel_err(s, c.stress_out) <= 1e-6 and rel_err(t, c.tangent_out) <= 1e-6
        assert rel_err(h["alpha"], c.hist_out["alpha"]) <= 1e-6
        assert st.plastic == int(np.sum(c.hist_out["alpha"] > c.hist_in["alpha"]))
    assert lib.fcamd_model_destroy(mdl) == 0 and lib.fcamd_context_destroy(ctx) == 0


def test_status_codes(lib):
    ctx, mdl = C.c_void_p(), C.c_void_p()
    assert lib.fcamd_context_create(0, None, C.byref(ctx)) == 0
    assert lib.fcamd_context_create(99, None, C.byref(C.c_void_p())) == 6  # FCAMD_ERR_BAD_ARG
    p4 = (C.c_double * 4)(42.0, 10.0, 10.0, 0.2)
    assert lib.fcamd_model_create(ctx, 77, 5, p4, 4, C.byref(mdl)) == 6  # unknown law
    assert lib.fcamd_model_create(ctx, 2, 3, p4, 4, C.byref(mdl)) == 8  # VonMises3D is FULL-only -> UNSUPPORTED
    assert lib.fcamd_model_create(ctx, 3, 5, p4, 3, C.byref(mdl)) == 6  # wrong parameter count
    assert lib.fcamd_model_create(ctx, 3, 5, p4, 4, C.byref(mdl)) == 0  # SpringMaxwellModel FULL
    n = 10
    g, s, t = np.zeros(9 * n), np.zeros(6 * n), np.zeros(36 * n)
    h = [np.zeros(6 * n), np.zeros(6 * n)]
    hp = (C.c_void_p * 2)(h[0].ctypes.data, h[1].ctypes.data)
    args = (C.c_int64(n), C.c_void_p(g.ctypes.data), C.c_void_p(s.ctypes.data), C.c_void_p(t.ctypes.data))
    assert lib.fcamd_evaluate_host(mdl, C.c_double(0), C.c_double(1.0), *args, None, 0, None) == 2  # NULL_HISTORY
    assert b"history must not be None" in lib.fcamd_last_error()
    assert lib.fcamd_evaluate_host(mdl, C.c_double(0), C.c_double(0.0), *args, hp, 2, None) == 3  # DEL_T
    assert lib.fcamd_evaluate_host(mdl, C.c_double(0), C.c_double(1.0), *args, hp, 1, None) == 1  # SIZE
    assert lib.fcamd_evaluate_host(mdl, C.c_double(0), C.c_double(1.0), *args, hp, 2, None) == 0
    assert lib.fcamd_evaluate_host(mdl, C.c_double(0), C.c_double(1.0), C.c_int64(0), None, None, None, hp, 2, None) == 0  # n = 0
    # the device entry (fcamd_evaluate_device_ex; in place: prev == out) rejects misaligned pointers
    from fenics_constitutive_amd._capi import EvalArgs

    d = torch.zeros(64 * 60, dtype=torch.float64, device="cuda")
    base = d.data_ptr()
    lib.fcamd_evaluate_device_ex.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int64, C.POINTER(EvalArgs)]
    hd = (C.c_void_p * 2)(base + 8 * 1024, base + 8 * 2048)

    def in_place(grad, stress):
        return EvalArgs(grad, stress, stress, None, hd, hd, 2, None, None, 0, None, None, None, None, 0, None)

    assert lib.fcamd_evaluate_device_ex(mdl, 0.0, 1.0, 8, C.byref(in_place(base + 8, base + 8 * 512))) == 7  # ALIGN
    assert lib.fcamd_evaluate_device_ex(mdl, 0.0, 1.0, 8, C.byref(in_place(base, base + 8 * 512))) == 0
    bad = in_place(base, base + 8 * 512)
    bad.flags = 2  # FCAMD_EVAL_DELTA_HISTORY of ABI 0.3: removed
    assert lib.fcamd_evaluate_device_ex(mdl, 0.0, 1.0, 8, C.byref(bad)) == 8  # UNSUPPORTED
    assert lib.fcamd_context_synchronize(ctx) == 0
    lib.fcamd_model_destroy(mdl)
    lib.fcamd_context_destroy(ctx)


def test_c_program_runs(tmp_path, lib):
    """examples/c_caller.c: a plain-C host, compiled with gcc, evaluates through the C ABI."""
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(LIB)
    exe = tmp_path / "c_caller"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "c_caller.c"),
                    "-o", str(exe), "-L", libdir, "-lfcamd", "-lm", f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "max error" in r.stdout


def test_c_program_drives_several_device_contexts(tmp_path, lib):
    """examples/c_caller_multi.c: a plain-C host runs the single-process multi-GPU entry (three contexts) and the
    multi-device resident state; its own checks compare with the single-device entry bit for bit."""
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(LIB)
    exe = tmp_path / "c_caller_multi"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "c_caller_multi.c"),
                    "-o", str(exe), "-L", libdir, "-lfcamd", "-lm", f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined"], check=True)
    r = subprocess.run([str(exe), "3"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "3 device contexts (3 used" in r.stdout and "bit for bit" in r.stdout


def test_c_program_batches_the_laws_of_one_form(tmp_path, lib):
    """examples/c_caller_batch.c: a plain-C host (no HIP runtime of its own: device memory and copies through the C ABI) evaluates
    two materials on interleaved cells of one mesh as ONE fcamd_evaluate_batch, twice (the second call replays the kept table)."""
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(LIB)
    exe = tmp_path / "c_caller_batch"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "c_caller_batch.c"),
                    "-o", str(exe), "-L", libdir, "-lfcamd", "-lm", f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "2 laws, 3004 + 3000 points in one call" in r.stdout and "0 plastic points" in r.stdout


def test_host_mapping_and_flag_errors(lib):
    """fcamd_host_device_pointer / FCAMD_EVAL_SPARSE_TANGENT argument checks, raw."""
    import mmap

    from fenics_constitutive_amd import _capi

    ctx, mdl = C.c_void_p(), C.c_void_p()
    assert lib.fcamd_context_create(0, None, C.byref(ctx)) == 0
    a = np.frombuffer(mmap.mmap(-1, 1 << 20), dtype=np.float64)
    dptr = C.c_void_p()
    lib.fcamd_host_device_pointer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    assert lib.fcamd_host_device_pointer(ctx, C.c_void_p(a.ctypes.data), a.nbytes, C.byref(dptr)) == _capi.ERR_BAD_ARG
    assert b"registered" in lib.fcamd_last_error()
    lib.fcamd_register_host_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    assert lib.fcamd_register_host_buffer(ctx, C.c_void_p(a.ctypes.data), a.nbytes) == 0
    assert lib.fcamd_host_device_pointer(ctx, C.c_void_p(a.ctypes.data), a.nbytes, C.byref(dptr)) == 0 and dptr.value
    sub = C.c_void_p()
    assert lib.fcamd_host_device_pointer(ctx, C.c_void_p(a.ctypes.data + 4096), 8192, C.byref(sub)) == 0
    assert sub.value - dptr.value == 4096                                       # sub-range of one registration
    assert lib.fcamd_host_device_pointer(ctx, C.c_void_p(a.ctypes.data + 8), 64, C.byref(sub)) == _capi.ERR_BAD_ARG  # alignment
    assert lib.fcamd_host_device_pointer(ctx, C.c_void_p(a.ctypes.data), a.nbytes + 8, C.byref(sub)) == _capi.ERR_BAD_ARG
    lib.fcamd_unregister_host_buffer.argtypes = [C.c_void_p, C.c_void_p]
    assert lib.fcamd_unregister_host_buffer(ctx, C.c_void_p(a.ctypes.data)) == 0
    assert lib.fcamd_host_device_pointer(ctx, C.c_void_p(a.ctypes.data), a.nbytes, C.byref(dptr)) == _capi.ERR_BAD_ARG
    # sparse tangent needs the history mask and a tangent array
    p = (C.c_double * 5)(175000.0, 80769.0, 1200.0, 2500.0, 200.0)
    assert lib.fcamd_model_create(ctx, 2, 5, p, 5, C.byref(mdl)) == 0
    n = 128
    f = dict(dtype=torch.float64, device="cuda")
    g, s0, s1, t = torch.zeros(9 * n, **f), torch.zeros(6 * n, **f), torch.zeros(6 * n, **f), torch.zeros(36 * n, **f)
    e0, a0, e1, a1 = torch.zeros(6 * n, **f), torch.zeros(n, **f), torch.zeros(6 * n, **f), torch.zeros(n, **f)
    hp = (C.c_void_p * 2)(e0.data_ptr(), a0.data_ptr())
    hc = (C.c_void_p * 2)(e1.data_ptr(), a1.data_ptr())
    x = _capi.EvalArgs(g.data_ptr(), s0.data_ptr(), s1.data_ptr(), t.data_ptr(), hp, hc, 2, None, None,
                       _capi.EVAL_SPARSE_TANGENT, None)
    lib.fcamd_evaluate_device_ex.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int64, C.POINTER(_capi.EvalArgs)]
    assert lib.fcamd_evaluate_device_ex(mdl, 0.0, 1.0, n, C.byref(x)) == _capi.ERR_BAD_ARG
    assert b"history_mask" in lib.fcamd_last_error()
    mask = torch.zeros(2, dtype=torch.int64, device="cuda")
    x.history_mask = mask.data_ptr()
    assert lib.fcamd_evaluate_device_ex(mdl, 0.0, 1.0, n, C.byref(x)) == 0
    assert lib.fcamd_context_synchronize(ctx) == 0
    assert lib.fcamd_model_destroy(mdl) == 0 and lib.fcamd_context_destroy(ctx) == 0
