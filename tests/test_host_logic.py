"""CPU tests of the host side: interface mirror, helpers, the C-ABI library's exports, the
loud failure without a GPU.  No compute calls (no GPU here)."""

import ctypes
import os
import re

import numpy as np
import pytest

import fenics_constitutive_amd as fc
from fenics_constitutive_amd import _capi
from fenics_constitutive_amd.sharded import ShardPlan

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = fc.StressStrainConstraint


def test_constraint_enum_matches_reference_values():
    # models/interfaces.py:14-73
    assert [c.value for c in (C.UNIAXIAL_STRAIN, C.UNIAXIAL_STRESS, C.PLANE_STRAIN, C.PLANE_STRESS, C.FULL)] == [1, 2, 3, 4, 5]
    assert [c.stress_strain_dim for c in C] == [1, 1, 4, 4, 6]
    assert [c.geometric_dim for c in C] == [1, 1, 2, 2, 3]


def test_interface_is_abstract():
    with pytest.raises(TypeError):
        fc.IncrSmallStrainModel()


def test_model_properties_without_gpu():
    vm = fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0})
    assert vm.constraint == C.FULL and vm.history_dim == {"eps_n": 6, "alpha": 1}
    assert vm.stress_strain_dim == 6 and vm.geometric_dim == 3
    assert np.allclose(vm.xpp, np.eye(6) - np.pad(np.ones((3, 3)), (0, 3)) / 3)
    le = fc.LinearElasticityModel({"E": 42.0, "nu": 0.3}, C.FULL)
    assert le.history_dim is None and le.D.shape == (6, 6)
    for cls in (fc.SpringMaxwellModel, fc.SpringKelvinModel):
        m = cls({"E0": 42.0, "E1": 10.0, "tau": 10.0, "nu": 0.2}, C.FULL)
        assert m.history_dim == {"strain_visco": 6, "strain": 6}
        m1 = cls({"E0": 42.0, "E1": 10.0, "tau": 10.0}, C.UNIAXIAL_STRESS)  # nu forced to 0 (:31-34)
        assert m1.nu == 0.0 and m1.history_dim == {"strain_visco": 1, "strain": 1}
    rs = fc.MisesPlasticityLinearHardening3D({k: np.array([v]) for k, v in
                                              {"mu": 1.0, "kappa": 2.0, "y_0": 3.0, "h": 4.0}.items()})
    assert rs.history_dim == {"history": 7} and rs.constraint == C.FULL
    assert fc.LinearElasticity3D({"mu": np.array([1.0]), "kappa": np.array([2.0])}).history_dim is None
    with pytest.raises(KeyError):  # Python-style dict handed to the Rust-style class (test_plasticity.py:33-37)
        fc.MisesPlasticityLinearHardening3D({"p_ka": 1.0})


def test_elastic_tangent_all_constraints():
    E, nu = 42.0, 0.3
    mu, lam = fc.lame_parameters(E, nu)
    assert mu == E / (2.0 * (1.0 + nu)) and lam == E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu))
    D = fc.get_elastic_tangent(E, nu, C.FULL)
    assert D.shape == (6, 6) and D[0, 0] == 2.0 * mu + lam and D[0, 1] == lam and D[3, 3] == 2.0 * mu and D[0, 3] == 0
    assert np.array_equal(fc.get_elastic_tangent(E, nu, C.PLANE_STRAIN), D[:4, :4])
    Dps = fc.get_elastic_tangent(E, nu, C.PLANE_STRESS)
    assert np.isclose(Dps[0, 0], E / (1 - nu**2)) and Dps[2, 2] == 0 and np.isclose(Dps[3, 3], E / (1 + nu))
    assert fc.get_elastic_tangent(E, nu, C.UNIAXIAL_STRESS)[0, 0] == E
    assert np.isclose(fc.get_elastic_tangent(E, nu, C.UNIAXIAL_STRAIN)[0, 0], lam + 2 * mu)
    assert list(fc.get_identity(6, C.FULL)) == [1, 1, 1, 0, 0, 0]
    assert list(fc.get_identity(4, C.PLANE_STRESS)) == [1, 1, 0, 0]
    assert list(fc.get_identity(1, C.UNIAXIAL_STRAIN)) == [1]


def test_strain_from_grad_u_needs_the_gpu_for_every_constraint():
    """No host-side arithmetic in the product: without a HIP device every constraint raises
    (the values are checked on the GPU: tests/test_gpu_parity.py::test_strain_from_grad_u_low_dimensional)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    for c, g in ((C.UNIAXIAL_STRAIN, np.array([[1.0]])), (C.PLANE_STRESS, np.array([[1.0, 2.0], [3.0, 4.0]]))):
        with pytest.raises(RuntimeError):
            fc.strain_from_grad_u(g, c)


def test_library_exports_every_declared_symbol():
    import subprocess

    core = open(os.path.join(ROOT, "include", "fcamd.h")).read()
    multi = open(os.path.join(ROOT, "include", "fcamd_multi.h")).read()
    pat = r"^FCAMD_API [a-z_ *]*?\b(fcamd_[a-z_0-9]+)\s*\("
    core_syms, multi_syms = set(re.findall(pat, core, re.M)), set(re.findall(pat, multi, re.M))
    # the measured core (one evaluate per data path -- bindings/src/lib.rs:76-129 has one per model -- + the batch of one form()) and
    # the multi-GPU forms (fcamd_multi.h: never run on a multi-GPU node) are kept apart
    assert len(core_syms) <= 26 and len(multi_syms) <= 20 and not (core_syms & multi_syms)
    assert all(s.startswith(("fcamd_multi_", "fcamd_ipc_", "fcamd_allgather_", "fcamd_shard_", "fcamd_gather_")) for s in multi_syms)
    assert not any(s.startswith(("fcamd_multi_", "fcamd_ipc_", "fcamd_allgather_")) for s in core_syms)
    assert "UNMEASURED ON MULTI-GPU HARDWARE" in multi
    hdr = core + multi
    declared = sorted(core_syms | multi_syms)
    assert declared == sorted(_capi.SYMBOLS)
    lib = ctypes.CDLL(_capi.library_path()) if os.path.exists(_capi.library_path()) else _capi.load()
    for name in declared:
        assert hasattr(lib, name), name
    # ... and nothing else: the library's dynamic symbol table holds exactly the declared entries
    nm = subprocess.run(["nm", "-D", "--defined-only", _capi.library_path()], capture_output=True, text=True)
    if nm.returncode == 0:
        exported = sorted(ln.split()[-1] for ln in nm.stdout.splitlines() if " T fcamd_" in ln)
        assert exported == declared
    # the header's static inline shorthands (ABI 0.3 names) are not symbols
    inline = set(re.findall(r"^FCAMD_INLINE [a-z_ *]*?\b(fcamd_[a-z_0-9]+)\s*\(", hdr, re.M))
    assert {"fcamd_evaluate_device", "fcamd_evaluate_device_from", "fcamd_evaluate_device_indexed", "fcamd_status_string"} <= inline
    assert not (inline & set(declared))
    major, minor = (int(re.search(rf"#define FCAMD_VERSION_{k} (\d+)", hdr).group(1)) for k in ("MAJOR", "MINOR"))
    assert _capi.load().fcamd_version() == 1000 * major + minor
    assert _capi.status_string(4).startswith("Newton-Raphson")
    # the ctypes module's table of status texts is the header's fcamd_status_string, case by case
    body = hdr[hdr.index("FCAMD_INLINE const char* fcamd_status_string"):]
    for code, text in re.findall(r'case (FCAMD_[A-Z_]+): return "([^"]+)";', body[: body.index("default:")]):
        value = int(re.search(rf"{code} = (\d+)", hdr).group(1))
        assert _capi.STATUS_STRINGS[value] == text, code


def test_no_cpu_fallback():
    """Without a HIP device the product path raises; it never computes on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    le = fc.LinearElasticityModel({"E": 42.0, "nu": 0.3}, C.FULL)
    s = np.ones(6)
    with pytest.raises(RuntimeError, match="no ROCm-capable device|hipGetDeviceCount"):
        le.evaluate(0.0, 1.0, np.ones(9), s, np.zeros(36), None)
    assert np.array_equal(s, np.ones(6))
    with pytest.raises(RuntimeError):
        fc.strain_from_grad_u(np.ones(9), C.FULL)


def test_validation_happens_before_any_device_work():
    le = fc.LinearElasticityModel({"E": 42.0, "nu": 0.3}, C.FULL)
    with pytest.raises(AssertionError):
        le.evaluate(0.0, 1.0, np.zeros(18), np.zeros(6), np.zeros(72), None)
    sls = fc.SpringMaxwellModel({"E0": 42.0, "E1": 10.0, "tau": 10.0, "nu": 0.2}, C.FULL)
    with pytest.raises(ValueError, match="history must not be None"):
        sls.evaluate(0.0, 1.0, np.zeros(9), np.zeros(6), np.zeros(36), None)
    with pytest.raises(AssertionError, match="Time step"):
        sls.evaluate(0.0, 0.0, np.zeros(9), np.zeros(6), np.zeros(36),
                     {"strain_visco": np.zeros(6), "strain": np.zeros(6)})


@pytest.mark.parametrize("n,world", [(0, 2), (1, 2), (64, 2), (65, 2), (1000, 3), (10**8, 8), (8 * 10**8, 8), (129, 8)])
def test_shard_plan(n, world):
    plan = ShardPlan.create(n, world)
    assert plan.per_rank % 64 == 0 and plan.per_rank * world >= n
    covered = 0
    for r in range(world):
        lo, hi = plan.bounds(r)
        assert lo == min(r * plan.per_rank, n) and lo <= hi <= n
        assert lo % 64 == 0 or lo == n
        covered += hi - lo
    assert covered == n


def test_header_is_plain_c_and_c_caller_links(tmp_path):
    """include/fcamd.h must compile as C99 and a C program must link against libfcamd.so (running it
    needs a GPU: tests/test_gpu_capi_raw.py)."""
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    _capi.load()
    exe = tmp_path / "c_caller"
    libdir = os.path.dirname(_capi.library_path())
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "c_caller.c"), "-o", str(exe), "-L", libdir, "-lfcamd", "-lm",
                        f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert exe.exists()


def test_default_device_selection(monkeypatch):
    import sys

    for v in ("FCAMD_DEVICE", "LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "SLURM_LOCALID"):
        monkeypatch.delenv(v, raising=False)
    import torch

    if torch.cuda.is_available():
        assert _capi.default_device() == torch.cuda.current_device()
        # a launcher's local rank maps onto the visible GPUs modulo their number (more ranks than GPUs share them) ...
        monkeypatch.setenv("OMPI_COMM_WORLD_LOCAL_RANK", str(torch.cuda.device_count() + 1))
        if torch.cuda.current_device() == 0:
            assert _capi.default_device() == 1 % torch.cuda.device_count()
    else:
        assert _capi.default_device() == 0 and _capi.device_count() == 0
        monkeypatch.setenv("OMPI_COMM_WORLD_LOCAL_RANK", "3")
        assert _capi.default_device() == 3
        monkeypatch.setenv("LOCAL_RANK", "5")
        assert _capi.default_device() == 5
    monkeypatch.setenv("FCAMD_DEVICE", "2")
    assert _capi.default_device() == 2


def test_devices_of_the_single_process_multi_gpu_path(monkeypatch):
    """FCAMD_DEVICES is the environment default of every law's use_devices (an unchanged dolfinx script on a whole node);
    the multi handle itself is created lazily, at the first ndarray evaluate."""
    import fenics_constitutive_amd as fc
    from fenics_constitutive_amd import _capi

    monkeypatch.delenv("FCAMD_DEVICES", raising=False)
    assert _capi.default_devices() is None
    law = fc.LinearElasticityModel({"E": 1.0, "nu": 0.3}, fc.StressStrainConstraint.FULL)
    assert law.devices is None and law._multi_handle is None
    assert law.use_devices([2, 0]) is law and law.devices == [2, 0] and law._multi_handle is None
    assert law.use_devices(None).devices is None
    monkeypatch.setenv("FCAMD_DEVICES", "0, 3,1")
    assert _capi.default_devices() == [0, 3, 1]
    assert fc.VonMises3D({"p_ka": 1.0, "p_mu": 1.0, "p_y0": 1.0, "p_y00": 2.0, "p_w": 1.0}).devices == [0, 3, 1]
    monkeypatch.setenv("FCAMD_DEVICES", "")
    assert _capi.default_devices() is None


def test_multi_handle_rejects_bad_arguments_without_touching_a_device():
    import ctypes as C

    from fenics_constitutive_amd import _capi

    lib = _capi.load()
    h = C.c_void_p()
    p = (C.c_double * 2)(1.0, 0.3)
    assert lib.fcamd_multi_create(None, 2, _capi.LINEAR_ELASTICITY, 5, p, 2, C.byref(h)) == _capi.ERR_BAD_ARG
    dv = (C.c_int * 1)(0)
    assert lib.fcamd_multi_create(dv, 0, _capi.LINEAR_ELASTICITY, 5, p, 2, C.byref(h)) == _capi.ERR_BAD_ARG
    assert lib.fcamd_multi_create(dv, _capi.MULTI_MAX_DEVICES + 1, _capi.LINEAR_ELASTICITY, 5, p, 2, C.byref(h)) == _capi.ERR_BAD_ARG
    assert not h.value
    assert lib.fcamd_multi_destroy(None) == _capi.OK and lib.fcamd_multi_state_destroy(None) == _capi.OK
    assert lib.fcamd_multi_evaluate_host(None, 0.0, 1.0, 0, None, None, None, None, 0, None) == _capi.ERR_BAD_ARG
    assert lib.fcamd_multi_state_commit(None) == _capi.ERR_BAD_ARG


def test_rows_of_cells_matches_quadrature_numbering():
    """Quadrature dofs are numbered cell by cell (solver/maps.py:43-79): cell c, point q -> c*Q + q."""
    from fenics_constitutive_amd.problem import rows_of_cells

    rows = rows_of_cells(np.array([5, 0, 2]), 3)
    assert rows.dtype == np.int32 and rows.tolist() == [15, 16, 17, 0, 1, 2, 6, 7, 8]
    assert rows_of_cells(np.array([], dtype=np.int32), 4).size == 0


def test_eval_args_struct_layout_matches_header():
    """ctypes mirror of fcamd_eval_args: sixteen fields in the header's order, pointer-sized except n_hist / flags / wrapper_constraint."""
    import ctypes as C
    import re

    from fenics_constitutive_amd import _capi

    hdr = open(os.path.join(ROOT, "include", "fcamd.h")).read()
    body = re.search(r"typedef struct fcamd_eval_args \{(.*?)\} fcamd_eval_args;", hdr, re.S).group(1)
    names = re.findall(r"(\w+);", body)
    assert names == [f[0] for f in _capi.EvalArgs._fields_]
    assert C.sizeof(_capi.EvalArgs) == 16 * C.sizeof(C.c_void_p)  # the three ints each padded to pointer size
    info = re.search(r"typedef struct fcamd_model_info \{(.*?)\} fcamd_model_info;", hdr, re.S).group(1)
    assert re.findall(r"(\w+)(?:\[\w+\])?;", info) == [f[0] for f in _capi.ModelInfo._fields_]
    stats = re.search(r"typedef struct fcamd_stats \{(.*?)\} fcamd_stats;", hdr, re.S).group(1)
    assert re.findall(r"^\s*\w+ (\w+);", stats, re.M) == [f[0] for f in _capi.Stats._fields_] and C.sizeof(_capi.Stats) == 40
    assert _capi.EvalArgs.flags.offset == 9 * C.sizeof(C.c_void_p)


def test_header_constants_match_the_ctypes_module():
    """Every flag / status / enum value include/fcamd.h defines and _capi.py mirrors has the same value on both sides."""
    import re

    from fenics_constitutive_amd import _capi

    hdr = open(os.path.join(ROOT, "include", "fcamd.h")).read() + open(os.path.join(ROOT, "include", "fcamd_multi.h")).read()
    defines = {k: int(v, 0) for k, v in re.findall(r"#define (FCAMD_[A-Z_0-9]+) (\d+|0x[0-9a-fA-F]+)\b", hdr)}
    pairs = {"FCAMD_EVAL_SPARSE_TANGENT": _capi.EVAL_SPARSE_TANGENT,
             "FCAMD_HOST_ZERO_COPY_IN": _capi.HOST_ZERO_COPY_IN, "FCAMD_HOST_ZERO_COPY_OUT": _capi.HOST_ZERO_COPY_OUT,
             "FCAMD_HOST_TEMP_LOCK": _capi.HOST_TEMP_LOCK, "FCAMD_HOST_BOUNCE": _capi.HOST_BOUNCE, "FCAMD_HOST_TANGENT_CPU": _capi.HOST_TANGENT_CPU,
             "FCAMD_COUNTER_SLOTS": _capi.COUNTER_SLOTS,
             "FCAMD_MAX_HISTORY": _capi.MAX_HISTORY, "FCAMD_IPC_HANDLE_BYTES": _capi.IPC_HANDLE_BYTES,
             "FCAMD_GATHER_PULL": _capi.GATHER_PULL, "FCAMD_ALLOC_SEQUENTIAL": _capi.ALLOC_SEQUENTIAL,
             "FCAMD_ALLOC_INTERLEAVED": _capi.ALLOC_INTERLEAVED, "FCAMD_ALLOC_IPC": _capi.ALLOC_IPC,
             "FCAMD_COPY_TO_DEVICE": _capi.COPY_TO_DEVICE, "FCAMD_COPY_TO_HOST": _capi.COPY_TO_HOST, "FCAMD_COPY_DEVICE": _capi.COPY_DEVICE, "FCAMD_EVAL_SPLIT_HISTORY": _capi.EVAL_SPLIT_HISTORY,
             "FCAMD_EVAL_PACKED_HISTORY": _capi.EVAL_PACKED_HISTORY,
             "FCAMD_MULTI_MAX_DEVICES": _capi.MULTI_MAX_DEVICES, "FCAMD_MULTI_MIN_POINTS": _capi.MULTI_MIN_POINTS}
    for name, value in pairs.items():
        assert defines[name] == value, name
    assert _capi.COUNTER_WORDS == 4 * defines["FCAMD_COUNTER_SLOTS"]  # FCAMD_COUNTER_WORDS is an expression in the header


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference checkout not present")
def test_drop_in_mode_subclasses_the_reference_interface():
    """With the reference importable (here: its NumPy part, dolfinx / the compiled bindings stubbed as
    in SURVEY Appendix A) the GPU-backed laws are subclasses of the REFERENCE ABC and carry the
    REFERENCE enum, which is what IncrSmallStrainProblem checks (solver/_solver.py:67-75)."""
    import subprocess
    import sys
    import textwrap

    code = textwrap.dedent(
        """
        import sys, types
        df = types.ModuleType("dolfinx"); common = types.ModuleType("dolfinx.common")
        common.timed = lambda name: (lambda f: f); df.common = common
        sys.modules["dolfinx"] = df; sys.modules["dolfinx.common"] = common
        b = types.ModuleType("fenics_constitutive._bindings")
        for n in ("PyDruckerPrager3D", "PyDruckerPragerHyperbolic3D", "PyLinearElasticity3D", "PyMisesPlasticity3D"):
            setattr(b, n, type(n, (), {}))
        sys.modules["fenics_constitutive._bindings"] = b
        sys.path.insert(0, "/root/reference/src")
        sys.path.insert(0, %r)
        import fenics_constitutive.models as ref
        import fenics_constitutive_amd as fc
        from fenics_constitutive_amd import interfaces
        assert interfaces.REFERENCE_INTERFACES
        assert fc.IncrSmallStrainModel is ref.IncrSmallStrainModel and fc.StressStrainConstraint is ref.StressStrainConstraint
        vm = fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0})
        le = fc.LinearElasticityModel({"E": 42.0, "nu": 0.3}, ref.StressStrainConstraint.PLANE_STRESS)
        assert isinstance(vm, ref.IncrSmallStrainModel) and isinstance(le, ref.IncrSmallStrainModel)
        assert vm.constraint is ref.StressStrainConstraint.FULL and le.constraint is ref.StressStrainConstraint.PLANE_STRESS
        assert vm.history_dim == ref.VonMises3D(dict(p_ka=1.0, p_mu=1.0, p_y0=1.0, p_y00=2.0, p_w=1.0)).history_dim
        assert le.stress_strain_dim == 4 and le.geometric_dim == 2
        print("drop-in ok")
        """
    ) % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "drop-in ok" in r.stdout, r.stderr[-2000:]


def test_parent_rows_of_a_subspace_map():
    """integration._parent_rows: SubSpaceMap (solver/maps.py:98-100: parent_array[parent] = sub_array[sub])
    read as local point sub[i] -> parent row parent[i]; IdentityMap -> None."""
    from types import SimpleNamespace

    from fenics_constitutive_amd.integration import _parent_rows

    class IdentityMap:
        pass

    assert _parent_rows(SimpleNamespace(submesh_map=IdentityMap()), 5) is None
    parent = np.array([7, 3, 9, 0])
    sub = np.array([2, 0, 3, 1])
    rows = _parent_rows(SimpleNamespace(submesh_map=SimpleNamespace(parent=parent, sub=sub)), 4)
    # map_to_parent with these rows == the reference's fancy-index statement
    sub_array = np.arange(4.0) * 10
    a = np.zeros(10)
    a[parent] = sub_array[sub]
    b = np.zeros(10)
    b[rows] = sub_array
    assert np.array_equal(a, b) and rows.tolist() == [3, 0, 7, 9]


def test_kernel_hash_ignores_comments_but_not_code():
    """profiles/traffic.json is keyed by the hash of the device code: a comment or whitespace edit must not invalidate a
    stored PMC measurement, any change of the code must."""
    from fenics_constitutive_amd import _build

    def tree(patch):
        def read(f):
            with open(os.path.join(_build.CSRC, f)) as fh:
                text = fh.read()
            return patch(text) if f == "fcamd_kernels.hip" else text
        return read

    base = _build.kernel_hash()
    assert _build.kernel_hash(tree(lambda t: t)) == base
    assert _build.kernel_hash(tree(lambda t: t.replace("namespace fcamd {", "namespace fcamd {  // a remark\n\n   /* and\n another */", 1))) == base
    assert _build.kernel_hash(tree(lambda t: t.replace("return 512 * num_cu;", "return 256 * num_cu;", 1))) != base
    assert "-pthread" not in _build.KERNEL_FLAGS and "-ffp-contract=off" in _build.KERNEL_FLAGS


@pytest.mark.parametrize("die", [True, False])
def test_bench_line_guard_prints_exactly_one_line(die):
    """bench.py's measured line survives the death of the process in an optional leg: the guard child prints the last
    line it was handed (provisional, marked incomplete) -- or the final one, and never both."""
    import json
    import subprocess
    import sys
    import textwrap

    code = textwrap.dedent(f"""
        import json, os, sys
        sys.path.insert(0, {ROOT!r})
        import bench
        g = bench.LineGuard()
        g.provisional({{"value": 1}}, "leg a")
        g.provisional({{"value": 2}}, "leg b")
        if {die!r}:
            os.kill(os.getpid(), 9)
        g.final(json.dumps({{"value": 3}}))
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    got = json.loads(lines[0])
    if die:
        assert r.returncode != 0 and got["value"] == 2 and "leg b" in got["incomplete"]
    else:
        assert r.returncode == 0 and got == {"value": 3}


def test_reference_import_paths_resolve():
    """every name of the reference's ``fenics_constitutive.models`` package and of its submodules (models/__init__.py:7-23,
    models/utils.py:8-15, models/interfaces.py:8-11, models/rust_models.py:85-145) is importable from the same path under
    ``fenics_constitutive_amd.models``"""
    import importlib

    paths = {
        "": ["LinearElasticityModel", "MisesPlasticityLinearHardening3D", "SpringKelvinModel", "SpringMaxwellModel", "VonMises3D",
             "IncrSmallStrainModel", "StressStrainConstraint", "lame_parameters", "get_elastic_tangent", "get_identity",
             "strain_from_grad_u", "UniaxialStrainFrom3D", "PlaneStrainFrom3D"],
        ".interfaces": ["IncrSmallStrainModel", "StressStrainConstraint"],
        ".utils": ["lame_parameters", "get_elastic_tangent", "get_identity", "strain_from_grad_u", "UniaxialStrainFrom3D", "PlaneStrainFrom3D"],
        ".linear_elasticity_model": ["LinearElasticityModel"],
        ".mises_plasticity_isotropic_hardening": ["VonMises3D"],
        ".spring_maxwell_model": ["SpringMaxwellModel"],
        ".spring_kelvin_model": ["SpringKelvinModel"],
        ".rust_models": ["LinearElasticity3D", "DruckerPrager3D", "DruckerPragerHyperbolic3D", "MisesPlasticityLinearHardening3D"],
    }
    for sub, names in paths.items():
        mod = importlib.import_module("fenics_constitutive_amd.models" + sub)
        for name in names:
            assert getattr(mod, name) is getattr(fc, name), (sub, name)


def test_small_ndarray_calls_warn_once(monkeypatch):
    """Below the measured crossover (bench.py: cpu_baseline.small_call_crossover) one ndarray call costs more
    than a NumPy evaluation of the law on the host: the first such call of a law warns, once; device tensors never do."""
    import warnings

    from fenics_constitutive_amd import device as fdev

    monkeypatch.delenv("FCAMD_SMALL_CALL_WARNING", raising=False)
    le = fc.LinearElasticityModel({"E": 42.0, "nu": 0.3}, C.FULL)
    fdev._small_call_warned.discard("LinearElasticityModel")
    limit = fdev.SMALL_CALL_POINTS["LinearElasticityModel"]
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        fdev._warn_small_call(le, limit)      # at the limit: nothing
        fdev._warn_small_call(le, 0)          # empty calls: nothing
    with pytest.warns(RuntimeWarning, match="fixed cost"):
        fdev._warn_small_call(le, limit - 1)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        fdev._warn_small_call(le, 10)         # once per law
    fdev._small_call_warned.discard("LinearElasticityModel")
    monkeypatch.setenv("FCAMD_SMALL_CALL_WARNING", "0")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        fdev._warn_small_call(le, 10)         # silenced
    assert set(fdev.SMALL_CALL_POINTS) >= {"LinearElasticityModel", "VonMises3D", "SpringMaxwellModel", "SpringKelvinModel"}


def test_scripts_compile():
    """bench.py, benchlib/, tools/, examples/ and the soak scripts are not imported by the CPU suite: at least they must parse"""
    import glob
    import py_compile

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]
    for d in ("benchlib", "tools", "examples", "oracle", "tests"):
        files += glob.glob(os.path.join(root, d, "*.py"))
    assert len(files) > 60
    for f in files:
        py_compile.compile(f, doraise=True)


def test_row_order_diagnostics_and_the_one_time_warning():
    """maps.row_order / ascending_order / warn_if_rows_not_ascending: what the fused indexed kernel meets for a submesh map (the
    reference builds ascending maps, solver/maps.py:127-178; a map that goes up and down costs 0.35 instead of 0.8 of the roofline)."""
    import warnings

    from fenics_constitutive_amd import maps

    rng = np.random.default_rng(0)
    assert maps.row_order(np.arange(1000)) == {"ascending": True, "consecutive_tiles": 1.0}
    cells = np.sort(rng.choice(500, 250, replace=False))
    rows = (cells[:, None] * 4 + np.arange(4)).ravel()  # ascending cells of 4 points: ascending, but no tile is one run
    o = maps.row_order(rows)
    assert o["ascending"] and o["consecutive_tiles"] < 0.2
    perm = rng.permutation(1000)
    assert not maps.row_order(perm)["ascending"]
    order = maps.ascending_order(perm)
    assert maps.row_order(perm[order]) == {"ascending": True, "consecutive_tiles": 1.0}
    maps._warned_rows = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        maps.warn_if_rows_not_ascending(rows, "test")  # ascending: silent
        maps.warn_if_rows_not_ascending(perm, "test")
        maps.warn_if_rows_not_ascending(perm, "test")  # once per process
    assert len(w) == 1 and "not ascending" in str(w[0].message) and "ascending_order" in str(w[0].message)
    maps._warned_rows = False
