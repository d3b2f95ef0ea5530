"""use_resident_state(problem): the run-time patch of the reference's LawOnSubMesh objects, driven
with stand-ins that restate the reference's host protocol (solver/_solver.py:130-159,
solver/_lawonsubmesh.py:47-110, solver/_history.py:37-90, solver/_incrementalunknowns.py:54-80,
solver/maps.py:82-123) -- dolfinx is not needed, only the attributes the patch touches.  The patched
problem must produce the same global stress / tangent / histories as the unpatched one."""

from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import fenics_constitutive_amd as fc  # noqa: E402
from fenics_constitutive_amd.integration import use_resident_problem_state, use_resident_state  # noqa: E402


class Fn:
    def __init__(self, size):
        self.x = SimpleNamespace(array=np.zeros(size), scatter_forward=lambda: None)


class SubMap:
    """SubSpaceMap of the reference (solver/maps.py:60-123): ``parent`` / ``sub`` index arrays."""

    def __init__(self, rows, sub=None):
        self.parent = rows
        self.sub = np.arange(rows.size) if sub is None else sub

    def map_to_sub(self, parent, sub):
        d = sub.x.array.size // self.parent.size
        sub.x.array.reshape(-1, d)[self.sub] = parent.x.array.reshape(-1, d)[self.parent]

    def map_to_parent(self, sub, parent):
        d = sub.x.array.size // self.parent.size
        parent.x.array.reshape(-1, d)[self.parent] = sub.x.array.reshape(-1, d)[self.sub]


class Stress:
    def __init__(self, n):
        self.current, self.previous = Fn(6 * n), Fn(6 * n)

    def update_previous(self):
        self.previous.x.array[:] = self.current.x.array


class History:
    def __init__(self, dims, n):
        self.history_0 = {k: Fn(d * n) for k, d in dims.items()}
        self.history_1 = {k: Fn(d * n) for k, d in dims.items()}

    def reset_trial_state(self):
        for k in self.history_0:
            self.history_1[k].x.array[:] = self.history_0[k].x.array
        return {k: f.x.array for k, f in self.history_1.items()}

    def update(self):
        for k in self.history_0:
            self.history_0[k].x.array[:] = self.history_1[k].x.array


class LawOnSubMesh:
    def __init__(self, law, cells, rows):
        n = rows.size
        self.law, self.cells = law, cells
        self.displacement_gradient_fn, self.stress, self.local_tangent = Fn(9 * n), Fn(6 * n), Fn(36 * n)
        self.submesh_map = SubMap(rows)
        self.history = None if law.history_dim is None else History(law.history_dim, n)

    def local_stress(self, stress):
        self.submesh_map.map_to_sub(stress.previous, self.stress)
        return self.stress.x.array

    def map_to_parent(self, global_stress, global_tangent):
        self.submesh_map.map_to_parent(self.stress, global_stress.current)
        self.submesh_map.map_to_parent(self.local_tangent, global_tangent)

    def evaluate(self, sim_time, incr_disp, global_stress, global_tangent):
        incr_disp.evaluate_local_incremental_gradient(self.cells, self.displacement_gradient_fn)
        h = self.history.reset_trial_state() if self.history is not None else None
        self.law.evaluate(sim_time.current, sim_time.dt, self.displacement_gradient_fn.x.array,
                          self.local_stress(global_stress), self.local_tangent.x.array, h)
        self.map_to_parent(global_stress, global_tangent)

    def update_history(self):
        if self.history is not None:
            self.history.update()


class IncrDisp:
    def __init__(self):
        self.grads = {}

    def evaluate_local_incremental_gradient(self, cells, fn):
        fn.x.array[:] = self.grads[cells.tobytes()]


class Problem:
    def __init__(self, laws, n):
        self.stress, self.tangent = Stress(n), Fn(36 * n)
        self._law_on_submeshs = [LawOnSubMesh(law, cells, rows) for law, cells, rows in laws]
        self.sim_time = SimpleNamespace(current=0.0, dt=0.5)
        self.incr_disp = IncrDisp()

    def form(self):
        for los in self._law_on_submeshs:
            los.evaluate(self.sim_time, self.incr_disp, self.stress, self.tangent)

    def update(self):
        self.stress.update_previous()
        for los in self._law_on_submeshs:
            los.update_history()
        self.sim_time.current += self.sim_time.dt


def build(n_cells, q, rng_seed):
    rng = np.random.default_rng(rng_seed)
    owner = rng.integers(0, 2, size=n_cells)
    vm = fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0})
    mx = fc.SpringMaxwellModel({"E0": 42.0, "E1": 10.0, "tau": 10.0, "nu": 0.2}, fc.StressStrainConstraint.FULL)
    laws = []
    for k, law in enumerate((vm, mx)):
        cells = np.flatnonzero(owner == k).astype(np.int32)
        rows = (cells[:, None] * q + np.arange(q)[None, :]).reshape(-1)
        laws.append((law, cells, rows))
    p = Problem(laws, n_cells * q)
    p.stress.previous.x.array[:] = rng.normal(scale=5.0, size=6 * n_cells * q)
    p._law_on_submeshs[0].history.history_0["alpha"].x.array[:] = rng.uniform(0, 0.02, size=laws[0][2].size)
    return p


def test_patched_problem_equals_unpatched():
    n_cells, q = 700, 4
    a, b = build(n_cells, q, 3), build(n_cells, q, 3)
    states = use_resident_state(b)
    assert len(states) == 2 and all(hasattr(los, "resident_state") for los in b._law_on_submeshs)
    rng = np.random.default_rng(8)
    for inc in range(3):
        for it in range(3):
            for los_a, los_b in zip(a._law_on_submeshs, b._law_on_submeshs):
                m = los_a.stress.x.array.size // 6
                scale = 10 ** rng.uniform(-4, -1.8 if it else -3.5, size=m)
                g = rng.normal(size=9 * m) * np.repeat(scale, 9)
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
        assert np.array_equal(a.stress.previous.x.array, b.stress.previous.x.array)
    assert b._law_on_submeshs[0].law.last_stats.n_plastic > 0
    for los in b._law_on_submeshs:
        los.law.unpin_arrays()


class IdentityMap:  # name as in the reference (solver/maps.py:29): what use_resident_state keys on
    def map_to_sub(self, parent, sub):
        sub.x.array[:] = parent.x.array[:]

    def map_to_parent(self, sub, parent):
        parent.x.array[:] = sub.x.array[:]


def build_single(n_cells, q, seed):
    rng = np.random.default_rng(seed)
    vm = fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0})
    cells = np.arange(n_cells, dtype=np.int32)
    rows = np.arange(n_cells * q)
    p = Problem([(vm, cells, rows)], n_cells * q)
    p._law_on_submeshs[0].submesh_map = IdentityMap()
    p.stress.previous.x.array[:] = rng.normal(scale=5.0, size=6 * n_cells * q)
    p._law_on_submeshs[0].history.history_0["alpha"].x.array[:] = rng.uniform(0, 0.02, size=n_cells * q)
    return p


@pytest.mark.parametrize("direct", [True, False])
def test_single_material_writes_the_global_arrays_directly(direct):
    """IdentityMap (one law on the whole mesh): with direct_global the patched evaluate writes stress and
    tangent into the problem's global arrays itself (no map_to_parent copy); same numbers either way."""
    n_cells, q = 900, 4
    a, b = build_single(n_cells, q, 5), build_single(n_cells, q, 5)
    (state,) = use_resident_state(b, direct_global=direct)
    los_b = b._law_on_submeshs[0]
    calls = []
    orig = los_b.map_to_parent
    los_b.map_to_parent = lambda *args: (calls.append(1), orig(*args))
    rng = np.random.default_rng(2)
    for inc in range(3):
        for it in range(3):
            m = n_cells * q
            g = rng.normal(size=9 * m) * np.repeat(10 ** rng.uniform(-4, -1.8 if it else -3.5, size=m), 9)
            a.incr_disp.grads[a._law_on_submeshs[0].cells.tobytes()] = g
            b.incr_disp.grads[los_b.cells.tobytes()] = g
            a.form()
            b.form()
            assert np.array_equal(a.stress.current.x.array, b.stress.current.x.array), (inc, it)
            assert np.array_equal(a.tangent.x.array, b.tangent.x.array), (inc, it)
        a.update()
        b.update()
        assert np.array_equal(a._law_on_submeshs[0].history.history_0["eps_n"].x.array,
                              los_b.history.history_0["eps_n"].x.array)
    assert (len(calls) == 0) == direct
    from fenics_constitutive_amd import _capi

    assert los_b.law._handle(_capi.default_device()).ctx.last_host_mode() == 3  # pinned: the kernel moved everything
    los_b.law.unpin_arrays()


@pytest.mark.parametrize("permuted", [False, True])
def test_fused_problem_state_equals_unpatched(permuted):
    """use_resident_problem_state: two laws on interleaved cells, the kernels write their rows of the global
    stress / tangent arrays themselves (no map_to_sub / map_to_parent, no local arrays); with ``permuted`` the
    submesh numbers its points in another order than the parent (general SubSpaceMap.sub)."""
    n_cells, q = 800, 4
    a, b = build(n_cells, q, 13), build(n_cells, q, 13)
    if permuted:
        rng_p = np.random.default_rng(1)
        for pa, pb in zip(a._law_on_submeshs, b._law_on_submeshs):
            perm = rng_p.permutation(pa.submesh_map.parent.size)
            pa.submesh_map.sub, pb.submesh_map.sub = perm, perm.copy()
            # local histories follow the local numbering
            for k, f in pa.history.history_0.items():
                d = f.x.array.size // perm.size
                v = f.x.array.reshape(-1, d).copy()
                f.x.array.reshape(-1, d)[perm] = v
                pb.history.history_0[k].x.array[:] = f.x.array
    for los in b._law_on_submeshs:   # no map call may survive the patch
        los.submesh_map.map_to_sub = los.submesh_map.map_to_parent = None
    state = use_resident_problem_state(b)
    rng = np.random.default_rng(8)
    for inc in range(3):
        for it in range(3):
            for los_a, los_b in zip(a._law_on_submeshs, b._law_on_submeshs):
                m = los_a.stress.x.array.size // 6
                scale = 10 ** rng.uniform(-4, -1.8 if it else -3.5, size=m)
                g = rng.normal(size=9 * m) * np.repeat(scale, 9)
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
    assert state._tangent is None  # the parent tangent never existed on the device
    for los in b._law_on_submeshs:
        los.law.unpin_arrays()


def test_fused_problem_state_on_a_tiny_unpinned_problem():
    """Arrays too small to be page-locked (the reference's own tests run on a handful of cells): the fused
    flow falls back to evaluating on the device arrays and copying each law's rows down."""
    n_cells, q = 40, 4
    a, b = build(n_cells, q, 21), build(n_cells, q, 21)
    use_resident_problem_state(b)
    rng = np.random.default_rng(4)
    for inc in range(2):
        for it in range(2):
            for los_a, los_b in zip(a._law_on_submeshs, b._law_on_submeshs):
                m = los_a.stress.x.array.size // 6
                g = rng.normal(size=9 * m) * np.repeat(10 ** rng.uniform(-4, -2, size=m), 9)
                a.incr_disp.grads[los_a.cells.tobytes()] = g
                b.incr_disp.grads[los_b.cells.tobytes()] = g
            a.form()
            b.form()
            assert np.array_equal(a.stress.current.x.array, b.stress.current.x.array), (inc, it)
            assert np.array_equal(a.tangent.x.array, b.tangent.x.array), (inc, it)
        a.update()
        b.update()
    for los in b._law_on_submeshs:
        los.law.unpin_arrays()
