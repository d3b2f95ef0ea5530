"""Material-point harness: our counterpart of the reference's FE-level model tests without dolfinx.

The reference checks its laws on one-cell / few-cell meshes under homogeneous deformation
(tests/models/test_plasticity.py:13-137, :140-300; tests/models/test_viscoelasticity.py:26-125,
:128-288, :291-366, :369-515).  For a homogeneous state the boundary-value problem reduces to the
material point: some strain components are prescribed, the others are free and found by Newton's
method with the law's consistent tangent until their stresses reach the targets (zero for the
symmetric-boundary uniaxial tests, the traction for the creep tests).  The harness reproduces
the solver's *call protocol* around ``evaluate``:

    per Newton iteration:  trial <- committed (solver/_lawonsubmesh.py:58-61, _history.py:64-79)
                           law.evaluate(t, del_t, grad_del_u, stress, tangent, history)
    per increment:         committed <- trial  (solver/_solver.py:149-159)

for ``n`` independent points at once (each point may carry its own load), with either a host state
(NumPy arrays, in-place ``evaluate`` -- works with the oracle adaptors and with the GPU laws' host
path) or the engine's device-resident state (``ResidentState``: out-of-place evaluate, pointer-swap
commit).
"""

from __future__ import annotations

import numpy as np

SQ2 = 2**0.5

# (geometric dim, stress/strain dim) per constraint name -- models/interfaces.py:14-73
DIMS = {"UNIAXIAL_STRAIN": (1, 1), "UNIAXIAL_STRESS": (1, 1), "PLANE_STRAIN": (2, 4), "PLANE_STRESS": (2, 4), "FULL": (3, 6)}


def grad_from_mandel_strain(eps: np.ndarray, constraint: str) -> np.ndarray:
    """A symmetric displacement gradient whose Mandel strain is ``eps`` (n x sdim); the
    out-of-plane strain of the plane constraints is not part of the gradient."""
    n = eps.shape[0]
    gdim, _ = DIMS[constraint]
    g = np.zeros((n, gdim * gdim))
    if gdim == 1:
        g[:, 0] = eps[:, 0]
    elif gdim == 2:
        g[:, 0], g[:, 3] = eps[:, 0], eps[:, 1]
        g[:, 1] = g[:, 2] = eps[:, 3] / SQ2
    else:
        g[:, 0], g[:, 4], g[:, 8] = eps[:, 0], eps[:, 1], eps[:, 2]
        g[:, 1] = g[:, 3] = eps[:, 3] / SQ2
        g[:, 2] = g[:, 6] = eps[:, 4] / SQ2
        g[:, 5] = g[:, 7] = eps[:, 5] / SQ2
    return g.reshape(-1)


class OracleLaw:
    """An oracle function behind the model interface (evaluate / constraint name / history_dim)."""

    def __init__(self, fn, params, history_dim, constraint="FULL", pass_constraint=False, **kw):
        self.fn, self.params, self.history_dim, self.constraint_name = fn, params, history_dim, constraint
        self.pass_constraint, self.kw = pass_constraint, kw

    def evaluate(self, t, del_t, grad, stress, tangent, history):
        if self.pass_constraint:
            self.fn(self.params, self.constraint_name, t, del_t, grad, stress, tangent, history, **self.kw)
        else:
            self.fn(self.params, t, del_t, grad, stress, tangent, history, **self.kw)


def constraint_name(law) -> str:
    return getattr(law, "constraint_name", None) or law.constraint.name


class HostState:
    """Committed + trial NumPy arrays and the reference's copy protocol around an in-place evaluate."""

    def __init__(self, law, n):
        self.law, self.n = law, n
        gdim, sd = DIMS[constraint_name(law)]
        self.sd = sd
        self.stress_c, self.stress = np.zeros(sd * n), np.zeros(sd * n)
        self.tangent = np.zeros(sd * sd * n)
        hd = law.history_dim
        self.hist_c = None if hd is None else {k: np.zeros(d * n) for k, d in hd.items()}
        self.hist = None if hd is None else {k: np.zeros(d * n) for k, d in hd.items()}

    def evaluate(self, t, del_t, grad):
        self.stress[:] = self.stress_c
        if self.hist is not None:
            for k in self.hist:
                self.hist[k][:] = self.hist_c[k]
        self.law.evaluate(t, del_t, grad, self.stress, self.tangent, self.hist)

    def update(self):
        self.stress_c[:] = self.stress
        if self.hist is not None:
            for k in self.hist:
                self.hist_c[k][:] = self.hist[k]

    def fetch(self):
        return self.stress.reshape(self.n, self.sd), self.tangent.reshape(self.n, self.sd, self.sd)

    def history_of(self, key):
        return self.hist[key]


class ResidentAdapter:
    """The engine's device-resident state behind the same four calls."""

    def __init__(self, law, n, host_assembler=False):
        from fenics_constitutive_amd.resident import ResidentState

        self.rs, self.n = ResidentState(law, n), n
        self.sd = law.stress_strain_dim
        self._s, self._t = np.zeros(self.sd * n), np.zeros(self.sd * self.sd * n)
        self.host_assembler = host_assembler  # True: the pipelined fcamd_evaluate_resident pass

    def evaluate(self, t, del_t, grad):
        if self.host_assembler:
            self.rs.evaluate_into(t, del_t, grad, self._s, self._t)
        else:
            self.rs.evaluate(t, del_t, grad)

    def update(self):
        self.rs.check()
        self.rs.update()

    def fetch(self):
        if not self.host_assembler:
            self.rs.download(self._s, self._t)
        return self._s.reshape(self.n, self.sd), self._t.reshape(self.n, self.sd, self.sd)

    def history_of(self, key):
        # read after update(): the values just committed (the trial copy is the stale one now)
        return self.rs.history_committed[key].cpu().numpy()


class MultiResidentAdapter:
    """The single-process multi-GPU resident state (fcamd_multi_state) behind the same four calls: the state of the n
    bodies sliced over several device contexts, stress / tangent straight into this process's NumPy arrays."""

    def __init__(self, law, n, devices=(0, 0, 0)):
        from fenics_constitutive_amd.multidevice import MultiDeviceResidentState

        self.rs, self.n = MultiDeviceResidentState(law, n, devices=list(devices)), n
        self.sd = law.stress_strain_dim
        self._s, self._t = np.zeros(self.sd * n), np.zeros(self.sd * self.sd * n)

    def evaluate(self, t, del_t, grad):
        self.rs.evaluate_into(t, del_t, np.ascontiguousarray(grad, dtype=np.float64), self._s, self._t)

    def update(self):
        self.rs.update()

    def fetch(self):
        return self._s.reshape(self.n, self.sd), self._t.reshape(self.n, self.sd, self.sd)

    def history_of(self, key):
        return self.rs.history_committed[key]


class MaterialPoints:
    """``n`` homogeneous bodies.  ``increment`` prescribes the Mandel strain increment of the
    controlled components and solves for the free ones so that ``stress[free] == target``."""

    def __init__(self, state, constraint: str, tol=1e-10, maxit=25):
        self.state, self.constraint, self.tol, self.maxit = state, constraint, tol, maxit
        self.n = state.n
        self.sd = DIMS[constraint][1]
        self.time = 0.0
        self.strain = np.zeros((self.n, self.sd))  # total strain of the controlled + free components
        self.iterations = []

    def increment(self, del_t, d_eps: dict, free=(), target=None):
        """d_eps: {component: array(n) or float} prescribed strain increments; ``free`` components
        start from a zero increment (the FE Newton starts from the last converged displacement)."""
        n, sd = self.n, self.sd
        de = np.zeros((n, sd))
        for c, v in d_eps.items():
            de[:, c] = v
        free = list(free)
        tgt = np.zeros((n, len(free))) if target is None else np.broadcast_to(np.asarray(target, dtype=float), (n, len(free)))
        it = 0
        while True:
            self.state.evaluate(self.time, del_t, grad_from_mandel_strain(de, self.constraint))
            s, C = self.state.fetch()
            if not free:
                break
            r = s[:, free] - tgt
            if np.max(np.abs(r)) < self.tol:
                break
            assert it < self.maxit, f"material-point Newton did not converge (|r| = {np.max(np.abs(r)):.3e})"
            J = C[:, free][:, :, free]
            de[:, free] -= np.linalg.solve(J, r[:, :, None])[:, :, 0]
            it += 1
        self.iterations.append(it)
        self.state.update()
        self.time += del_t
        self.strain += de
        return s.copy()
