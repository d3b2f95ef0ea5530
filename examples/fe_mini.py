"""A dolfinx-free miniature of the reference's incremental small-strain solver, for end-to-end checks of a law inside a
global Newton loop (the role of ``IncrSmallStrainProblem`` + ``NewtonSolver``, src/fenics_constitutive/solver/_solver.py:54-159):
trilinear hexahedra on the unit cube, 2 x 2 x 2 Gauss points, displacement control, SciPy sparse direct solves.

What it takes from a law is exactly the reference's contract -- ``evaluate(t, del_t, grad_del_u, stress, tangent, history)``
on flat point-major arrays, in place -- and the protocol around it is the reference's: per Newton iteration the trial
stress / history are reset to the committed ones (``solver/_lawonsubmesh.py:58-61``, ``solver/_history.py:64-79``), the law
sees the gradient of the displacement INCREMENT of the load step (``solver/_incrementalunknowns.py:23-27``), and a
converged step commits (``_solver.py:149-159``).  A resident state (``evaluate_into`` / ``update``) can stand in for
the copy protocol.  The tangent the law returns is used as it is: the Newton iteration converges quadratically only if
it is the consistent one.

Not product code and not FE infrastructure for users (dolfinx is): test and example scaffolding.
"""

from __future__ import annotations

import numpy as np

SQ2 = np.sqrt(2.0)


class Cube:
    """nx x ny x nz trilinear hexahedra on [0, 1]^3."""

    def __init__(self, nx: int, ny: int, nz: int):
        self.shape = (nx, ny, nz)
        xs, ys, zs = (np.linspace(0.0, 1.0, k + 1) for k in (nx, ny, nz))
        X, Y, Z = np.meshgrid(xs, ys, zs, indexing="ij")
        self.nodes = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
        nid = np.arange(self.nodes.shape[0]).reshape(nx + 1, ny + 1, nz + 1)
        cells = []
        for i in range(nx):
            for j in range(ny):
                for k in range(nz):
                    cells.append([nid[i + a, j + b, k + c] for a, b, c in
                                  ((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1))])
        self.cells = np.array(cells)
        self.n_cells, self.n_nodes = self.cells.shape[0], self.nodes.shape[0]
        self.n_dofs, self.n_points = 3 * self.n_nodes, 8 * self.n_cells
        h = np.array([1.0 / nx, 1.0 / ny, 1.0 / nz])
        sign = np.array([(-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1)], dtype=float)
        gp = sign / np.sqrt(3.0)  # 2 x 2 x 2 Gauss points, weights 1
        # dN_a/dx_c at Gauss point q (the same for every cell: the mesh is a regular box)
        dN = np.empty((8, 8, 3))
        for q in range(8):
            for a in range(8):
                for c in range(3):
                    others = [d for d in range(3) if d != c]
                    dN[q, a, c] = 0.125 * sign[a, c] * np.prod([1.0 + sign[a, d] * gp[q, d] for d in others]) * (2.0 / h[c])
        self.dN = dN
        self.w = np.prod(h) / 8.0  # weight x det J of every Gauss point
        # G[q]: (9 x 24) gradient operator, grad[3 r + c] = d u_r / d x_c;  B[q]: (6 x 24) Mandel strain operator
        self.G = np.zeros((8, 9, 24))
        self.B = np.zeros((8, 6, 24))
        for q in range(8):
            for a in range(8):
                for r in range(3):
                    for c in range(3):
                        self.G[q, 3 * r + c, 3 * a + r] = dN[q, a, c]
                dx, dy, dz = dN[q, a]
                self.B[q, :, 3 * a: 3 * a + 3] = [[dx, 0, 0], [0, dy, 0], [0, 0, dz],
                                                  [dy / SQ2, dx / SQ2, 0], [dz / SQ2, 0, dx / SQ2], [0, dz / SQ2, dy / SQ2]]
        self.cell_dofs = (3 * self.cells[:, :, None] + np.arange(3)[None, None, :]).reshape(self.n_cells, 24)
        self._rows = np.repeat(self.cell_dofs, 24, axis=1).ravel()
        self._cols = np.tile(self.cell_dofs, (1, 24)).ravel()

    def gradient(self, u: np.ndarray) -> np.ndarray:
        """flat grad array (9 per point, point = 8 cell + q) of the displacement field u"""
        ue = u[self.cell_dofs]                              # (cells, 24)
        return np.einsum("qgd,ed->eqg", self.G, ue).reshape(-1)

    def internal_force(self, stress: np.ndarray) -> np.ndarray:
        s = stress.reshape(self.n_cells, 8, 6)
        fe = self.w * np.einsum("qmd,eqm->ed", self.B, s)   # (cells, 24)
        f = np.zeros(self.n_dofs)
        np.add.at(f, self.cell_dofs.ravel(), fe.ravel())
        return f

    def stiffness(self, tangent: np.ndarray):
        import scipy.sparse as sp

        C = tangent.reshape(self.n_cells, 8, 6, 6)
        ke = self.w * np.einsum("qmd,eqmn,qnf->edf", self.B, C, self.B)  # (cells, 24, 24)
        return sp.coo_matrix((ke.ravel(), (self._rows, self._cols)), shape=(self.n_dofs, self.n_dofs)).tocsr()


class CopyProtocolState:
    """The reference's host protocol around an in-place ``evaluate``: committed + trial NumPy arrays."""

    def __init__(self, law, n):
        self.law, self.n = law, n
        self.stress_c, self.stress, self.tangent = np.zeros(6 * n), np.zeros(6 * n), np.zeros(36 * n)
        hd = law.history_dim
        self.hist_c = None if hd is None else {k: np.zeros(d * n) for k, d in hd.items()}
        self.hist = None if hd is None else {k: np.zeros(d * n) for k, d in hd.items()}

    def evaluate(self, t, del_t, grad):
        self.stress[:] = self.stress_c
        if self.hist is not None:
            for k in self.hist:
                self.hist[k][:] = self.hist_c[k]
        self.law.evaluate(t, del_t, grad, self.stress, self.tangent, self.hist)

    def commit(self):
        self.stress_c[:] = self.stress
        if self.hist is not None:
            for k in self.hist:
                self.hist_c[k][:] = self.hist[k]


class ResidentProtocolState:
    """A device-resident state (``ResidentState`` / ``MultiDeviceResidentState``) behind the same two calls."""

    def __init__(self, resident_state, n):
        self.rs, self.n = resident_state, n
        self.stress, self.tangent = np.zeros(6 * n), np.zeros(36 * n)

    def evaluate(self, t, del_t, grad):
        self.rs.evaluate_into(t, del_t, grad, self.stress, self.tangent)

    def commit(self):
        self.rs.update()


def tension_test(mesh: Cube, state, steps: int = 8, top_displacement: float = 0.0065, tilt: float = 0.6, rtol: float = 1e-10,
                 maxit: int = 12, fixed_iterations: int = 0):
    """Pull the top face of the cube (u_z = d (1 + tilt (x - 1/2)), so the strain field is not uniform and a part of the
    cube yields first), bottom face on rollers, rigid-body motion pinned.  Returns per load step the reaction force on the
    top face and the residual norms of the Newton iterations.  ``fixed_iterations`` > 0: exactly that many Newton updates
    per load step, converged or not (to compare two laws' trajectories when the tangent is not the consistent one)."""
    import scipy.sparse.linalg as spla

    X = mesh.nodes
    top, bottom = np.flatnonzero(X[:, 2] > 1 - 1e-12), np.flatnonzero(X[:, 2] < 1e-12)
    fixed = set((3 * bottom + 2).tolist()) | set((3 * top + 2).tolist())
    origin = int(np.flatnonzero((np.abs(X) < 1e-12).all(axis=1))[0])
    xcorner = int(np.flatnonzero((np.abs(X - [1.0, 0.0, 0.0]) < 1e-12).all(axis=1))[0])
    fixed |= {3 * origin, 3 * origin + 1, 3 * xcorner + 1}
    fixed = np.array(sorted(fixed))
    free = np.setdiff1d(np.arange(mesh.n_dofs), fixed)
    shape = 1.0 + tilt * (X[top, 0] - 0.5)
    u, u_prev = np.zeros(mesh.n_dofs), np.zeros(mesh.n_dofs)
    reactions, histories = [], []
    for step in range(1, steps + 1):
        u[3 * top + 2] = top_displacement * step / steps * shape  # the prescribed part of this load step
        norms = []
        for it in range(maxit + 1):
            state.evaluate(float(step - 1), 1.0, mesh.gradient(u - u_prev))
            f = mesh.internal_force(state.stress)
            r = f[free]
            norms.append(float(np.linalg.norm(r)))
            if fixed_iterations and it == fixed_iterations:
                break
            if not fixed_iterations and norms[-1] <= rtol * max(np.linalg.norm(f[fixed]), 1.0):
                break
            K = mesh.stiffness(state.tangent)
            u[free] -= spla.spsolve(K[free][:, free].tocsc(), r)
        else:
            raise RuntimeError(f"Newton iteration of load step {step} did not converge: {norms}")
        state.commit()
        u_prev[:] = u
        reactions.append(float(f[3 * top + 2].sum()))
        histories.append(norms)
    return np.array(reactions), histories, u


def convergence_orders(histories, floor: float = 1e-9):
    """observed order p of every Newton iteration with three consecutive residuals above the rounding floor:
    r3 / r2 = (r2 / r1)^p.  A consistent tangent gives p -> 2."""
    orders = []
    for h in histories:
        for r1, r2, r3 in zip(h[:-2], h[1:-1], h[2:]):
            if r3 > floor * h[0] and r2 < 0.2 * r1:
                orders.append(float(np.log(r3 / r2) / np.log(r2 / r1)))
    return orders


class OracleLaw:
    """a CPU oracle function behind the reference's model interface (tests only)"""

    def __init__(self, fn, params, history_dim):
        self.fn, self.params, self.history_dim = fn, params, history_dim

    def evaluate(self, t, del_t, grad, stress, tangent, history):
        self.fn(self.params, t, del_t, grad, stress, tangent, history)
