"""Multi-GPU form of the hot path: one process per GPU (``torch.distributed``, backend
"nccl" = RCCL over xGMI), the quadrature-point axis cut into contiguous slices.

Every point is independent, so the evaluation itself needs no collective:
rank r owns points ``[lo_r, hi_r)`` of every array, history stays sharded for ever.  The
only exchange step is optional: an all-gather of stress (6 f64/pt) and tangent (36 f64/pt)
for a single assembling process (north star).  Under dolfinx/MPI each rank assembles its own
cells and no gather is needed.

Slices are aligned to 64 points (one wavefront tile) and padded to a common length so the
gather is a single in-place ``all_gather_into_tensor`` per array (one large message per
peer: xGMI is point-to-point, so fewer, larger transfers are what it wants).
"""

from __future__ import annotations

from dataclasses import dataclass

TILE = 64


@dataclass(frozen=True)
class ShardPlan:
    n: int  # global number of points
    world: int
    per_rank: int  # padded points per rank (multiple of TILE)

    @staticmethod
    def create(n: int, world: int) -> "ShardPlan":
        per = -(-n // world)  # ceil
        per = -(-per // TILE) * TILE
        return ShardPlan(int(n), int(world), int(per))

    def bounds(self, rank: int) -> tuple[int, int]:
        """[lo, hi) of rank's points in the global arrays (hi - lo may be 0 for trailing ranks)."""
        lo = min(rank * self.per_rank, self.n)
        hi = min(lo + self.per_rank, self.n)
        return lo, hi

    def count(self, rank: int) -> int:
        lo, hi = self.bounds(rank)
        return hi - lo


class ShardedEvaluator:
    """Evaluate a law on this rank's slice and (optionally) all-gather stress and tangent.

    ``law`` is any ``IncrSmallStrainModel`` whose ``evaluate`` accepts the local arrays (GPU
    tensors for the device-backed laws).  ``group`` defaults to the world group.
    """

    def __init__(self, law, n_global: int, group=None):
        import torch.distributed as dist

        self.law = law
        self.group = group
        self.dist = dist
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.plan = ShardPlan.create(n_global, self.world)
        self.lo, self.hi = self.plan.bounds(self.rank)
        self.n_local = self.hi - self.lo

    def local_view(self, global_array, dim: int):
        """Slice of a global flat array (dim values per point) owned by this rank."""
        return global_array[dim * self.lo : dim * self.hi]

    def evaluate_local(self, t, del_t, grad_local, stress_local, tangent_local, history_local):
        """The sharded hot path: no communication."""
        if self.n_local == 0:
            return
        self.law.evaluate(t, del_t, grad_local, stress_local, tangent_local, history_local)

    def allgather(self, local, gathered, dim: int):
        """In-place all-gather of a per-point array: ``gathered`` has ``dim * per_rank * world``
        entries, rank r's slice at ``[dim*per_rank*r, ...)``; ``local`` may already be that
        slice (then nothing is copied locally).  Returns the view of the ``n`` valid points
        when no rank is padded, else ``gathered`` itself (use :meth:`compact`)."""
        per = dim * self.plan.per_rank
        mine = gathered[per * self.rank : per * (self.rank + 1)]
        if local.data_ptr() != mine.data_ptr():
            mine[: local.numel()].copy_(local)
        self.dist.all_gather_into_tensor(gathered, mine, group=self.group)
        return gathered

    def allgather_direct(self, local, gathered, dim: int):
        """Same result as :meth:`allgather`, as one-hop point-to-point transfers: this rank's slice
        is sent to every peer and every peer's slice is received straight into its slot, all in one
        batched group (``batch_isend_irecv`` = one RCCL group call).  On a fully connected xGMI node
        the 7 transfers of a rank use 7 different links at once, where a ring all-gather forwards
        every slice 7 times over one link per step."""
        per = dim * self.plan.per_rank
        mine = gathered[per * self.rank : per * (self.rank + 1)]
        if local.data_ptr() != mine.data_ptr():
            mine[: local.numel()].copy_(local)
        if self.world == 1:
            return gathered
        dist = self.dist
        ops = []
        for shift in range(1, self.world):  # staggered peers: rank r talks to r+shift / r-shift in step `shift`
            dst = (self.rank + shift) % self.world
            src = (self.rank - shift) % self.world
            gdst = dst if self.group is None else dist.get_global_rank(self.group, dst)
            gsrc = src if self.group is None else dist.get_global_rank(self.group, src)
            ops.append(dist.P2POp(dist.isend, mine, gdst, group=self.group))
            ops.append(dist.P2POp(dist.irecv, gathered[per * src : per * (src + 1)], gsrc, group=self.group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        return gathered

    def compact(self, gathered, dim: int):
        """View of the n valid points.  Every rank before the last non-empty one is full, so
        rank r's slot offset ``r * per_rank`` equals its global offset: the valid points are
        contiguous at the front of the gathered buffer."""
        return gathered[: dim * self.plan.n]

    def evaluate_and_gather(self, t, del_t, grad_local, stress_gathered, tangent_gathered, history_local,
                            direct: bool = False):
        """Evaluate directly into this rank's slice of the gathered buffers, then all-gather
        both in place (no staging copy: sendbuf = recvbuf + rank*count).  ``direct`` selects the
        one-hop point-to-point variant."""
        sd, td = 6, 36
        per = self.plan.per_rank
        s_mine = stress_gathered[sd * per * self.rank : sd * per * self.rank + sd * self.n_local]
        t_mine = tangent_gathered[td * per * self.rank : td * per * self.rank + td * self.n_local]
        self.evaluate_local(t, del_t, grad_local, s_mine, t_mine, history_local)
        gather = self.allgather_direct if direct else self.allgather
        gather(s_mine, stress_gathered, sd)
        gather(t_mine, tangent_gathered, td)
        return self.compact(stress_gathered, sd), self.compact(tangent_gathered, td)
