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

Three gather variants, same result (``tests/test_sharded_gloo.py``, ``tests/test_gpu_sharded.py``):

* ``allgather``         RCCL's all-gather (``all_gather_into_tensor``, in place);
* ``allgather_direct``  one batched group of point-to-point sends/receives (RCCL ``isend``/``irecv``);
* ``allgather_peer``    the C ABI's ``fcamd_allgather_direct``: every peer's gathered buffer is mapped
  into this process through HIP IPC (``PeerBuffers``) and the rank's slice is pushed into all of them by
  world-1 concurrent peer copies on world-1 streams -- no collective library on the data path.

``ChunkedGather`` serves shards whose gathered tangent does not fit next to the working set (config 5:
8 x 1e8 points -> 230 GB of tangent per GPU): the consumer receives the gathered array chunk by chunk
through two chunk buffers; the memory budget is checked before anything is allocated.
The rule for the slices (``ShardPlan``) is the C ABI's ``fcamd_shard_bounds``.
"""

from __future__ import annotations

from dataclasses import dataclass

TILE = 64


def _dbg(msg: str) -> None:
    """Progress lines of the exchange steps on stderr (FCAMD_GATHER_DEBUG=1): a hang between ranks is otherwise mute."""
    import os
    import sys
    import time

    if os.environ.get("FCAMD_GATHER_DEBUG") == "1":
        print(f"# [gather rank {os.environ.get('RANK', '?')} t={time.perf_counter():.2f}] {msg}", file=sys.stderr, flush=True)


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

    def allgather_peer(self, local, gathered, dim: int, peers: "PeerBuffers", offset: int = 0, count: int | None = None,
                       pull: bool = False):
        """Same result as :meth:`allgather` through the C ABI (``fcamd_allgather_direct``): ``peers`` holds
        every rank's ``gathered`` buffer mapped into this process (HIP IPC); this rank's slice -- or
        ``count`` values of it from ``offset`` on -- is copied into its slot of all of them by world-1
        concurrent peer copies.  Starts with stream synchronise + barrier (nobody is still reading the previous
        contents of a buffer that is about to be overwritten by its peers) and ends with a barrier: on return every
        rank's buffer is complete.  Consumers of ``gathered`` must be enqueued on this rank's current stream (or be
        finished) before the next call.  ``ChunkedGather`` with ``peer_copies`` keeps the same contract per chunk."""
        per = dim * self.plan.per_rank
        mine = gathered[per * self.rank : per * (self.rank + 1)]
        if local.data_ptr() != mine.data_ptr():
            mine[: local.numel()].copy_(local)
        count = per - offset if count is None else count
        # Entry synchronisation, both modes.  pull: every slot must be complete before anybody reads it.  push: this rank
        # is about to write into every PEER's gathered buffer, and a peer's consumer (assembly kernels on its own stream)
        # may still be reading the data of the previous call -- a fast rank that has run its next evaluate already would
        # overwrite it under the reader (write after read).  So every rank first waits for its own stream (the evaluate
        # that produced its slot AND the consumers of its gathered buffer), then all ranks meet: after the barrier nobody
        # reads the old contents any more.
        import torch

        torch.cuda.current_stream(gathered.device).synchronize()
        peers.ctx.synchronize()
        self.dist.barrier(group=self.group)
        _dbg(f"allgather_peer: {8 * count / 1e9:.2f} GB per slot, {'pull' if pull else 'push'}")
        peers.gather(8 * per, 8 * offset, 8 * count, pull=pull)
        peers.ctx.allgather_direct_wait(host_sync=True)
        _dbg("allgather_peer: own copies done")
        self.dist.barrier(group=self.group)
        return gathered

    def compact(self, gathered, dim: int):
        """View of the n valid points.  Every rank before the last non-empty one is full, so
        rank r's slot offset ``r * per_rank`` equals its global offset: the valid points are
        contiguous at the front of the gathered buffer."""
        return gathered[: dim * self.plan.n]

    def evaluate_and_gather(self, t, del_t, grad_local, stress_gathered, tangent_gathered, history_local,
                            direct: bool = False, peers=None):
        """Evaluate directly into this rank's slice of the gathered buffers, then all-gather
        both in place (no staging copy: sendbuf = recvbuf + rank*count).  ``direct`` selects the
        one-hop point-to-point variant; ``peers = (PeerBuffers of stress_gathered, PeerBuffers of
        tangent_gathered)`` the C ABI's peer copies."""
        sd, td = 6, 36
        per = self.plan.per_rank
        s_mine = stress_gathered[sd * per * self.rank : sd * per * self.rank + sd * self.n_local]
        t_mine = tangent_gathered[td * per * self.rank : td * per * self.rank + td * self.n_local]
        self.evaluate_local(t, del_t, grad_local, s_mine, t_mine, history_local)
        if peers is not None:
            self.allgather_peer(s_mine, stress_gathered, sd, peers[0])
            self.allgather_peer(t_mine, tangent_gathered, td, peers[1])
        else:
            gather = self.allgather_direct if direct else self.allgather
            gather(s_mine, stress_gathered, sd)
            gather(t_mine, tangent_gathered, td)
        return self.compact(stress_gathered, sd), self.compact(tangent_gathered, td)


class _SharedMemory:
    """Owner of a ``fcamd_ipc_alloc`` buffer: freed when the last tensor view is gone."""

    def __init__(self, ctx, ptr):
        self.ctx, self.ptr = ctx, ptr

    def __del__(self):
        try:
            if self.ptr:
                self.ctx.ipc_free(self.ptr)
                self.ptr = 0
        except Exception:
            pass


def shared_empty(numel: int, device, ctx=None):
    """Uninitialised float64 device tensor that peers can map (``PeerBuffers``): allocated through
    ``fcamd_ipc_alloc``, whose sizes hipIpcOpenMemHandle can handle -- a torch allocation may have a size
    ((size mod 4 GiB) >= 2 GiB) for which the mapping call of this ROCm stack never returns, and
    ``fcamd_ipc_export`` refuses those."""
    import torch

    from . import _capi
    from .placement import tensor_from_pointer

    device = torch.device(device)
    ctx = ctx if ctx is not None else _capi.get_context(device.index or 0)
    ptr = ctx.ipc_alloc(8 * int(numel))
    return tensor_from_pointer(ptr, int(numel), device, owner=_SharedMemory(ctx, ptr))


class PeerBuffers:
    """Every rank's copy of one gathered buffer, mapped into this process (one rank per process: HIP IPC,
    ``fcamd_ipc_export`` / ``fcamd_ipc_open``; the 64-byte handles travel through
    ``all_gather_object``).  ``close()`` unmaps the peers' buffers; call it on every rank before any rank
    frees its buffer.  Allocate the buffers with ``shared_empty``: the export refuses allocations whose size
    the mapping call of this ROCm stack cannot handle (NotImplementedError) instead of letting the peers hang."""

    def __init__(self, gathered, group=None, ctx=None):
        import torch
        import torch.distributed as dist

        from . import _capi
        from .device import _current_stream_ptr

        assert gathered.is_cuda and gathered.is_contiguous()
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        dev = gathered.device.index or 0
        self.ctx = ctx if ctx is not None else _capi.get_context(dev)
        self.ctx.set_stream(_current_stream_ptr(dev))
        self._own = gathered  # kept alive while mapped elsewhere
        # Every step that can fail on one rank alone (an allocation the export refuses, a mapping error) is
        # followed by an exchange of the outcome, so that ALL ranks raise together: a rank that left alone
        # would leave the others waiting in the next collective for ever.
        nbytes = gathered.numel() * gathered.element_size()
        _dbg(f"PeerBuffers: exporting {nbytes / 1e9:.2f} GB")
        try:
            handle, offset = self.ctx.ipc_export(gathered.data_ptr())
            mine = ("ok", handle, offset, nbytes)
        except Exception as e:
            mine = ("error", f"{type(e).__name__}: {e}")
        infos = [None] * self.world
        dist.all_gather_object(infos, mine, group=group)
        bad = [f"rank {p}: {i[1]}" for p, i in enumerate(infos) if i[0] != "ok"]
        if bad:
            raise RuntimeError("PeerBuffers: export failed -- " + "; ".join(bad)[:600])
        assert all(i[3] == infos[0][3] for i in infos), "every rank's gathered buffer must have the same size"
        _dbg("PeerBuffers: handles exchanged, opening the peers'")
        self.ptrs, self._opened, err = [], [], None
        for p, (_, h, off, _n) in enumerate(infos):
            if p == self.rank:
                self.ptrs.append(gathered.data_ptr())
                continue
            try:
                ptr = self.ctx.ipc_open(h, off)
                self.ptrs.append(ptr)
                self._opened.append((ptr, off))
            except Exception as e:
                err = f"{type(e).__name__}: {e}"
                break
        status = [None] * self.world
        dist.all_gather_object(status, err, group=group)
        bad = [f"rank {p}: {e}" for p, e in enumerate(status) if e is not None]
        if bad:
            for ptr, off in self._opened:
                self.ctx.ipc_close(ptr, off)
            self._opened = []
            raise RuntimeError("PeerBuffers: mapping a peer's buffer failed -- " + "; ".join(bad)[:600])
        _dbg("PeerBuffers: mapped")
        torch.cuda.synchronize(gathered.device)
        dist.barrier(group=group)

    def gather(self, slot_bytes: int, offset_bytes: int = 0, nbytes: int | None = None, pull: bool = False) -> None:
        """Enqueue the world-1 peer copies of this rank (asynchronous; see ``Context.allgather_direct_wait``)."""
        from .device import _current_stream_ptr

        self.ctx.set_stream(_current_stream_ptr(self._own.device.index or 0))
        self.ctx.allgather_direct(self.world, self.rank, self.ptrs, slot_bytes, offset_bytes, nbytes, pull=pull)

    def close(self) -> None:
        if self._opened:
            _dbg("PeerBuffers: closing")
            self.ctx.allgather_direct_wait(host_sync=True)
            self.dist.barrier(group=self.group)  # nobody is still copying into / out of a mapping
            for ptr, off in self._opened:
                self.ctx.ipc_close(ptr, off)
            self._opened = []
            self.dist.barrier(group=self.group)


@dataclass(frozen=True)
class GatherChunks:
    """Plan of a chunked gather (``fcamd_gather_chunk_plan``): chunk k covers points
    [k*chunk, min((k+1)*chunk, per_rank)) of EVERY rank's slot."""

    per_rank: int
    world: int
    dim: int
    chunk: int
    n_chunks: int
    n_buffers: int

    @staticmethod
    def create(per_rank: int, world: int, dim: int, budget_bytes: int, n_buffers: int = 2) -> "GatherChunks":
        from . import _capi

        chunk, k = _capi.gather_chunk_plan(per_rank, world, dim, budget_bytes, n_buffers)
        return GatherChunks(int(per_rank), int(world), int(dim), chunk, k, int(n_buffers))

    @property
    def buffer_numel(self) -> int:
        """Elements of one chunk buffer: ``world`` slices of ``chunk`` points."""
        return self.world * self.chunk * self.dim

    def span(self, k: int) -> tuple[int, int]:
        """[lo, hi) of chunk k in slot-local points."""
        lo = k * self.chunk
        return lo, min(lo + self.chunk, self.per_rank)

    def slot_offset(self, rank: int) -> int:
        """Element offset of rank's slice inside a chunk buffer."""
        return rank * self.chunk * self.dim


class ChunkedGather:
    """All-gather of a per-point array (``dim`` values per point) that is consumed chunk by chunk:
    ``for k, view in cg.chunks(local): consume(view)`` -- ``view[r]`` is rank r's points of chunk k, shape
    (world, points, dim).  Two chunk buffers alternate, so the gather of chunk k+1 may be enqueued while
    the consumer still reads chunk k.  The buffers are sized against ``budget_bytes`` up front
    (``GatherChunks.create`` raises AssertionError when not even one tile per rank fits)."""

    def __init__(self, evaluator: ShardedEvaluator, dim: int, budget_bytes: int, like, n_buffers: int = 2,
                 peer_copies: bool = False):
        """``peer_copies``: move the chunks with the C ABI's peer copies (``fcamd_allgather_direct`` on
        IPC-mapped chunk buffers) instead of the process group's all-gather."""
        import torch

        self.ev, self.dim = evaluator, dim
        self.plan = GatherChunks.create(evaluator.plan.per_rank, evaluator.world, dim, budget_bytes, n_buffers)
        if peer_copies:  # the chunk buffers are mapped by the peers: sizes the IPC mapping call can handle
            self.buffers = [shared_empty(self.plan.buffer_numel, like.device) for _ in range(n_buffers)]
        else:
            self.buffers = [torch.empty(self.plan.buffer_numel, dtype=like.dtype, device=like.device) for _ in range(n_buffers)]
        self.peers = [PeerBuffers(b, evaluator.group) for b in self.buffers] if peer_copies else None

    def close(self) -> None:
        for p in self.peers or []:
            p.close()
        self.peers = None

    def chunks(self, local):
        """``local``: this rank's slice (dim * n_local values).  Yields (k, view of the gathered chunk)."""
        p, ev = self.plan, self.ev
        for k in range(p.n_chunks):
            lo, hi = p.span(k)
            buf = self.buffers[k % len(self.buffers)]
            mine = buf[p.slot_offset(ev.rank) : p.slot_offset(ev.rank) + p.chunk * p.dim]
            have = max(0, min(hi, ev.n_local) - lo)  # the last rank's slice may end inside (or before) the chunk
            if have:
                mine[: have * p.dim].copy_(local[lo * p.dim : (lo + have) * p.dim])
            if self.peers is None:
                ev.dist.all_gather_into_tensor(buf, mine, group=ev.group)
            else:
                # pushing chunk k into buffer b overwrites chunk k - n_buffers there: every rank has consumed
                # it, because it passed the barrier of chunk k - 1 only after leaving that iteration
                peer = self.peers[k % len(self.buffers)]
                _dbg(f"chunk {k}: pushing {8 * p.chunk * p.dim / 1e9:.2f} GB")
                peer.gather(8 * p.chunk * p.dim)
                peer.ctx.allgather_direct_wait(host_sync=True)
                _dbg(f"chunk {k}: own copies done")
                ev.dist.barrier(group=ev.group)
            yield k, buf.view(p.world, p.chunk, p.dim)[:, : hi - lo]
