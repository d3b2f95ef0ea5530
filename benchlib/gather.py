"""bench.py, N > 1: the stress / tangent all-gather of the single-assembler mode (BASELINE configs[4]), timed separately."""

from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def time_allgather(args, dist, torch, device, rank, world, n, stress_t, tangent, agree=None, budget_left=None):
    """The exchange step of the single-assembler mode (BASELINE config 5), timed separately
    and never part of `value`: every rank's stress slice (6/pt) in one piece and its tangent slice (36/pt)
    in chunks through two chunk buffers that are sized against the free device memory up front
    (fcamd_gather_chunk_plan) -- at 8 x 1e8 points the gathered tangent alone would be 230 GB.
      rccl_*    in-place all_gather_into_tensor (RCCL);
      direct_*  the C ABI's peer copies (fcamd_allgather_direct on IPC-mapped buffers): world-1 concurrent
                copies per rank, one per xGMI link;
      p2p_*     (--gather-direct) one batched isend/irecv group to all peers (RCCL point-to-point)."""
    from fenics_constitutive_amd.sharded import ChunkedGather, PeerBuffers, ShardedEvaluator, shared_empty

    ng = ((min(args.gather_points, n) if args.gather_points > 0 else n) // 64) * 64  # whole tiles: every slot is full
    if ng == 0:
        raise ValueError("fewer than 64 points per rank: nothing to gather")
    ev = ShardedEvaluator(None, ng * world)
    per = ev.plan.per_rank
    assert per == ng == ev.n_local  # whole tiles: every slot is full
    shard_bytes = 42 * 8 * ng
    nccl = args.backend == "nccl"
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info(device)
    # every rank must derive the SAME chunk plan (the chunks are collectives): the smallest free memory of all ranks
    fr = torch.tensor([free], dtype=torch.int64, device=device if nccl else "cpu")
    dist.all_reduce(fr, op=dist.ReduceOp.MIN)
    free = int(fr.item()) // (1 if nccl else world)  # gloo rehearsal: the ranks share one GPU
    reserve = 8 << 30
    out_s_bytes = 6 * per * world * 8
    budget = free - reserve - out_s_bytes
    if budget <= 0:  # the same on every rank: nobody enters a collective
        raise MemoryError(f"{free / 1e9:.1f} GB free: no room for the gathered stress ({out_s_bytes / 1e9:.1f} GB) + {reserve >> 30} GiB reserve")
    out_s = shared_empty(6 * per * world, device)  # mapped by the peers (direct variant): an IPC-safe allocation
    s_mine = out_s[6 * per * rank : 6 * per * rank + 6 * ng]
    s_mine.copy_(stress_t[: 6 * ng])
    t_mine = tangent[: 36 * ng]
    result = {"points_per_rank": ng, "shard_GB": round(shard_bytes / 1e9, 3), "free_GB_before": round(free / 1e9, 1),
              "note": "stress gathered whole (in place), tangent through 2 chunk buffers sized against free memory; outside the timed steps"}

    def timed(fn, reps=2):
        best = None
        for _ in range(reps):
            torch.cuda.synchronize()
            dist.barrier()
            t_ = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            dist.barrier()
            dt_ = time.perf_counter() - t_
            best = dt_ if best is None else min(best, dt_)
        tt_ = torch.tensor([best], dtype=torch.float64, device=device if nccl else "cpu")
        dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
        return float(tt_.item())

    def report(prefix, t):
        result[prefix + "_ms"] = round(t * 1e3, 3)
        result[prefix + "_recv_GBs_per_gpu"] = round(shard_bytes * (world - 1) / t / 1e9, 1)

    for variant in (["rccl"] if nccl else []) + ["direct"] + (["p2p"] if (nccl and args.gather_direct) else []):
        peer = variant == "direct"
        if agree is not None and not agree(budget_left() > 75):  # every variant is a set of collectives: all ranks or none
            result[variant + "_skipped"] = "wall budget"
            continue
        if rank == 0:
            print(f"# allgather leg: {variant}, {ng} points per rank, budget {budget / 1e9:.1f} GB", file=sys.stderr, flush=True)
        try:  # set-up failures are raised on all ranks together (PeerBuffers exchanges the outcome of every step)
            cg = ChunkedGather(ev, 36, budget, like=tangent, peer_copies=peer)  # raises up front if the budget holds no tile
            peers_s = PeerBuffers(out_s) if peer else None
        except Exception as e:
            result[variant + "_error"] = f"{type(e).__name__}: {e}"[:300]
            torch.cuda.empty_cache()
            continue
        result["tangent_chunks"], result["chunk_points"] = cg.plan.n_chunks, cg.plan.chunk
        result["chunk_buffers_GB"] = round(2 * cg.plan.buffer_numel * 8 / 1e9, 2)

        def run():
            if peer:
                ev.allgather_peer(s_mine, out_s, 6, peers_s)
            elif variant == "p2p":
                ev.allgather_direct(s_mine, out_s, 6)
            else:
                ev.allgather(s_mine, out_s, 6)
            for _k, _view in cg.chunks(t_mine):
                pass  # the consumer (the assembler) would read _view here

        try:
            report(variant, timed(run))
        finally:
            if peers_s is not None:
                peers_s.close()
            cg.close()
            del cg
            torch.cuda.empty_cache()
    del out_s
    return result
