"""Multi-process test of the sharded path on CPU: world_size 2, gloo backend.  The GPU law is
replaced by an oracle-backed stand-in (tests may use the oracle); what is under test is the
shard plan, the in-place gather layout and that sharded == unsharded."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import numpy_oracle as O

VM_P = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}


class OracleLaw:
    """Duck-typed IncrSmallStrainModel on CPU torch tensors (shares memory with NumPy)."""

    def evaluate(self, t, del_t, grad, stress, tangent, history):
        h = None if history is None else {k: v.numpy() for k, v in history.items()}
        O.von_mises_3d(VM_P, t, del_t, grad.numpy(), stress.numpy(), tangent.numpy(), h)


def make_inputs(n):
    rng = np.random.default_rng(42)
    scale = np.repeat(10 ** rng.uniform(-4, -2, size=n), 9)
    return (rng.normal(size=9 * n) * scale, rng.normal(scale=30.0, size=6 * n),
            {"eps_n": np.zeros(6 * n), "alpha": rng.uniform(0, 0.02, size=n)})


def worker(rank, world, port, n, out_dir, direct=False):
    from fenics_constitutive_amd.sharded import ShardedEvaluator

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g, s, h = make_inputs(n)
        ev = ShardedEvaluator(OracleLaw(), n)
        per = ev.plan.per_rank
        # gathered buffers; this rank's slice starts as the committed stress (in-place semantics)
        sg = torch.zeros(6 * per * world, dtype=torch.float64)
        tg = torch.zeros(36 * per * world, dtype=torch.float64)
        sg[6 * per * rank : 6 * per * rank + 6 * ev.n_local] = torch.from_numpy(ev.local_view(s, 6).copy())
        hl = {"eps_n": torch.from_numpy(ev.local_view(h["eps_n"], 6).copy()),
              "alpha": torch.from_numpy(ev.local_view(h["alpha"], 1).copy())}
        gl = torch.from_numpy(ev.local_view(g, 9).copy())
        s_all, t_all = ev.evaluate_and_gather(0.0, 1.0, gl, sg, tg, hl, direct=direct)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), stress=s_all.numpy(), tangent=t_all.numpy(),
                 alpha=hl["alpha"].numpy(), lo=ev.lo, hi=ev.hi)
    finally:
        dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,direct,n", [(2, False, 1000), (2, True, 1000), (3, True, 1000), (2, False, 128), (2, True, 65), (3, True, 128), (3, True, 65),
                                            # the target shape, world 8 (BASELINE configs[4]): full shards, a ragged last one, empty trailing ranks
                                            (8, True, 64 * 8 * 3), (8, False, 1000), (8, True, 129)])
def test_sharded_equals_unsharded(n, world, direct, tmp_path):
    mp.spawn(worker, args=(world, free_port(), n, str(tmp_path), direct), nprocs=world, join=True)
    g, s, h = make_inputs(n)
    t = np.zeros(36 * n)
    O.von_mises_3d(VM_P, 0.0, 1.0, g, s, t, h)
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(z["stress"], s), f"rank {r} gathered stress"
        assert np.array_equal(z["tangent"], t), f"rank {r} gathered tangent"
        assert np.array_equal(z["alpha"], h["alpha"][int(z["lo"]) : int(z["hi"])])  # history stays sharded


# ---- the C ABI's shard rule and the chunked config-5 gather (host logic, no GPU) --------------------------

@pytest.mark.parametrize("n,world", [(0, 2), (1, 2), (64, 2), (65, 2), (1000, 3), (10**8, 8), (8 * 10**8, 8), (129, 8), (7, 1)])
def test_c_shard_bounds_equal_the_python_plan(n, world):
    """fcamd_shard_bounds / fcamd_shard_slot_points (include/fcamd.h) are the rule ShardPlan implements."""
    from fenics_constitutive_amd import _capi
    from fenics_constitutive_amd.sharded import ShardPlan

    plan = ShardPlan.create(n, world)
    assert _capi.shard_slot_points(n, world) == plan.per_rank
    covered = 0
    for r in range(world):
        assert _capi.shard_bounds(n, world, r) == plan.bounds(r)
        covered += plan.count(r)
    assert covered == n
    with pytest.raises(ValueError):
        _capi.shard_bounds(n, world, world)


def test_gather_chunk_plan_config5_fits_next_to_the_working_set():
    """Config 5 (8 x 1e8 points): the gathered tangent would be 230.4 GB per GPU.  Next to the 56.8 GB
    working set and the 38.4 GB gathered stress the chunk buffers get what is left of 288 GB minus a
    reserve; the plan must cover every point exactly once with tile-aligned, equalised chunks."""
    from fenics_constitutive_amd.sharded import GatherChunks

    per, world = 10**8, 8
    budget = int(288e9 - 56.8e9 - 38.4e9 - 16e9)
    p = GatherChunks.create(per, world, 36, budget)
    assert p.chunk % 64 == 0 and p.n_chunks * p.chunk >= per > (p.n_chunks - 1) * p.chunk
    assert p.n_buffers * p.buffer_numel * 8 <= budget
    assert p.n_chunks == 3  # two buffers of <= 88 GB each: three chunks of 33.3 M points
    spans = [p.span(k) for k in range(p.n_chunks)]
    assert spans[0][0] == 0 and spans[-1][1] == per and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans[:-1]) == 0  # equalised
    assert [p.slot_offset(r) for r in range(3)] == [0, 36 * p.chunk, 72 * p.chunk]
    # the budget is asserted up front: one tile per rank in two buffers needs 8*36*8*64*2 bytes
    with pytest.raises(AssertionError, match="holds not even one"):
        GatherChunks.create(per, world, 36, 8 * 36 * 8 * 64 * 2 - 1)
    assert GatherChunks.create(per, world, 36, 8 * 36 * 8 * 64 * 2).chunk == 64
    # everything fits: one chunk
    assert GatherChunks.create(1000, 2, 6, 1 << 30).n_chunks == 1


def chunk_worker(rank, world, port, n, budget, out_dir):
    from fenics_constitutive_amd.sharded import ChunkedGather, ShardedEvaluator

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(36 * n, dtype=torch.float64)
        ev = ShardedEvaluator(OracleLaw(), n)
        local = ev.local_view(full, 36).clone()
        cg = ChunkedGather(ev, 36, budget, like=local)
        out = torch.full((world, ev.plan.per_rank, 36), -1.0, dtype=torch.float64)
        for k, view in cg.chunks(local):
            lo, hi = cg.plan.span(k)
            out[:, lo:hi] = view
        # rank r's slot holds its n_r valid points first: the global array is the concatenation of the valid parts
        parts = [out[r, : ev.plan.count(r)].reshape(-1) for r in range(world)]
        np.save(os.path.join(out_dir, f"chunked{rank}.npy"), torch.cat(parts).numpy())
        np.save(os.path.join(out_dir, f"nchunks{rank}.npy"), np.array([cg.plan.n_chunks, cg.plan.chunk]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,world,tiles_per_chunk", [(1000, 2, 2), (1000, 3, 1), (130, 2, 8), (64 * 7 + 1, 2, 3),
                                                     # config 5 scaled down: 8 ranks, shards of 9 tiles, the gathered tangent in THREE chunks
                                                     (8 * 64 * 9, 8, 3), (8 * 64 * 9 - 100, 8, 3)])
def test_chunked_gather_equals_the_global_array(n, world, tiles_per_chunk, tmp_path):
    """ChunkedGather: every point of every rank arrives exactly once, whatever the chunk length (ragged
    last rank, last chunk shorter than the others, slices that end inside a chunk)."""
    budget = world * 36 * 8 * 64 * tiles_per_chunk * 2
    mp.spawn(chunk_worker, args=(world, free_port(), n, budget, str(tmp_path)), nprocs=world, join=True)
    ref = np.arange(36 * n, dtype=np.float64)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"chunked{r}.npy"), ref), f"rank {r}"
        k, chunk = np.load(tmp_path / f"nchunks{r}.npy")
        assert chunk <= 64 * tiles_per_chunk and (k > 1 or n <= world * chunk)


def test_config5_chunk_plan_scaled_down_is_the_full_size_plan():
    """the world-8 case above runs the SAME plan shape as config 5 at full size: three equalised chunks per shard"""
    from fenics_constitutive_amd.sharded import GatherChunks

    full = GatherChunks.create(10**8, 8, 36, int(288e9 - 56.8e9 - 38.4e9 - 16e9))
    small = GatherChunks.create(64 * 9, 8, 36, 8 * 36 * 8 * 64 * 3 * 2)
    assert full.n_chunks == small.n_chunks == 3 and small.chunk == 64 * 3 and full.n_buffers == small.n_buffers == 2


def agree_worker(rank, world, port, out_dir):
    import importlib.util

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        # a leg with collectives is entered by all ranks or by none: one rank short of budget vetoes it for everybody
        votes = [bench.all_agree(True, dist, "cpu"), bench.all_agree(rank != 5, dist, "cpu"), bench.all_agree(False, dist, "cpu")]
        # max-over-ranks timing and the per-rank kernel times, as the timed region reduces them
        t = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tk = torch.zeros(world, dtype=torch.float64)
        tk[rank] = 8.0 + rank
        dist.all_reduce(tk, op=dist.ReduceOp.SUM)
        np.save(os.path.join(out_dir, f"agree{rank}.npy"), np.array([*votes, float(t.item()), *tk.tolist()]))
    finally:
        dist.destroy_process_group()


def test_bench_collective_decisions_at_world_8(tmp_path):
    """bench.py's N > 1 control decisions at the TARGET world size (the box's process guard allows no 8-rank GPU run: the
    kernels of that shape are rehearsed with 4 ranks in tests/test_gpu_bench_cli.py, the decisions here)."""
    world = 8
    mp.spawn(agree_worker, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        z = np.load(tmp_path / f"agree{r}.npy")
        assert list(z[:3]) == [1.0, 0.0, 0.0], (r, z)
        assert z[3] == pytest.approx(0.008) and list(z[4:]) == [8.0 + k for k in range(world)]
