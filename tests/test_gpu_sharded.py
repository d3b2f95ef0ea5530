"""The multi-GPU path on the hardware that is available here: ONE MI355X.

* world 1 (degenerate): the HIP law through ``ShardedEvaluator.evaluate_and_gather`` with every gather
  variant -- slices, slots and the in-place layout are exercised, the exchange is empty.
* world 2 with both ranks on GPU 0: the HIP law on two shards, each rank's gathered buffers mapped into
  the other process through HIP IPC (``fcamd_ipc_export`` / ``fcamd_ipc_open``) and exchanged by the C
  ABI's peer copies (``fcamd_allgather_direct``, push and pull) and by the chunked gather -- the same
  calls an 8-GPU node makes, over device-local copies instead of xGMI.  The process group is gloo (two
  ranks cannot share one GPU under RCCL); it only carries the handles and the barriers.

The reference's requirement mirrored here: a distributed run reproduces the serial one
(tests/solver/test_solver_mpi.py:92-121)."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fenics_constitutive_amd as fc

pytestmark = pytest.mark.gpu

VM_P = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}


def make_inputs(n):
    rng = np.random.default_rng(42)
    scale = np.repeat(10 ** rng.uniform(-4, -2, size=n), 9)
    return (rng.normal(size=9 * n) * scale, rng.normal(scale=30.0, size=6 * n),
            {"eps_n": np.zeros(6 * n), "alpha": rng.uniform(0, 0.02, size=n)})


def unsharded(n):
    g, s, h = make_inputs(n)
    dev = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    sd, td, hd = dev(s), torch.empty(36 * n, dtype=torch.float64, device="cuda"), {k: dev(v) for k, v in h.items()}
    fc.VonMises3D(VM_P).evaluate(0.0, 1.0, dev(g), sd, td, hd, check=True)
    return sd.cpu().numpy(), td.cpu().numpy(), hd["alpha"].cpu().numpy()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def worker(rank, world, port, n, out_dir, backend):
    from fenics_constitutive_amd.sharded import ChunkedGather, PeerBuffers, ShardedEvaluator, shared_empty

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g, s, h = make_inputs(n)
        law = fc.VonMises3D(VM_P)
        ev = ShardedEvaluator(law, n)
        per = ev.plan.per_rank
        f = dict(dtype=torch.float64, device="cuda")
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
        gl = dev(ev.local_view(g, 9))
        results = {}
        variants = [("peer_push", dict(pull=False)), ("peer_pull", dict(pull=True))]
        if backend == "nccl":
            variants += [("rccl", None), ("p2p", None)]
        for name, peer_kw in variants:
            if peer_kw is None:
                sg, tg = torch.zeros(6 * per * world, **f), torch.full((36 * per * world,), float("nan"), **f)
            else:  # buffers the peers map: IPC-safe allocations
                sg, tg = shared_empty(6 * per * world, "cuda:0").zero_(), shared_empty(36 * per * world, "cuda:0").fill_(float("nan"))
            sg[6 * per * rank : 6 * per * rank + 6 * ev.n_local] = dev(ev.local_view(s, 6))
            hl = {"eps_n": dev(ev.local_view(h["eps_n"], 6)), "alpha": dev(ev.local_view(h["alpha"], 1))}
            if peer_kw is None:
                s_all, t_all = ev.evaluate_and_gather(0.0, 1.0, gl, sg, tg, hl, direct=(name == "p2p"))
            else:
                peers = (PeerBuffers(sg), PeerBuffers(tg))
                s_mine = sg[6 * per * rank : 6 * per * rank + 6 * ev.n_local]
                t_mine = tg[36 * per * rank : 36 * per * rank + 36 * ev.n_local]
                ev.evaluate_local(0.0, 1.0, gl, s_mine, t_mine, hl)
                ev.allgather_peer(s_mine, sg, 6, peers[0], **peer_kw)
                ev.allgather_peer(t_mine, tg, 36, peers[1], **peer_kw)
                s_all, t_all = ev.compact(sg, 6), ev.compact(tg, 36)
                for p in peers:
                    p.close()
            law.device_stats(0)
            results[name + "_stress"] = s_all.cpu().numpy()
            results[name + "_tangent"] = t_all.cpu().numpy()
            results["alpha"] = hl["alpha"].cpu().numpy()
            if name == "peer_push":
                # chunked gather of this rank's tangent slice, peer copies on IPC-mapped chunk buffers
                cg = ChunkedGather(ev, 36, budget_bytes=world * 36 * 8 * 64 * 3 * 2, like=t_mine, peer_copies=True)
                out = torch.full((world, per, 36), float("nan"), **f)
                for k, view in cg.chunks(t_mine):
                    lo, hi = cg.plan.span(k)
                    out[:, lo:hi] = view
                cg.close()
                results["chunked_tangent"] = torch.cat([out[r, : ev.plan.count(r)].reshape(-1) for r in range(world)]).cpu().numpy()
                results["n_chunks"] = np.array(cg.plan.n_chunks)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=ev.lo, hi=ev.hi, **results)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [1000, 64 * 31])
def test_world1_hip_law_through_the_sharded_evaluator(n, tmp_path):
    """Degenerate world: every gather variant leaves exactly the unsharded arrays (RCCL with one rank)."""
    mp.spawn(worker, args=(1, free_port(), n, str(tmp_path), "nccl"), nprocs=1, join=True)
    s_ref, t_ref, a_ref = unsharded(n)
    z = np.load(tmp_path / "rank0.npz")
    for name in ("rccl", "p2p", "peer_push", "peer_pull"):
        assert np.array_equal(z[name + "_stress"], s_ref), name
        assert np.array_equal(z[name + "_tangent"], t_ref), name
    assert np.array_equal(z["chunked_tangent"], t_ref)
    assert np.array_equal(z["alpha"], a_ref)


@pytest.mark.parametrize("n", [1000, 64 * 40 + 1, 100_000])
def test_two_ranks_on_one_gpu_exchange_through_ipc_peer_copies(n, tmp_path):
    """Two processes, two shards, one GPU: both ranks end up with the unsharded stress and tangent, through
    push and pull peer copies and through the chunked gather; the history stays sharded."""
    mp.spawn(worker, args=(2, free_port(), n, str(tmp_path), "gloo"), nprocs=2, join=True)
    s_ref, t_ref, a_ref = unsharded(n)
    for r in range(2):
        z = np.load(tmp_path / f"rank{r}.npz")
        for name in ("peer_push", "peer_pull"):
            assert np.array_equal(z[name + "_stress"], s_ref), (r, name)
            assert np.array_equal(z[name + "_tangent"], t_ref), (r, name)
        assert np.array_equal(z["chunked_tangent"], t_ref), r
        assert int(z["n_chunks"]) > 1 or n <= 2 * 64 * 3
        assert np.array_equal(z["alpha"], a_ref[int(z["lo"]) : int(z["hi"])])


def test_ipc_export_refuses_sizes_the_mapping_call_cannot_handle():
    """hipIpcOpenMemHandle of this ROCm stack spins for ever when the exported allocation's size has bit 31
    set (tools/ipc_open_probe.py); the export refuses such allocations and fcamd_ipc_alloc avoids them."""
    from fenics_constitutive_amd import _capi
    from fenics_constitutive_amd.sharded import shared_empty

    ctx = _capi.get_context(0)
    free, _ = torch.cuda.mem_get_info()
    if free < 12 << 30:
        pytest.skip("needs 12 GiB of free HBM")
    torch.cuda.empty_cache()
    bad = torch.empty((3 << 30) // 8, dtype=torch.float64, device="cuda")  # a fresh 3 GiB segment
    with pytest.raises(NotImplementedError, match="mod 4 GiB"):
        ctx.ipc_export(bad.data_ptr())
    del bad
    good = shared_empty((3 << 30) // 8, "cuda:0")  # rounded up to 4 GiB behind the scenes
    handle, offset = ctx.ipc_export(good.data_ptr())
    assert len(handle) == 64 and offset == 0
    small = shared_empty(1000, "cuda:0")
    assert ctx.ipc_export(small.data_ptr())[1] == 0


@pytest.mark.parametrize("pull", [False, True])
def test_allgather_direct_one_process_several_buffers(pull):
    """fcamd_allgather_direct in its "one process drives several GPUs" form, rehearsed with three gathered buffers
    on ONE device (devices[] given explicitly): every "rank" pushes (or pulls) its slot, in two chunks through
    offset / bytes; afterwards all three buffers hold all slots.  Argument checks included."""
    from fenics_constitutive_amd import _capi

    world, slot = 3, 64 * 50 * 36
    ctx = _capi.get_context(0)
    ctx.set_stream(torch.cuda.current_stream(0).cuda_stream)
    bufs = [torch.full((world * slot,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(world)]
    ref = torch.arange(world * slot, dtype=torch.float64, device="cuda")
    for r in range(world):
        bufs[r][r * slot:(r + 1) * slot] = ref[r * slot:(r + 1) * slot]  # rank r produced its own slot
    ptrs = [b.data_ptr() for b in bufs]
    half = 8 * slot // 2
    torch.cuda.synchronize()
    for r in range(world):
        for off, nb in ((0, half), (half, 8 * slot - half)):  # two chunks of the slot
            ctx.allgather_direct(world, r, ptrs, 8 * slot, off, nb, devices=[0] * world, pull=pull)
        ctx.allgather_direct_wait(host_sync=(r % 2 == 0))
    torch.cuda.synchronize()
    for b in bufs:
        assert torch.equal(b, ref)
    with pytest.raises(AssertionError):  # offset + bytes beyond the slot
        ctx.allgather_direct(world, 0, ptrs, 8 * slot, 8, 8 * slot)
    with pytest.raises(ValueError):
        ctx.allgather_direct(world, world, ptrs, 8 * slot)
    ctx.allgather_direct(1, 0, ptrs[:1], 8 * slot)  # world 1: nothing to do
