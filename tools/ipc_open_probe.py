#!/usr/bin/env python3
"""How long does fcamd_ipc_open (hipIpcOpenMemHandle) take for large buffers, two processes on one GPU, opened
simultaneously or one rank after the other?   python tools/ipc_open_probe.py <GiB> <simultaneous|serial>"""
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, gib, mode):
    from fenics_constitutive_amd import _capi

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = _capi.get_context(0)
    numel = int(gib * (1 << 30)) // 8
    if "empty" in mode:  # never touched before the export
        buf = torch.empty(numel, dtype=torch.float64, device="cuda")
    else:
        buf = torch.full((numel,), float(rank), dtype=torch.float64, device="cuda")
    if "churn" in mode:  # allocator history: big blocks allocated and returned to the driver before
        junk = [torch.empty(numel, dtype=torch.float64, device="cuda") for _ in range(4)]
        del junk
        torch.cuda.empty_cache()
        buf = torch.empty(numel, dtype=torch.float64, device="cuda")
    h, off = ctx.ipc_export(buf.data_ptr())
    infos = [None] * world
    dist.all_gather_object(infos, (h, off))
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    ptr = None
    for turn in range(world):
        if "serial" not in mode or turn == rank:
            if ptr is None:
                hh, oo = infos[1 - rank]
                ptr = ctx.ipc_open(hh, oo)
                print(f"rank {rank}: opened {gib} GiB in {time.perf_counter() - t0:.3f} s ({mode})", flush=True)
        if "serial" in mode:
            dist.barrier()
    from fenics_constitutive_amd.placement import tensor_from_pointer

    peer = tensor_from_pointer(ptr, buf.numel(), "cuda:0")
    assert "empty" in mode or "churn" in mode or (float(peer[0]) == float(1 - rank) and float(peer[-1]) == float(1 - rank))
    del peer
    dist.barrier()
    ctx.ipc_close(ptr, infos[1 - rank][1])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    gib, mode = float(sys.argv[1]), sys.argv[2]
    if "RANK" in os.environ:  # under torch.distributed.run
        worker(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["MASTER_PORT"]), gib, mode)
    else:
        mp.spawn(worker, args=(2, 29541, gib, mode), nprocs=2, join=True)
