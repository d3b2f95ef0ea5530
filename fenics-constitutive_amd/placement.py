"""Placement tuning of the large written arrays (MI355X finding, DESIGN.md 6 "Run-to-run variance").

On MI355X the time of the evaluate kernels depends on WHICH device memory the driver hands out for
the arrays the kernel writes -- above all the tangent, 51-63 % of the traffic: with everything else
fixed, candidate allocations of the tangent alone move the kernel between 4.49 and 5.49 ms at 5e7
points (round-2 probe tangent_placement_probe.py (git history)), reproducibly for a given allocation and invisible to
plain fill / read bandwidth tests of the same memory.  Virtual offsets inside an allocation do not
matter (round-2 probe offset_probe.py (git history)), so there is nothing to align: the remedy is to allocate a few
candidates while the earlier ones are still alive (the driver then has to hand out different
memory), time the real kernel on each and keep the fastest.  For a simulation that evaluates the
same arrays for thousands of Newton iterations this costs a few launches once.
"""

from __future__ import annotations

__all__ = ["fastest_allocation", "max_tries_for_memory", "VmmArraySet", "tensor_from_pointer"]


def fastest_allocation(numel: int, probe, tries: int = 4, device=None, first=None, launches: int = 3):
    """Return ``(tensor, info)``: the float64 device array of ``numel`` elements, out of up to
    ``tries`` candidate allocations, on which ``probe(tensor)`` (one enqueue of the real workload
    writing into ``tensor``) runs fastest.

    ``first``: an existing array to count as candidate 0 (its contents are NOT copied: callers tune
    before the array holds anything they need, or re-evaluate afterwards).  Candidates that do not
    fit into the free device memory are skipped; the rejected ones are returned to the driver.
    ``info`` = ``{"candidate_ms": [...], "chosen": index}``."""
    import torch

    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    cands, times = [], []
    for k in range(max(int(tries), 1)):
        if k == 0 and first is not None:
            t = first
        else:
            free, _ = torch.cuda.mem_get_info(dev)
            if cands and free < 8 * numel + (2 << 30):
                break
            try:
                t = torch.empty(numel, dtype=torch.float64, device=dev)
            except torch.OutOfMemoryError:
                if not cands:
                    raise  # not even one array fits: that is the caller's out-of-memory, not a tuning result
                break
        probe(t)  # warm
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
        for a, b in ev:
            a.record()
            probe(t)
            b.record()
        torch.cuda.synchronize(dev)
        cands.append(t)
        times.append(min(a.elapsed_time(b) for a, b in ev))
    best = times.index(min(times))
    out = cands[best]
    del cands, t
    torch.cuda.empty_cache()
    return out, {"candidate_ms": [round(x, 4) for x in times], "chosen": best}


def max_tries_for_memory(numel: int, tries: int, device=None, reserve_bytes: int = 8 << 30) -> int:
    """Largest number of candidates (<= ``tries``) whose simultaneous allocation leaves ``reserve_bytes`` of the
    device free -- the candidates are alive together while they are timed.  Used where the working set is
    sized against the whole device (``bench.py --gpus N`` next to the gather buffers)."""
    import torch

    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    free, _ = torch.cuda.mem_get_info(dev)
    fit = int(max(free - reserve_bytes, 0) // (8 * numel))
    return max(1, min(int(tries), 1 + fit))  # candidate 0 is the array that exists already



class _RawDeviceArray:
    """Minimal ``__cuda_array_interface__`` carrier: lets torch wrap device memory it did not allocate.
    torch keeps this object alive for as long as any tensor (or view of one) made from it exists, so
    ``owner`` -- whatever must outlive the memory's users -- rides along."""

    def __init__(self, ptr: int, numel: int, typestr: str = "<f8", owner=None):
        self.__cuda_array_interface__ = {"shape": (int(numel),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}
        self.owner = owner


def tensor_from_pointer(ptr: int, numel: int, device, dtype_str: str = "<f8", owner=None):
    """float64 device tensor over ``numel`` doubles at ``ptr`` (no copy).  The tensor and its views keep
    ``owner`` alive; without an owner the caller guarantees that the memory outlives them."""
    import torch

    return torch.as_tensor(_RawDeviceArray(ptr, numel, dtype_str, owner), device=torch.device(device))


class _VmmMemory:
    """The physical side of a VmmArraySet: released when the last tensor view and the set itself are gone."""

    def __init__(self, ctx, ptrs):
        self.ctx, self.ptrs = ctx, list(ptrs)

    def release(self):
        ptrs, self.ptrs = self.ptrs, []
        for p in ptrs:
            self.ctx.free(p)  # synchronises the device first

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class VmmArraySet:
    """A working set placed through the virtual-memory API (``fcamd_device_alloc_set``): one float64 array
    per entry of ``numels`` (name -> element count), physical handles of ``granule`` bytes (0 = 2 MiB)
    created array after array or -- the default -- interleaved over all arrays in proportion to their sizes.

    Why: on MI355X the evaluate kernels run 10-28 % slower on one hipMalloc'ed working set than on another
    (which HBM banks the concurrent streams share); with the arrays of the step interleaved in
    2 MiB handles, twelve fresh working sets in four processes were within 1.2 % (all arrays) / 2.6 % (the
    arrays a resident state owns) of each other, at the level of the better torch draws
    (profiles/r02_placement_vmm.md).

    ``set[name]`` is a torch view; the memory is released when the set and every view are gone
    (``free()`` releases it at once -- only when no view is in use any more)."""

    def __init__(self, ctx, numels: dict, granule: int = 0, interleaved: bool = True, device=None):
        import torch

        self.ctx = ctx
        self.device = torch.device("cuda", ctx.device) if device is None else torch.device(device)
        self.names = list(numels)
        self.numels = {k: int(v) for k, v in numels.items()}
        ptrs = ctx.alloc_set([8 * self.numels[k] for k in self.names], granule, interleaved)
        self.ptrs = dict(zip(self.names, ptrs))
        self._mem = _VmmMemory(ctx, ptrs)

    def __getitem__(self, name):
        return tensor_from_pointer(self.ptrs[name], self.numels[name], self.device, owner=self._mem)

    def free(self) -> None:
        self._mem.release()
        self.ptrs = {}
