"""Placement tuning of the large written arrays (MI355X finding, DESIGN.md 6 "Run-to-run variance").

On MI355X the time of the evaluate kernels depends on WHICH device memory the driver hands out for
the arrays the kernel writes -- above all the tangent, 51-63 % of the traffic: with everything else
fixed, candidate allocations of the tangent alone move the kernel between 4.49 and 5.49 ms at 5e7
points (tools/tangent_placement_probe.py), reproducibly for a given allocation and invisible to
plain fill / read bandwidth tests of the same memory.  Virtual offsets inside an allocation do not
matter (tools/offset_probe.py), so there is nothing to align: the remedy is to allocate a few
candidates while the earlier ones are still alive (the driver then has to hand out different
memory), time the real kernel on each and keep the fastest.  For a simulation that evaluates the
same arrays for thousands of Newton iterations this costs a few launches once.
"""

from __future__ import annotations

__all__ = ["fastest_allocation"]


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
            if k > 0 and free < 8 * numel + (2 << 30):
                break
            try:
                t = torch.empty(numel, dtype=torch.float64, device=dev)
            except torch.OutOfMemoryError:
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
