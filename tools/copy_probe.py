#!/usr/bin/env python3
"""Device copy bandwidth (torch copy_, 16 GiB) inside one big slab vs separate allocations: is the
run-to-run variance of the evaluate kernel a property of the memory the driver hands out?"""
import torch

dev = torch.device("cuda", 0)
N = 2 * (1 << 30)  # doubles = 16 GiB per array


def bw(src, dst, tag):
    for _ in range(2):
        dst.copy_(src)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
    for a, b in ev:
        a.record()
        dst.copy_(src)
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    print(f"{tag:24s} {2 * 8 * N / (sum(ms) / len(ms) * 1e-3) / 1e12:.3f} TB/s (min-time {2 * 8 * N / (ms[0] * 1e-3) / 1e12:.3f})", flush=True)


slab = torch.empty(80 * (1 << 30), dtype=torch.uint8, device=dev)
a = slab[: 8 * N].view(torch.float64)
b = slab[40 * (1 << 30) : 40 * (1 << 30) + 8 * N].view(torch.float64)
a.fill_(1.0)
bw(a, b, "slab (80 GiB alloc)")
del slab, a, b
torch.cuda.empty_cache()
for i in range(4):
    a = torch.ones(N, dtype=torch.float64, device=dev)
    b = torch.empty(N, dtype=torch.float64, device=dev)
    bw(a, b, f"separate allocs {i}")
    del a, b
    torch.cuda.empty_cache()
