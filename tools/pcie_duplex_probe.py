"""What the PCIe link of an MI355X box gives the host entries: DMA engines and kernel accesses to page-locked host memory, one
direction at a time and both at once.

    python tools/pcie_duplex_probe.py [--mb 1024] [--out gpurun_out/pcie_duplex.json]
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=int, default=1024)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch

    from fenics_constitutive_amd import _capi

    nbytes = args.mb << 20
    dev = torch.device("cuda", _capi.default_device())
    ctx = _capi.get_context(dev.index or 0)
    h_in, h_out = np.ones(nbytes // 8), np.zeros(nbytes // 8)
    ctx.register_host_buffer(h_in)
    ctx.register_host_buffer(h_out)
    p_in, p_out = ctx.device_pointer(h_in), ctx.device_pointer(h_out)
    t_in, t_out = torch.from_numpy(h_in), torch.from_numpy(h_out)  # page-locked by the registration: torch's copies are DMA
    d_a = torch.empty(nbytes // 8, dtype=torch.float64, device=dev)
    d_b = torch.ones(nbytes // 8, dtype=torch.float64, device=dev)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def kernel_copy(dst, src, stream):
        ctx.set_stream(stream.cuda_stream)
        ctx.copy_device(dst, src, nbytes)

    def dma_h2d(stream):
        with torch.cuda.stream(stream):
            d_a.copy_(t_in, non_blocking=True)

    def dma_d2h(stream):
        with torch.cuda.stream(stream):
            t_out.copy_(d_b, non_blocking=True)

    cases = {
        "dma_h2d": (lambda: dma_h2d(s1), 1, 0),
        "dma_d2h": (lambda: dma_d2h(s1), 0, 1),
        "dma_h2d+dma_d2h": (lambda: (dma_h2d(s1), dma_d2h(s2)), 1, 1),
        "kernel_read_host": (lambda: kernel_copy(d_a.data_ptr(), p_in, s1), 1, 0),
        "kernel_write_host": (lambda: kernel_copy(p_out, d_b.data_ptr(), s1), 0, 1),
        "kernel_host_to_host": (lambda: kernel_copy(p_out, p_in, s1), 1, 1),
        "kernel_read_host+kernel_write_host": (lambda: (kernel_copy(d_a.data_ptr(), p_in, s1), kernel_copy(p_out, d_b.data_ptr(), s2)), 1, 1),
        "dma_h2d+kernel_write_host": (lambda: (dma_h2d(s1), kernel_copy(p_out, d_b.data_ptr(), s2)), 1, 1),
        "kernel_read_host+dma_d2h": (lambda: (kernel_copy(d_a.data_ptr(), p_in, s1), dma_d2h(s2)), 1, 1),
    }
    out = {"mb": args.mb, "rows": {}}
    for name, (fn, up, down) in cases.items():
        fn()
        torch.cuda.synchronize()
        best = None
        for _ in range(4):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        row = {"ms": round(best * 1e3, 2), "up_GBs": round(up * nbytes / best / 1e9, 1), "down_GBs": round(down * nbytes / best / 1e9, 1),
               "total_GBs": round((up + down) * nbytes / best / 1e9, 1)}
        out["rows"][name] = row
        print(name, json.dumps(row), flush=True)
    ctx.unregister_host_buffer(h_in)
    ctx.unregister_host_buffer(h_out)
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
