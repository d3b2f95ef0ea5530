#!/usr/bin/env python3
"""Which path does the HIP runtime take for pageable host<->device copies of a given size (staging buffer or
on-the-fly page-lock of the caller's memory + its cache of page-locked ranges)?  Run with AMD_LOG_LEVEL=4 and grep
"HSA Copy Using":   AMD_LOG_LEVEL=4 python tools/pageable_copy_probe.py 2>&1 | grep -a "HSA Copy Using\\|^size"  """
import sys

import numpy as np
import torch

torch.cuda.init()
torch.zeros(1, device="cuda")
for kib in (64, 512, 1024, 1025, 2048, 16 * 1024, 40 * 1024):
    a = np.ones(kib * 128)
    print(f"size {kib} KiB H2D", file=sys.stderr, flush=True)
    d = torch.from_numpy(a).cuda()
    torch.cuda.synchronize()
    print(f"size {kib} KiB D2H", file=sys.stderr, flush=True)
    b = d.cpu()
    torch.cuda.synchronize()
    print(f"size {kib} KiB H2D non_blocking", file=sys.stderr, flush=True)
    d.copy_(torch.from_numpy(a), non_blocking=True)
    torch.cuda.synchronize()
