"""pytest configuration: marker registration and import paths."""

import os
import sys

# The tests move NumPy arrays with torch (``torch.from_numpy(a).cuda()``, ``t.cpu()``): pageable copies, which the HIP
# runtime (above 1 MiB) performs on pieces of the caller's memory that it page-locks on the fly and remembers in a
# cache keyed by address and size.  On this stack memory that appears later at a remembered address (a freed and
# re-allocated array, a heap that shrank and grew) is not GPU-accessible, the cache still calls it locked, and the copy
# dies with "Memory access fault by GPU" -- it did, intermittently, at different tests of this suite (DESIGN.md 6,
# tools/hsa_lock_probe.c).  The product never uses that path (csrc/fcamd_hostpath.cpp: CallerArrays; hostio.py); for
# torch's own copies in the tests the runtime is told to stage everything through its own buffers instead
# (the unit is MiB; read when HIP initialises, i.e. after this line).
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1048576")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")
