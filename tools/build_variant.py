#!/usr/bin/env python3
"""Build a VARIANT of libfcamd.so with extra compiler flags, next to the shipped one, for A/B timing on the GPU box:
    python tools/build_variant.py <name> [-DFOO=1 ...]      ->  tools/_ab/libfcamd_<name>.so   (git-ignored; travels with gpurun)
    FCAMD_LIBRARY=tools/_ab/libfcamd_<name>.so python bench.py ...
The working tree's sources are used as they are."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fenics_constitutive_amd import _build  # noqa: E402

name, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "tools", "_ab", f"libfcamd_{name}.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
cmd = ["/opt/rocm/bin/hipcc", f"--offload-arch={_build.ARCH}", *_build.FLAGS, *extra, "-o", out, *[os.path.join(_build.CSRC, s) for s in _build.SOURCES]]
r = subprocess.run(cmd, capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr[-3000:])
print(out)
