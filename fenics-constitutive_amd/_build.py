"""Build libfcamd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

``python -m fenics_constitutive_amd._build`` or ``build_library()``.  hipcc cross-compiles
without a GPU; the resulting ``lib/libfcamd.so`` travels to the GPU box with the snapshot.
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libfcamd.so")
SOURCES = ["fcamd_kernels.hip", "fcamd_aux_kernels.hip", "fcamd_capi.cpp", "fcamd_hostpath.cpp", "fcamd_hosttangent.cpp", "fcamd_multigpu.cpp", "fcamd_memory.cpp", "fcamd_multi.cpp"]
# the device code lives in per-law headers that fcamd_kernels.hip includes
KERNEL_HEADERS = [os.path.join("kernels", h) for h in (
    "tile_io.h", "tangent_writers.h", "wrapped_io.h", "history_rows.h", "law_linear_elasticity.h", "law_sls.h",
    "law_von_mises.h", "law_comfe_mises.h", "law_drucker_prager.h", "law_lowdim.h")]
HEADERS = ["fcamd_internal.h", "fcamd_host.h", os.path.join("..", "..", "include", "fcamd.h"), os.path.join("..", "..", "include", "fcamd_multi.h"), *KERNEL_HEADERS]
ARCH = "gfx950"
# -ffp-contract=off: arithmetic order is part of the parity contract (see fcamd_kernels.hip)
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


HASHFILE = os.path.join(LIBDIR, "libfcamd.srchash")


def _source_hash() -> str:
    """Content hash of everything the library is built from (sources, headers, flags).  mtimes are
    not used: a snapshot copy to the GPU box does not preserve their order."""
    import hashlib

    h = hashlib.sha256(" ".join([ARCH, *FLAGS]).encode())
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


KERNEL_SOURCES = ["fcamd_kernels.hip", "fcamd_internal.h", *KERNEL_HEADERS]
# flags that change the device code (host-only flags such as -pthread must not invalidate a PMC measurement)
KERNEL_FLAGS = [f for f in FLAGS if f not in ("-pthread", "-fPIC", "-shared", "-Wall", "-Wno-unused-function")]
KERNEL_HASHFILE = os.path.join(LIBDIR, "libfcamd.kernelhash")


def _code_only(text: str) -> str:
    """C / C++ source without comments and with whitespace collapsed: what the compiler's output depends on"""
    import re

    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return " ".join(text.split())


def kernel_hash(read=None) -> str:
    """Content hash of the DEVICE code alone (kernels, their headers, arch and codegen flags; comments and whitespace
    do not count): what measured HBM traffic depends on.  profiles/traffic.json is keyed by it, so host-side changes
    of the C ABI layer and comment edits do not invalidate a PMC measurement, a kernel change does.  `read(relative
    path) -> text` substitutes another source tree (tools: the hash of an earlier commit)."""
    import hashlib

    if read is None:
        def read(f):
            with open(os.path.join(CSRC, f)) as fh:
                return fh.read()

    h = hashlib.sha256(" ".join([ARCH, *KERNEL_FLAGS]).encode())
    for f in KERNEL_SOURCES:
        h.update(f.encode() + b"\0" + _code_only(read(f)).encode() + b"\0")
    return h.hexdigest()


def built_kernel_hash():
    """kernel_hash() of the sources the library in lib/ was built from (None if unknown)."""
    try:
        with open(KERNEL_HASHFILE) as fh:
            return fh.read().strip()
    except OSError:
        return None


def _stale() -> bool:
    if not (os.path.exists(LIB) and os.path.exists(HASHFILE)):
        return True
    with open(HASHFILE) as fh:
        return fh.read().strip() != _source_hash()


def build_library(force: bool = False, verbose: bool = False, keep_temps: bool = False) -> str:
    """Compile the library if missing or older than its sources; return its path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found; cannot build libfcamd.so")
    os.makedirs(LIBDIR, exist_ok=True)
    tmp = f"{LIB}.{os.getpid()}.tmp"  # several ranks may build at once; the rename below is atomic
    cmd = [hipcc, f"--offload-arch={ARCH}", *FLAGS]
    if keep_temps:
        bdir = os.path.join(CSRC, "build")
        os.makedirs(bdir, exist_ok=True)
        cmd += [f"-save-temps={'obj'}", "-Rpass-analysis=kernel-resource-usage"]
        tmp = os.path.join(bdir, "libfcamd.so")
    cmd += ["-o", tmp, *[os.path.join(CSRC, s) for s in SOURCES]]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed ({r.returncode}):\n{r.stdout}\n{r.stderr}")
    if keep_temps:
        shutil.copyfile(tmp, LIB + ".cp")
        os.replace(LIB + ".cp", LIB)
        sys.stderr.write(r.stderr)
    else:
        os.replace(tmp, LIB)
    with open(HASHFILE + f".{os.getpid()}", "w") as fh:
        fh.write(_source_hash())
    os.replace(HASHFILE + f".{os.getpid()}", HASHFILE)
    with open(KERNEL_HASHFILE + f".{os.getpid()}", "w") as fh:
        fh.write(kernel_hash())
    os.replace(KERNEL_HASHFILE + f".{os.getpid()}", KERNEL_HASHFILE)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True, keep_temps="--temps" in sys.argv))
