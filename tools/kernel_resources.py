"""Register / scratch / occupancy table of every kernel of libfcamd, from the compiler's own remarks.

``python tools/kernel_resources.py [--md]`` compiles the device code of the two .hip translation units for gfx950
(device only, nothing is linked or run; works without a GPU) with ``-Rpass-analysis=kernel-resource-usage`` and prints
one row per kernel.  ``tests/test_kernel_resources.py`` fails on any kernel with scratch (a spilled VGPR costs HBM
traffic on a path whose bound is HBM) or fewer than 3 waves per SIMD.
"""

from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FIELDS = {"TotalSGPRs": "sgpr", "VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch",
          "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill",
          "LDS Size [bytes/block]": "lds"}


def demangle(names):
    filt = shutil.which("c++filt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    try:
        out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return [o.replace("fcamd::", "").replace("(fcamd::EvalArgs)", "").replace("(anonymous namespace)::", "") for o in out[: len(names)]]
    except Exception:
        return list(names)


def kernel_resources(sources=None):
    """[{name, mangled, vgpr, sgpr, scratch, occupancy, vgpr_spill, sgpr_spill, lds, source}] for every kernel"""
    from fenics_constitutive_amd import _build

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    flags = [f for f in _build.FLAGS if f not in ("-shared", "-fPIC", "-pthread")]
    rows = []
    for src in sources or [s for s in _build.SOURCES if s.endswith(".hip")]:
        with tempfile.TemporaryDirectory() as tmp:
            cmd = [hipcc, f"--offload-arch={_build.ARCH}", *flags, "--cuda-device-only", "-c", "-Rpass-analysis=kernel-resource-usage",
                   "-o", os.path.join(tmp, "k.o"), os.path.join(_build.CSRC, src)]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-2000:]}")
        cur = None
        for line in r.stderr.splitlines():
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                cur = {"mangled": m.group(1), "source": src}
                rows.append(cur)
                continue
            m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass", line)
            if m and cur is not None and m.group(1).strip() in FIELDS:
                v = m.group(2)
                cur[FIELDS[m.group(1).strip()]] = int(v) if v.lstrip("-").isdigit() else v
    for row, name in zip(rows, demangle([r["mangled"] for r in rows])):
        row["name"] = name
    return rows


def sgpr_spill_traffic(source="fcamd_kernels.hip"):
    """{kernel: (v_writelane_b32, v_readlane_b32)} from the device assembly: an SGPR that does not fit is parked in a VGPR lane, and every
    use of it is a VALU instruction (v_readlane) -- round 4: three more 64-bit uniform words in the packed VonMises3D tile took
    the reloads of that kernel from 57 to 365 and its executed VALU instructions up 22 %, invisible in the register table."""
    import collections

    from fenics_constitutive_amd import _build

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    flags = [f for f in _build.FLAGS if f not in ("-shared", "-fPIC", "-pthread")]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        r = subprocess.run([hipcc, f"--offload-arch={_build.ARCH}", *flags, "--cuda-device-only", "-S", "-o", out, os.path.join(_build.CSRC, source)],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        counts, cur = collections.defaultdict(lambda: [0, 0]), None
        for line in open(out):
            m = re.match(r"^(_Z\w+):", line)
            if m:
                cur = m.group(1)
            elif line.startswith(".Lfunc_end"):
                cur = None
            elif cur and line.startswith("\t"):
                op = line.split()[0] if line.split() else ""
                if op == "v_writelane_b32":
                    counts[cur][0] += 1
                elif op == "v_readlane_b32":
                    counts[cur][1] += 1
    names = list(counts)
    return {n: tuple(counts[m]) for m, n in zip(names, demangle(names))}


def main():
    rows = kernel_resources()
    md = "--md" in sys.argv
    hdr = ["kernel", "VGPR", "SGPR", "scratch B/lane", "VGPR spill", "SGPR spill", "LDS B/block", "waves/SIMD"]
    if md:
        print("| " + " | ".join(hdr) + " |\n|" + "---|" * len(hdr))
    for r in sorted(rows, key=lambda r: r["name"]):
        vals = [r["name"], r.get("vgpr"), r.get("sgpr"), r.get("scratch"), r.get("vgpr_spill"), r.get("sgpr_spill"), r.get("lds"), r.get("occupancy")]
        print(("| " + " | ".join(str(v) for v in vals) + " |") if md else "  ".join(f"{str(v):>6}" if i else f"{str(v):<78}" for i, v in enumerate(vals)))
    bad = [r["name"] for r in rows if r.get("scratch", 0) or r.get("occupancy", 8) < 3]
    print(f"{len(rows)} kernels; with scratch or occupancy < 3: {bad or 'none'}", file=sys.stderr)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
