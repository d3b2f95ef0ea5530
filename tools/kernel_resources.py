#!/usr/bin/env python3
"""Per-kernel resource usage of fcamd_kernels.hip (hipcc -Rpass-analysis=kernel-resource-usage), one line per
kernel: VGPRs, AGPRs, SGPR / VGPR spills, scratch, occupancy, LDS.   python tools/kernel_resources.py [out.tsv]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "fenics-constitutive_amd", "csrc", "fcamd_kernels.hip")
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c", src,
                    "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur, rows = None, {}
for line in r.stderr.splitlines():
    m = re.search(r"remark: .*?:\d+:\d+: +(Function Name|Name): (\S+)", line) or re.search(r"(Function Name|Name): (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()
        cur = cur.replace("fcamd::", "").split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"(VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(2))
cols = ["VGPRs", "AGPRs", "SGPRs Spill", "VGPRs Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"]
out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
out.write("kernel\t" + "\t".join(cols) + "\n")
for k in sorted(rows):
    out.write(k + "\t" + "\t".join(str(rows[k].get(c, "")) for c in cols) + "\n")
