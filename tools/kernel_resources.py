#!/usr/bin/env python3
"""tools/kernel_resources.py FILE.hip -- registers, scratch and LDS of every kernel in a HIP source (hipcc -Rpass-analysis=kernel-resource-usage, gfx950)."""
import re
import subprocess
import sys

source = sys.argv[1]
output = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", source, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True).stdout
kernel = None
rows = {}
for line in output.splitlines():
    match = re.search(r"remark: .*?(Function Name|Name): (\S+)", line)
    if match:
        kernel = match.group(2)
        rows[kernel] = {}
        continue
    match = re.search(r"remark: .*?\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs): (\d+)", line)
    if match and kernel:
        rows[kernel][match.group(1).split(" ")[0]] = int(match.group(2))
print("%-60s %6s %6s %8s %6s %8s" % ("kernel", "VGPRs", "SGPRs", "scratch", "occ", "LDS"))
for kernel, r in rows.items():
    if "rocprim" in kernel:
        continue
    name = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", kernel)
    name = re.sub(r"E(N4agpu|PK|Pj|Pm|j|m).*$", "", name)
    print("%-60s %6s %6s %8s %6s %8s" % (name[:60], r.get("VGPRs"), r.get("SGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS")))
