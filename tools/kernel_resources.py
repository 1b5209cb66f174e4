#!/usr/bin/env python3
"""Registers, scratch memory and LDS of the kernels of one .hip file as the compiler reports them (-Rpass-analysis=kernel-resource-usage), one line per kernel:
    python tools/kernel_resources.py arriba_amd/csrc/device/agpu_ingest.hip [name-filter ...]
Scratch bytes per lane are what a kernel's "zero-scratch" claim is checked against without a GPU (hipcc cross-compiles gfx950 here)."""
import re
import subprocess
import sys

source, wanted = sys.argv[1], sys.argv[2:]
result = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", source] ,
                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
kernels, current = [], None
for line in result.stdout.splitlines():
    match = re.search(r"remark: [^ ]+ +(Function Name|Name): (\S+)", line) or re.search(r"(Function Name|Name): (\S+)", line)
    if match and "remark" in line:
        demangled = subprocess.run(["c++filt", match.group(2)], stdout=subprocess.PIPE, universal_newlines=True).stdout.strip()
        current = {"name": re.sub(r"\(anonymous namespace\)::", "", demangled).split("(")[0]}
        kernels.append(current)
        continue
    for key, pattern in (("sgpr", r"SGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"), ("spill_v", r"VGPRs Spill: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        found = re.search(pattern, line)
        if found and current is not None:
            current[key] = int(found.group(1))
if result.returncode != 0:
    sys.stderr.write(result.stdout[-3000:])
    sys.exit(1)
print("%-64s %5s %5s %8s %6s %6s %4s" % ("kernel", "VGPR", "AGPR", "scratch", "spillV", "LDS", "occ"))
for kernel in kernels:
    if kernel["name"].startswith("rocprim") or kernel["name"].startswith("void rocprim"):
        continue
    if wanted and not any(w in kernel["name"] for w in wanted):
        continue
    print("%-64s %5d %5d %8d %6d %6d %4d" % (kernel["name"][:64], kernel.get("vgpr", -1), kernel.get("agpr", 0), kernel.get("scratch", -1), kernel.get("spill_v", 0), kernel.get("lds", 0), kernel.get("occupancy", -1)))
