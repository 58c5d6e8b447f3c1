#!/usr/bin/env python
"""Register / spill table of one translation unit: hipcc -Rpass-analysis=kernel-resource-usage, one line per kernel.
    python tools/kernel_resources.py foundpose_amd/csrc/gemm_f16.hip [extra hipcc flags]"""
import re
import subprocess
import sys

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wno-unused-value", "-mllvm", "-amdgpu-mfma-vgpr-form"]
out = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *sys.argv[2:], "-x", "hip", "-c", sys.argv[1], "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                     capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(2))
for k, v in rows.items():
    print(f"{v.get('VGPRs', 0):4d} vgpr {v.get('AGPRs', 0):4d} agpr  spill {v.get('VGPRs Spill', 0):3d}  scratch {v.get('ScratchSize [bytes/lane]', 0):4d}  occ {v.get('Occupancy [waves/SIMD]', 0)}  {k[:150]}")
