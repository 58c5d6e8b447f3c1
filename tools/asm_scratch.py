#!/usr/bin/env python
"""Where a kernel's scratch (spill) accesses sit relative to its MFMA main loop: device assembly of one translation unit, per kernel the line range
of its v_mfma instructions and every scratch_ access (a spill inside the MFMA range costs every K-tile, one outside it once per tile of output).
    python tools/asm_scratch.py foundpose_amd/csrc/gemm_f16.hip [substring of the mangled kernel name]"""
import re
import subprocess
import sys

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wno-unused-value", "-mllvm", "-amdgpu-mfma-vgpr-form"]
subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-x", "hip", "--cuda-device-only", "-S", sys.argv[1], "-o", "/tmp/_k.s"], check=True, capture_output=True)
lines = open("/tmp/_k.s").read().splitlines()
starts = [(i, l[:-1].split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\S+:\s", l + " ")]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
for n, (i, name) in enumerate(starts):
    end = starts[n + 1][0] if n + 1 < len(starts) else len(lines)
    if filt not in name:
        continue
    body = lines[i:end]
    mf = [j for j, l in enumerate(body) if "v_mfma" in l]
    sc = [(j, l.strip().split("\t")[0] if False else " ".join(l.split())[:70]) for j, l in enumerate(body) if "scratch_" in l]
    if not sc:
        continue
    print(f"{name[:110]}\n   {len(body)} lines, v_mfma lines {mf[0] if mf else -1}..{mf[-1] if mf else -1} ({len(mf)})")
    for j, l in sc:
        print(f"   {'IN-LOOP ' if mf and mf[0] <= j <= mf[-1] else '        '}{j}: {l}")
