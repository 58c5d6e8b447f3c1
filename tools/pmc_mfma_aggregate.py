#!/usr/bin/env python
"""Matrix-pipe utilisation of the bf16 step's ViT forward from a tools/pmc_mfma.sh table: sum(launches x SQ_VALU_MFMA_BUSY_CYCLES) over the
ViT-forward kernels / (1024 SIMDs x sum(launches x cycles)), and the same ratio per kernel.  The numbers bench.py reports as `mfma_busy_pmc`.
    python tools/pmc_mfma_aggregate.py profiles/r5_bf16_pmc_mfma.txt"""
import re
import sys

VIT = ("gemm_bf16_kernel", "attn_bf16_w64", "ln_finalize", "rowstats_cast", "hilo_rows", "patchify", "prefix_tokens", "ln_sample", "gather_rows", "query_")
busy = cyc = 0
for ln in open(sys.argv[1]):
    m = re.match(r"(.*?)\s+(\d+)\s+(\d+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$", ln)
    if not m or not any(k in m.group(1) for k in VIT):
        continue
    n, c, b = int(m.group(2)), int(m.group(3)), int(m.group(4))
    busy += n * b
    cyc += n * c
    print(f"{m.group(1)[:72]:72s} {n:5d} x {c:9d} cycles  {b / (1024 * c):.3f}")
print(f"vit_forward: {busy / (1024 * cyc):.4f}")
