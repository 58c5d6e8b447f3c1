#!/usr/bin/env python
"""Same-box A/B of the GEMM / attention kernels on bf16 and fp16 operands at the bench shapes (ViT-L, batch 32): which part of the f16 mode's step-time
difference is the epilogue and which the matrix pipe itself (epilogue 5 = bias -> fp32 has no 16-bit output: its difference is the MFMA's)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from foundpose_amd import ops
from foundpose_amd._lib import call, ptr, stream

def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

B, ntok, D, Hd = 32, 1374, 1024, 4096
mv = B * ntok
M = (mv + 1279) // 1280 * 1280
dev = "cuda"
rows = []
for rep in range(2):
  for dt in (torch.bfloat16, torch.float16):
    bit = (1 << 21) if dt == torch.float16 else 0
    for name, n, k, epi in (("qkv", 3 * D, D, 0), ("qkv_f32out", 3 * D, D, 5), ("fc1_gelu", Hd, D, 1), ("fc1_bias", Hd, D, 0), ("proj_hilo", D, D, 8), ("fc2_hilo", D, Hd, 8)):
        a = torch.randn(M, k, device=dev).to(dt); w = (torch.randn(n, k, device=dev) * 0.02).to(dt)
        bias = torch.zeros(n, device=dev)
        if epi == 8:
            xb = torch.randn(M, n, device=dev).to(dt); xl = torch.zeros(M, n, dtype=dt, device=dev); st = torch.zeros(n // 128, M, 2, device=dev)
            f = lambda: call("fp_gemm_bf16_ln", ptr(a), a.stride(0), ptr(w), w.stride(0), M, n, k, mv, ptr(bias), ptr(xl), n, 8 | bit, None, None, ptr(xb), n, ptr(st), stream())
        elif epi == 5:
            out = torch.zeros(M, n, dtype=torch.float32, device=dev)
            f = lambda: ops.gemm_bf16(a, w, bias, out=out, epilogue=5, m_valid=mv)
        else:
            out = torch.zeros(M, n, dtype=dt, device=dev); cs = torch.zeros(n, device=dev); ln_row = torch.ones(M, 2, device=dev)
            f = lambda: ops.gemm_bf16_ln(a, w, bias, cs, ln_row, epilogue=epi, out=out, m_valid=mv)
        rows.append((rep, str(dt).split(".")[1], name, timeit(f)))
    # all-zero operands: no toggling in the multiplier arrays -- if the two formats then take the same time, the difference above is power (DVFS), not issue rate
    az, wz, oz, bz = torch.zeros(M, D, dtype=dt, device=dev), torch.zeros(3 * D, D, dtype=dt, device=dev), torch.zeros(M, 3 * D, dtype=torch.float32, device=dev), torch.zeros(3 * D, device=dev)
    rows.append((rep, str(dt).split(".")[1], "qkv_f32out_zero_operands", timeit(lambda: ops.gemm_bf16(az, wz, bz, out=oz, epilogue=5, m_valid=mv))))
    xq = torch.randn(M, 3 * D, device=dev).to(dt)
    rows.append((rep, str(dt).split(".")[1], "attention", timeit(lambda: ops.attention(xq, B, ntok, D, 16))))
for r in rows:
    print("%d %-9s %-26s %8.1f us" % r)
