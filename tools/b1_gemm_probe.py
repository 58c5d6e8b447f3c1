"""The residual GEMMs of a ONE-crop forward (ViT-L: 1374 rows, N = 1024, K = 1024 / 4096, (hi, lo) stream epilogue) per block tile: us per launch.
    gpurun -- python tools/b1_gemm_probe.py      (FOUNDPOSE_AMD_LIB=<variant .so> for A/B runs)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from foundpose_amd._lib import call, ptr, stream

def t(fn, iters=400):
    for _ in range(600): fn()          # ~30 ms of the same launch first: the clock has ramped when the timed burst starts
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for B in (1, 2):
    mv = B * 1374
    M = (mv + 255) // 256 * 256
    for k in (1024, 4096):
        n = 1024
        a = torch.randn(M, k, device="cuda").bfloat16(); w = (torch.randn(n, k, device="cuda") * 0.02).bfloat16()
        bias = torch.zeros(n, device="cuda"); xb = torch.randn(M, n, device="cuda").bfloat16(); xl = torch.zeros(M, n, dtype=torch.bfloat16, device="cuda")
        st = torch.zeros(n // 128, M, 2, device="cuda")
        for tile in ([int(v) for v in os.environ.get("TILES", "64,128,0").split(",")]):
            us = t(lambda: call("fp_gemm_bf16_ln", ptr(a), a.stride(0), ptr(w), w.stride(0), M, n, k, mv, ptr(bias), ptr(xl), n, 8 | (tile << 8), None, None, ptr(xb), n, ptr(st), stream()))
            print(f"B={B} K={k} tile={tile:3d}: {us:7.1f} us")
