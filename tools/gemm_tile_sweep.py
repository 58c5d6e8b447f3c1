"""qkv / fc1 GEMM (folded-LayerNorm form) at 18 batch sizes with 256- and 320-row block tiles: the data the launcher's tile chooser
(csrc/gemm_bf16.hip::tall_tile_wins, XCD rounds) is calibrated and validated on.   python tools/gemm_tile_sweep.py > profiles/rN_gemm_tile_sweep.txt"""
import torch, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundpose_amd import ops
from tools.bench_kernels import timeit
dev = 'cuda'; K = 1024
for name, n, epi in (("qkv", 3072, 0), ("fc1", 4096, 1)):
    w = (torch.randn(n, K, device=dev) * 0.02).to(torch.bfloat16); bias = torch.randn(n, device=dev)
    cs, = (torch.zeros(n, device=dev),)
    for B in (8, 12, 16, 20, 24, 28, 30, 32, 33, 34, 35, 36, 38, 40, 44, 48, 56, 64):
        mv = B * 1374
        M = (mv + 1279) // 1280 * 1280
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        out = torch.zeros(M, n, dtype=torch.bfloat16, device=dev)
        ln_row = torch.ones(M, 2, device=dev)
        r = {}
        for rep in range(2):
            for tile in (256, 320):
                ms = timeit(lambda: ops.gemm_bf16_ln(a, w, bias, cs, ln_row, epilogue=epi, out=out, tile=tile, m_valid=mv), iters=10)
                r[tile] = min(r.get(tile, 1e9), ms * 1e3)
        t256 = ((mv + 255) // 256) * (n // 256); t320 = ((mv + 319) // 320) * (n // 256)
        print(f"{name} B={B} mv={mv} r256={t256/256:.2f} r320={t320/256:.2f} t256={r[256]:.1f} t320={r[320]:.1f} ratio={r[320]/r[256]:.3f}", flush=True)
