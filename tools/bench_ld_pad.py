"""Does the row stride of the GEMM operands matter (L2 channel camping)?  Times the four ViT-L GEMMs with the leading
dimensions of A and W padded by `pad` bf16 elements:  python tools/bench_ld_pad.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundpose_amd import ops
from tools.bench_kernels import timeit

B, N, D = 32, 1374, 1024
M = (B * N + 255) // 256 * 256
dev = "cuda"
for name, n, k, epi in (("qkv", 3 * D, D, 0), ("proj", D, D, 3), ("fc1", 4 * D, D, 1), ("fc2", D, 4 * D, 3)):
    for pad_a, pad_w, pad_o in ((0, 0, 0), (64, 64, 0), (64, 64, 32), (64, 64, 64), (64, 64, 128)):
        a = torch.randn(M, k + pad_a, device=dev).to(torch.bfloat16)[:, :k]
        w = (torch.randn(n, k + pad_w, device=dev) * 0.02).to(torch.bfloat16)[:, :k]
        bias, gamma = torch.randn(n, device=dev), torch.randn(n, device=dev)
        out = torch.zeros(M, n + pad_o, dtype=torch.float32 if epi == 3 else torch.bfloat16, device=dev)[:, :n]
        ms = timeit(lambda: ops.gemm_bf16(a, w, bias, gamma=gamma, out=out, epilogue=epi, m_valid=B * N))
        print(f"{name:5s} pad A {pad_a:3d} W {pad_w:3d} out {pad_o:3d}: {ms*1e3:8.1f} us  {2.0*B*N*n*k/ms/1e9:7.1f} TF/s", flush=True)
# attention with a padded qkv row stride
H = 16
for pad in (0, 64, 128):
    qkv = torch.randn(M, 3 * D + pad, device=dev).to(torch.bfloat16)[:, :3 * D]
    ms = timeit(lambda: ops.attention(qkv, B, N, D, H))
    print(f"attn  qkv row stride 3D + {pad:3d}: {ms*1e3:8.1f} us", flush=True)
