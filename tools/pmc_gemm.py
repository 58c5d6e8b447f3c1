"""Runs a few launches of one GEMM / attention config for rocprofv3 --pmc collection."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundpose_amd import ops
what = sys.argv[1] if len(sys.argv) > 1 else "fc2"
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B, N, D, H = 32, 1374, 1024, 16
M = (B * N + 255) // 256 * 256
dev = "cuda"
if what == "attn":
    qkv = torch.randn(M, 3 * D, device=dev).to(torch.bfloat16)
    for _ in range(4):
        ops.attention(qkv, B, N, D, H)
else:
    n, k, epi = {"qkv": (3 * D, D, 0), "proj": (D, D, 3), "fc1": (4 * D, D, 1), "fc2": (D, 4 * D, 3)}[what]
    a = torch.randn(M, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16)
    bias, gamma = torch.randn(n, device=dev), torch.randn(n, device=dev)
    out = torch.zeros(M, n, dtype=torch.float32 if epi == 3 else torch.bfloat16, device=dev)
    for _ in range(4):
        ops.gemm_bf16(a, w, bias, gamma=gamma, out=out, epilogue=epi | (tile << 8), m_valid=B * N)
torch.cuda.synchronize()
