"""Is the attention kernel bound by its structure or by the clock the chip holds under its switching activity?
Times the same launch on random, constant and zero qkv (same instruction stream, different operand toggling)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundpose_amd import ops
from tools.bench_kernels import timeit

B, N, D, H = 32, 1374, 1024, 16
M = (B * N + 255) // 256 * 256
for name, qkv in (("randn*1.5", torch.randn(M, 3 * D, device="cuda") * 1.5), ("randn*0.1", torch.randn(M, 3 * D, device="cuda") * 0.1),
                  ("ones", torch.ones(M, 3 * D, device="cuda")), ("zeros", torch.zeros(M, 3 * D, device="cuda"))):
    q = qkv.to(torch.bfloat16)
    ms = timeit(lambda: ops.attention(q, B, N, D, H), iters=30)
    print(f"attention on {name:10s}: {ms*1e3:8.1f} us  {4.0*B*N*N*D/ms/1e9:7.1f} TF/s", flush=True)
