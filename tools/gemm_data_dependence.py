"""Same question as tools/attn_data_dependence.py for the fc1 GEMM: random vs constant operands (same instruction stream)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundpose_amd import ops
from tools.bench_kernels import timeit

B, N, D = 32, 1374, 1024
M = (B * N + 255) // 256 * 256
for name, n, k, epi in (("fc1", 4 * D, D, 1), ("fc2", D, 4 * D, 3)):
    for kind in ("randn", "small", "ones", "zeros"):
        mk = {"randn": lambda *s: torch.randn(*s, device="cuda"), "small": lambda *s: torch.randn(*s, device="cuda") * 0.02,
              "ones": lambda *s: torch.ones(*s, device="cuda"), "zeros": lambda *s: torch.zeros(*s, device="cuda")}[kind]
        a, w = mk(M, k).to(torch.bfloat16), mk(n, k).to(torch.bfloat16)
        bias, gamma = torch.zeros(n, device="cuda"), torch.ones(n, device="cuda")
        out = torch.zeros(M, n, dtype=torch.float32 if epi == 3 else torch.bfloat16, device="cuda")
        ms = timeit(lambda: ops.gemm_bf16(a, w, bias, gamma=gamma, out=out, epilogue=epi, m_valid=B * N))
        print(f"{name} on {kind:6s}: {ms*1e3:8.1f} us  {2.0*B*N*n*k/ms/1e9:7.1f} TF/s", flush=True)
