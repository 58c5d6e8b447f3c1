"""Same-box A/B of the qkv / fc1 GEMMs with the plain epilogue and with the folded-LayerNorm epilogue, and of the A operand's
statistics (LayerNorm output ~ N(0,1) vs the raw residual stream ~ wider, off-centre): where the fold's cost sits."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundpose_amd import ops
from tools.bench_kernels import timeit
B, N, D = 32, 1374, 1024
M = (B * N + 255) // 256 * 256
dev = "cuda"
for name, n, epi in (("qkv", 3 * D, 0), ("fc1", 4 * D, 1)):
    w = (torch.randn(n, D, device=dev) * 0.02).to(torch.bfloat16)
    bias, cs, ln_row = torch.randn(n, device=dev), torch.randn(n, device=dev), torch.rand(M, 2, device=dev) + 0.5
    out = torch.zeros(M, n, dtype=torch.bfloat16, device=dev)
    for label, a in (("A ~ N(0,1)", torch.randn(M, D, device=dev)), ("A ~ 6 N(0,1) + 1.5", 6 * torch.randn(M, D, device=dev) + 1.5)):
        a = a.to(torch.bfloat16)
        for rep in range(2):
            t0 = timeit(lambda: ops.gemm_bf16(a, w, bias, out=out, epilogue=epi, m_valid=B * N), iters=30)
            t1 = timeit(lambda: ops.gemm_bf16_ln(a, w, bias, cs, ln_row, epilogue=epi, out=out, m_valid=B * N), iters=30)
            print(f"{name} {label:20s} plain {t0*1e3:7.1f} us   folded epilogue {t1*1e3:7.1f} us", flush=True)

for name, k in (("proj", D), ("fc2", 4 * D)):
    a = torch.randn(M, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(D, k, device=dev) * 0.02).to(torch.bfloat16)
    bias, gamma = torch.randn(D, device=dev), torch.rand(D, device=dev)
    x = torch.zeros(M, D, device=dev)
    xb, st = torch.zeros(M, D, dtype=torch.bfloat16, device=dev), torch.zeros(D // 128, M, 2, device=dev)
    from foundpose_amd._lib import call, ptr, stream
    for rep in range(2):
        t0 = timeit(lambda: ops.gemm_bf16(a, w, bias, gamma=gamma, out=x, epilogue=3, m_valid=B * N), iters=30)
        t1 = timeit(lambda: call("fp_gemm_bf16_ln", ptr(a), a.stride(0), ptr(w), w.stride(0), M, D, k, B * N, ptr(bias), ptr(x), D, 7, None, None, None, 0, None, stream()), iters=30)
        t2 = timeit(lambda: call("fp_gemm_bf16_ln", ptr(a), a.stride(0), ptr(w), w.stride(0), M, D, k, B * N, ptr(bias), ptr(x), D, 7, None, None, ptr(xb), D, ptr(st), stream()), iters=30)
        print(f"{name}: LayerScale+residual {t0*1e3:7.1f} us   residual only {t1*1e3:7.1f} us   residual + bf16 copy + row sums {t2*1e3:7.1f} us", flush=True)
