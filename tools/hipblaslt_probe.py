"""Library yardstick: what rocBLAS/hipBLASLt reach on the four ViT-L GEMM shapes of the bench workload.
Not part of the product path; run under `rocprofv3 --kernel-trace --stats` to see the kernel names (tile configuration)."""
import torch, time

M = 32 * 1374
SHAPES = {"qkv": (M, 3072, 1024), "proj": (M, 1024, 1024), "fc1": (M, 4096, 1024), "fc2": (M, 1024, 4096)}
dev = "cuda"
for name, (m, n, k) in SHAPES.items():
    a = (torch.randn(m, k, device=dev) * 0.5).bfloat16()
    w = (torch.randn(n, k, device=dev) * 0.05).bfloat16()
    b = torch.randn(n, device=dev).bfloat16()
    for _ in range(5):
        c = torch.nn.functional.linear(a, w, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        c = torch.nn.functional.linear(a, w, b)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name}: {us:8.1f} us  {2 * m * n * k / us / 1e9:7.3f} PF/s", flush=True)
