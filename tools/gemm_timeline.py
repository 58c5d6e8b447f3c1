"""Per-block phase timeline of the bf16 GEMM (debug stamps): prologue / main loop / epilogue in shader-clock ticks."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundpose_amd import ops
import numpy as np
B, N, D = 32, 1374, 1024
M = (B * N + 255) // 256 * 256
dev = "cuda"
for what in sys.argv[1:] or ["proj", "qkv", "fc2"]:
    n, k, epi = {"qkv": (3 * D, D, 0), "proj": (D, D, 3), "fc1": (4 * D, D, 1), "fc2": (D, 4 * D, 3)}[what]
    a = torch.randn(M, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16)
    bias, gamma = torch.randn(n, device=dev), torch.randn(n, device=dev)
    out = torch.zeros(M, n, dtype=torch.float32 if epi == 3 else torch.bfloat16, device=dev)
    grid = (M // 256) * (n // 256)
    dbg = torch.zeros(grid, 4, dtype=torch.int64, device=dev)
    for _ in range(2):
        ops.gemm_bf16(a, w, bias, gamma=gamma, out=out, epilogue=epi | (256 << 8), m_valid=B * N)
    torch.cuda.synchronize()
    os.environ["FP_GEMM_DBG_PTR"] = str(dbg.data_ptr())
    ops.gemm_bf16(a, w, bias, gamma=gamma, out=out, epilogue=epi | (256 << 8), m_valid=B * N)
    torch.cuda.synchronize()
    del os.environ["FP_GEMM_DBG_PTR"]
    t = dbg.cpu().numpy().astype(np.float64)
    t0 = t[:, 0].min()
    pro, main, epi_t = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    print(f"{what}: grid {grid}  kernel span {t[:,3].max()-t0:.0f} ticks")
    print(f"  prologue  mean {pro.mean():8.0f}  p10 {np.percentile(pro,10):8.0f} p90 {np.percentile(pro,90):8.0f}")
    print(f"  main loop mean {main.mean():8.0f}  p10 {np.percentile(main,10):8.0f} p90 {np.percentile(main,90):8.0f}  ({main.mean()/(k/64):.0f} per K-tile)")
    print(f"  epilogue  mean {epi_t.mean():8.0f}  p10 {np.percentile(epi_t,10):8.0f} p90 {np.percentile(epi_t,90):8.0f}")
    order = np.argsort(t[:, 0])
    starts = (t[order, 0] - t0)
    print("  block start times (ticks) every 64th:", [int(x) for x in starts[::64]][:16])
