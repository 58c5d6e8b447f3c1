"""LayerNorm kernel at the shapes of the modes that run it per block (fp8: ViT-g rows of config 5's share; split modes: ViT-L rows): us per launch and bytes/s.
The ViT-L inputs (180 MB) fit the 256 MiB Infinity Cache when launched back to back: only the ViT-g figure is an HBM rate.   gpurun -- python tools/ln_probe.py"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from foundpose_amd import _lib
from foundpose_amd._lib import call, ptr, stream
def t(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for rows, D, dt, name in ((128 * 1374, 1536, _lib.FP_FP8, "vitg fp8"), (32 * 1374, 1024, _lib.FP_F16X3, "vitl f16x3"), (32 * 1374, 1024, _lib.FP_F16F8, "vitl f16f8"), (32*1374, 1024, _lib.FP_BF16, "vitl bf16")):
    x = torch.randn(rows, D, device="cuda"); w = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")
    if dt == _lib.FP_BF16:
        out = torch.empty(rows, D, dtype=torch.bfloat16, device="cuda")
        us = t(lambda: call("fp_layernorm", ptr(x), D, ptr(w), ptr(b), 1e-6, ptr(out), D, dt, D, rows, rows, rows, 0, stream()))
        ob = 2
    else:
        em = 1 if dt == _lib.FP_FP8 else 4
        out = torch.empty(rows, D * em, dtype=torch.uint8, device="cuda")
        ld = D if dt == _lib.FP_FP8 else 2 * D
        us = t(lambda: call("fp_layernorm_scaled", ptr(x), D, ptr(w), ptr(b), 1e-6, ptr(out), ld, dt, 16.0, D, rows, stream()))
        ob = em
    print(f"{name:12s} rows {rows} D {D}: {us:8.1f} us  {(rows*D*(4+ob))/us/1e6:6.2f} TB/s")
