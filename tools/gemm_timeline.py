"""Per-block phase timeline of the bf16 GEMM (debug stamps): prologue / main loop / epilogue in shader-clock ticks.

Needs the measurement build of the library (the shipped one has no stamp code and no such entry point):
    python tools/gemm_timeline.py --build      # in the build container: foundpose_amd/lib/timeline.so (-DFP_GEMM_TIMELINE)
    python tools/gemm_timeline.py proj qkv     # on the GPU box
"""
import ctypes as C
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TL_SO = os.path.join(ROOT, "foundpose_amd", "lib", "timeline.so")
if "--build" in sys.argv:
    from foundpose_amd import build as fb
    fb.build(verbose=False)
    objs = []
    for src in fb.SOURCES:
        o = os.path.join(fb.OBJDIR, os.path.splitext(src)[0] + ".o")
        if src in ("api.cpp", "gemm_bf16.hip"):
            o = f"/tmp/timeline_{os.path.splitext(src)[0]}.o"
            subprocess.check_call(["/opt/rocm/bin/hipcc", *fb.FLAGS, "-DFP_GEMM_TIMELINE", "-x", "hip", "-c", os.path.join(fb.CSRC, src), "-o", o])
        objs.append(o)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", TL_SO, *objs])
    print(TL_SO)
    sys.exit(0)
import torch
import numpy as np
_tl = C.CDLL(TL_SO)
_tl.fp_gemm_bf16_timeline.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]


def gemm(a, w, bias, gamma, out, epilogue, m_valid, dbg=None):
    scratch = dbg if dbg is not None else torch.zeros(4 * (a.shape[0] // 128) * (w.shape[0] // 128) * 2, dtype=torch.int64, device=a.device)
    rc = _tl.fp_gemm_bf16_timeline(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), a.shape[0], w.shape[0], a.shape[1], m_valid, bias.data_ptr(),
                                   gamma.data_ptr(), out.data_ptr(), out.stride(0), epilogue, scratch.data_ptr(), scratch.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, _tl.fp_last_error()
B, N, D = 32, 1374, 1024
M = (B * N + 255) // 256 * 256
dev = "cuda"
for what in [x for x in sys.argv[1:] if not x.startswith("-")] or ["proj", "qkv", "fc2"]:
    n, k, epi = {"qkv": (3 * D, D, 0), "proj": (D, D, 3), "fc1": (4 * D, D, 1), "fc2": (D, 4 * D, 3)}[what]
    a = torch.randn(M, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16)
    bias, gamma = torch.randn(n, device=dev), torch.randn(n, device=dev)
    out = torch.zeros(M, n, dtype=torch.float32 if epi == 3 else torch.bfloat16, device=dev)
    grid = (M // 256) * (n // 256)
    dbg = torch.zeros(4 * (M // 128) * (n // 128) * 2, dtype=torch.int64, device=dev)
    for _ in range(2):
        gemm(a, w, bias, gamma, out, epi | (256 << 8), B * N)
    torch.cuda.synchronize()
    gemm(a, w, bias, gamma, out, epi | (256 << 8), B * N, dbg)
    torch.cuda.synchronize()
    t = dbg[:4 * grid].reshape(grid, 4).cpu().numpy().astype(np.float64)
    t0 = t[:, 0].min()
    pro, main, epi_t = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    print(f"{what}: grid {grid}  kernel span {t[:,3].max()-t0:.0f} ticks")
    print(f"  prologue  mean {pro.mean():8.0f}  p10 {np.percentile(pro,10):8.0f} p90 {np.percentile(pro,90):8.0f}")
    print(f"  main loop mean {main.mean():8.0f}  p10 {np.percentile(main,10):8.0f} p90 {np.percentile(main,90):8.0f}  ({main.mean()/(k/64):.0f} per K-tile)")
    print(f"  epilogue  mean {epi_t.mean():8.0f}  p10 {np.percentile(epi_t,10):8.0f} p90 {np.percentile(epi_t,90):8.0f}")
    order = np.argsort(t[:, 0])
    starts = (t[order, 0] - t0)
    print("  block start times (ticks) every 64th:", [int(x) for x in starts[::64]][:16])
