#!/usr/bin/env python
"""Micro-benchmarks of the hot kernels at the benchmark shapes (ViT-L/14-reg, 518^2, batch 32).
    python tools/bench_kernels.py [gemm] [fp8] [attn] [cos] [ln] [crop] [match]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundpose_amd import ops  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters




def main():
    what = sys.argv[1:] or ["gemm", "attn", "ln"]
    B, N, D, H = int(os.environ.get("BENCH_BATCH", 32)), int(os.environ.get("BENCH_TOKENS", 1374)), 1024, 16   # BENCH_TOKENS=611: the selected rows of the hooked block
    M = (B * N + 1279) // 1280 * 1280   # whole tiles of 256 and of 320 rows
    dev = "cuda"
    if "gemm" in what:
        for name, n, k, epi in (("qkv(bias)", 3 * D, D, 0), ("proj(ls)", D, D, 3), ("fc1(gelu)", 4 * D, D, 1), ("fc2(ls)", D, 4 * D, 3)):
            a = torch.randn(M, k, device=dev).to(torch.bfloat16)
            w = (torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16)
            bias = torch.randn(n, device=dev)
            gamma = torch.randn(n, device=dev)
            out = torch.zeros(M, n, dtype=torch.float32 if epi == 3 else torch.bfloat16, device=dev)
            for tile in ((0, 256, 320, 256, 320) if epi in (0, 1) else (0, 256, 128)):   # 0 = the launcher's choice
                ms = timeit(lambda: ops.gemm_bf16(a, w, bias, gamma=gamma, out=out, epilogue=epi | (tile << 8), m_valid=B * N))
                print(f"gemm {name:10s} M={M} N={n} K={k} tile={tile}: {ms*1e3:8.1f} us  {2.0*B*N*n*k/ms/1e9:7.1f} TF/s", flush=True)
    if "gemmhilo" in what:   # the residual GEMMs on the (hi, lo) stream (epilogue 8), 256- vs 320-row tiles
        for name, n, k in (("proj", D, D), ("fc2", D, 4 * D)):
            a = torch.randn(M, k, device=dev).to(torch.bfloat16)
            w = (torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16)
            bias = torch.randn(n, device=dev)
            xb = torch.randn(M, n, device=dev).to(torch.bfloat16)
            xl = (torch.randn(M, n, device=dev) * 0.003).to(torch.bfloat16)
            for tile in (256, 320, 256, 320):
                ms = timeit(lambda: ops.gemm_bf16_resid_hilo(a, w, bias, xb, xl, tile=tile, m_valid=B * N))
                print(f"gemm hilo {name:5s} M={M} N={n} K={k} tile={tile}: {ms*1e3:8.1f} us  {2.0*B*N*n*k/ms/1e9:7.1f} TF/s", flush=True)
    if "gemmsplit" in what:   # the f16x3 mode's GEMMs: split-fp16 operands, three fp16 MFMAs per product (TF/s = fp32-product equivalent)
        for name, n, k, epi, osc in (("qkv(bias)", 3 * D, D, 0, 16.0), ("proj(ls)", D, D, 3, 0.0), ("fc1(gelu)", 4 * D, D, 1, 4.0), ("fc2(ls)", D, 4 * D, 3, 0.0)):
            a = ops.split16_pack(torch.randn(M, k, device=dev), 16.0, 64)
            wf = torch.randn(n, k, device=dev) * 0.02
            w = ops.split16_pack(wf, ops.pow2_scale(wf), 64)
            bias, gamma = torch.randn(n, device=dev), torch.randn(n, device=dev)
            out = torch.zeros(M, n, dtype=torch.float32, device=dev) if epi == 3 else torch.zeros(M, 2 * n, dtype=torch.float16, device=dev)
            for rep in range(2):
                ms = timeit(lambda: ops.gemm_split(a, w, bias, 1.0 / (16.0 * ops.pow2_scale(wf)), gamma=gamma, out=out, epilogue=epi, out_scale=osc, m_valid=B * N))
                print(f"gemm f16x3 {name:10s} M={M} N={n} K={k}: {ms*1e3:8.1f} us  {2.0*B*N*n*k/ms/1e9:7.1f} TF/s", flush=True)
    if "fp8" in what:
        for name, n, k, epi in (("qkv(bias)", 3 * D, D, 0), ("proj(ls)", D, D, 3), ("fc1(gelu)", 4 * D, D, 1), ("fc2(ls)", D, 4 * D, 3)):
            a = ops.quantize_fp8(torch.randn(M, k, device=dev), 100.0)
            w = ops.quantize_fp8(torch.randn(n, k, device=dev) * 0.02, 5000.0)
            bias, col = torch.randn(n, device=dev), torch.rand(n, device=dev) * 1e-5
            out = torch.zeros(M, n, dtype=torch.float32 if epi == 3 else torch.bfloat16, device=dev)
            for tile in ((256, 320, 256, 320) if epi != 3 else (256,)):
                ms = timeit(lambda: ops.gemm_fp8(a, w, bias, col, out=out, epilogue=epi | (tile << 8), m_valid=B * N))
                print(f"gemm_fp8 {name:10s} M={M} N={n} K={k} tile={tile}: {ms*1e3:8.1f} us  {2.0*B*N*n*k/ms/1e9:7.1f} TF/s", flush=True)
        x = torch.randn(M, 4 * D, device=dev).to(torch.bfloat16)
        ms = timeit(lambda: ops.quantize_fp8(x, 10.0))
        print(f"quantize_fp8 bf16 [{M}, {4*D}]: {ms*1e3:8.1f} us  {M*4*D*3/ms/1e6:7.1f} GB/s", flush=True)
    if "attn" in what:
        qkv = (torch.randn(M, 3 * D, device=dev)).to(torch.bfloat16)
        for rep in range(2):
            for variant in [int(v) for v in os.environ.get("BENCH_ATTN_VARIANTS", "0,3,2").split(",")]:
                ms = timeit(lambda: ops.attention(qkv, B, N, D, H, variant=variant))
                print(f"attn B={B} N={N} H={H} variant={variant}: {ms*1e3:8.1f} us  {4.0*B*N*N*D/ms/1e9:7.1f} TF/s", flush=True)
    if "attnsplit" in what:   # the f16x3 mode's attention (split-fp16 operands, three fp16 MFMAs per product)
        x = torch.randn(M, 3 * D, device=dev)
        packed = torch.cat([ops.split16_pack(x[:, i * D:(i + 1) * D].contiguous(), 16.0) for i in range(3)], dim=1)
        for rep in range(3):
            for variant in (2, 1):   # 2 = the role-split kernel, 1 = the lock-step kernel
                ms = timeit(lambda: ops.attention_split(packed, B, N, D, H, 16.0, 16.0, variant=variant))
                print(f"attn f16x3 B={B} N={N} H={H} variant={variant}: {ms*1e3:8.1f} us  {4.0*B*N*N*D/ms/1e9:7.1f} TF/s (fp32-product equivalent)", flush=True)
    if "attn32" in what:   # the exact-fp32 mode's attention: fp32 MFMA kernel (variant 0) vs the thread-per-query VALU kernel (variant 1)
        qkv = torch.randn(M, 3 * D, device=dev)
        for variant in (0, 1, 0):
            ms = timeit(lambda: ops.attention(qkv, B, N, D, H, variant=variant), iters=5)
            print(f"attn fp32 B={B} N={N} H={H} variant={variant}: {ms*1e3:8.1f} us  {4.0*B*N*N*D/ms/1e9:7.1f} TF/s", flush=True)
    if "cos" in what:
        from foundpose_amd._lib import call, ptr, stream
        for T, W, Bq in ((10000, 2048, 32), (800, 2048, 32), (50000, 2048, 128)):
            # tf-idf-like rows: ~half of the words present (a template's ~375 patches x 3 words of 2048), queries likewise.  (Dense iid
            # uniform rows are the degenerate case of the prefiltered call -- all scores within the candidate window -- BENCH_DENSE=1.)
            if os.environ.get("BENCH_DENSE") == "1":
                bank_n, desc_n = ops.normalize_rows(torch.rand(T, W, device=dev)), ops.normalize_rows(torch.rand(Bq, W, device=dev))
            else:
                bank_n = ops.normalize_rows(torch.rand(T, W, device=dev) * (torch.rand(T, W, device=dev) < 0.5) + 1e-6)
                desc_n = ops.normalize_rows(torch.rand(Bq, W, device=dev) * (torch.rand(Bq, W, device=dev) < 0.5) + 1e-6)
            seg = torch.tensor([0, Bq], dtype=torch.int32, device=dev)
            tpl = torch.tensor([0, T], dtype=torch.int32, device=dev)
            nt = torch.full((Bq,), T, dtype=torch.int32, device=dev)
            from foundpose_amd._lib import cosine_scratch_floats
            sims = torch.empty(cosine_scratch_floats(Bq, T), device=dev)
            sc, ids = torch.empty(Bq, 5, device=dev), torch.empty(Bq, 5, dtype=torch.int32, device=dev)
            for mode in (0, 1):
                ms = timeit(lambda: call("fp_cosine_topk", ptr(desc_n), ptr(seg), ptr(nt), Bq, Bq, ptr(bank_n), ptr(tpl), 1, T, W, 5,
                                         ptr(sims), ptr(sc), ptr(ids), mode, stream()), iters=50)
                passes = (Bq + 31) // 32
                print(f"cosine_topk T={T} W={W} B={Bq} tie_mode={mode}: {ms*1e3:8.1f} us  {passes*(T*W*4)/ms/1e6:7.1f} GB/s (bank bytes x {passes} pass(es))", flush=True)
            from foundpose_amd._lib import cosine_prefilter_scratch_floats
            bank_bf = bank_n.to(torch.float16).contiguous()
            sims2 = torch.empty(cosine_prefilter_scratch_floats(Bq, T), device=dev)
            for mode in (0, 1):
                ms = timeit(lambda: call("fp_cosine_topk_prefiltered", ptr(desc_n), ptr(seg), ptr(nt), Bq, Bq, ptr(bank_n), ptr(bank_bf), ptr(tpl), 1, T, W, 5,
                                         ptr(sims2), ptr(sc), ptr(ids), mode | 256, stream()), iters=50)
                print(f"cosine_topk_prefiltered T={T} W={W} B={Bq} tie_mode={mode}: {ms*1e3:8.1f} us  {passes*(T*W*4)/ms/1e6:7.1f} GB/s of fp32 bank bytes", flush=True)
    if "match" in what:
        # the three exact-fp32 tile launches of one matching step at the bench shapes: PCA 1024 -> 256 of 32 x 517 patches,
        # visual-word 3-NN (2048 words), cyclic distance tiles of 32 x 5 (query crop, template) pairs
        from foundpose_amd._lib import call, ptr, stream
        g = torch.Generator(device=dev).manual_seed(0)
        Bq, Q, d, T, n_top, K = 32, 517, 256, 10000, 5, 300
        raw = torch.randn(Bq * Q, 1024, device=dev, generator=g)
        comps = torch.linalg.qr(torch.randn(1024, d, device=dev, generator=g))[0].T.contiguous()
        mp = torch.randn(d, device=dev, generator=g)
        ms = timeit(lambda: ops.pca_project(raw, comps, mp))
        print(f"pca_project [{Bq*Q}, 1024] -> {d}: {ms*1e3:8.1f} us  {2.0*Bq*Q*1024*d/ms/1e9:7.1f} TF/s (fp32 MFMA peak 157)", flush=True)
        qf = ops.pca_project(raw, comps, mp)
        words = torch.randn(2048, d, device=dev, generator=g)
        qn, wn = ops.sqnorm_rows(qf), ops.sqnorm_rows(words)
        ms = timeit(lambda: ops.knn_l2(qf, words, 3, qn, wn))
        print(f"knn_l2 k=3 [{Bq*Q}] x 2048 words d={d}: {ms*1e3:8.1f} us  {2.0*Bq*Q*2048*d/ms/1e9:7.1f} TF/s", flush=True)
        pc = torch.randint(300, 451, (T,), generator=torch.Generator().manual_seed(1))
        tpl_off = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(pc, 0)]).to(torch.int32).to(dev)
        P = int(tpl_off[-1])
        feats = torch.randn(P, d, device=dev, generator=g)
        fn = ops.sqnorm_rows(feats)
        verts = torch.randn(P, 3, device=dev, generator=g)
        pts = torch.rand(Bq * Q, 2, device=dev, generator=g) * 518
        q_off = (torch.arange(Bq + 1, dtype=torch.int32) * Q).to(dev)
        tpl_ids = torch.randint(0, T, (Bq, n_top), generator=torch.Generator().manual_seed(2)).to(torch.int32).to(dev)
        feat_base = torch.zeros(Bq, dtype=torch.int32, device=dev)
        p_max = int(pc.max())
        from foundpose_amd._lib import cyclic_scratch_bytes
        scratch = torch.empty(cyclic_scratch_bytes(Bq * n_top, Q, p_max) // 8, dtype=torch.int64, device=dev)
        o = dict(counts=torch.empty(Bq, n_top, dtype=torch.int32, device=dev), q_ids=torch.empty(Bq, n_top, K, dtype=torch.int32, device=dev),
                 f_ids=torch.empty(Bq, n_top, K, dtype=torch.int32, device=dev), dists=torch.empty(Bq, n_top, K, device=dev),
                 conf=torch.empty(Bq, n_top, K, device=dev), c2d=torch.empty(Bq, n_top, K, 2, device=dev), c3d=torch.empty(Bq, n_top, K, 3, device=dev))
        ms = timeit(lambda: call("fp_cyclic_buddies", ptr(qf), ptr(qn), ptr(pts), ptr(q_off), Bq, Q, ptr(feats), ptr(fn), ptr(tpl_off), p_max,
                                 ptr(verts), ptr(tpl_ids), None, ptr(feat_base), n_top, d, K, K, ptr(scratch), ptr(o["counts"]), ptr(o["q_ids"]),
                                 ptr(o["f_ids"]), ptr(o["dists"]), ptr(o["conf"]), ptr(o["c2d"]), ptr(o["c3d"]), 1, stream()))
        live = float((pc[tpl_ids.cpu().long()].double() * Q).sum())
        print(f"cyclic_buddies {Bq} x {n_top} pairs, Q={Q}, P=300..450 (p_max {p_max}): {ms*1e3:8.1f} us whole call "
              f"(memset + distance tiles + selection)  {2.0*live*d/ms/1e9:7.1f} TF/s on the live distances", flush=True)
    if "ln" in what:
        x = torch.randn(M, D, device=dev)
        w, b = torch.randn(D, device=dev), torch.randn(D, device=dev)
        ms = timeit(lambda: ops.layernorm(x, w, b, torch.bfloat16))
        print(f"layernorm rows={M}: {ms*1e3:8.1f} us  {M*D*6/ms/1e6:7.1f} GB/s", flush=True)
    if "crop" in what:
        import numpy as np
        from foundpose_amd import crop_util
        cam = crop_util.PinholePlaneCameraModel(640, 480, (572.4114, 573.57043), (325.2611, 242.04899), np.eye(4))
        image = torch.rand(480, 640, 3, device=dev)
        rng = np.random.default_rng(0)
        boxes = [(l, t, l + rng.uniform(60, 220), t + rng.uniform(60, 170)) for l, t in zip(rng.uniform(0, 400, B), rng.uniform(0, 300, B))]
        masks = torch.zeros(B, 480, 640, dtype=torch.uint8, device=dev)
        cams = [crop_util.construct_crop_camera(crop_util.calc_crop_box(crop_util.AlignedBox2f(*b), make_square=True), cam, (518, 518), 0.2) for b in boxes]
        params = np.stack([crop_util.camera_pair_params(cam, c) for c in cams])
        idx = torch.zeros(B, dtype=torch.int32, device=dev)
        ms = timeit(lambda: crop_util._warp(image[None], crop_util.INTER_LINEAR, idx, params, (518, 518), True), iters=20)
        ms2 = timeit(lambda: crop_util._warp(masks, crop_util.INTER_NEAREST, None, params, (518, 518), True), iters=20)
        print(f"warp_crops B={B} 518x518 from 640x480: rgb {ms*1e3:8.1f} us ({B*3*518*518*4/ms/1e6:7.1f} GB/s written, incl. the "
              f"host->device copy of {params.nbytes} B of camera parameters), mask {ms2*1e3:8.1f} us", flush=True)


if __name__ == "__main__":
    main()
