#!/usr/bin/env python
"""Headline benchmark: detections/sec (ViT + descriptor matching) on 518x518 crops against a 10k-template
bank, one process per GPU.  Contract: see the task prompt / DESIGN.md "Measurement".

  python bench.py                       # 1 GPU, finishes in a few minutes
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N ...          # no launcher: starts the line above itself (self_launch) and returns its exit code

The workload is PLANTED (foundpose_amd/workload.py): every crop's fp32 features sit, with graded noise, in five
consecutive templates of the bank, so the expected output of a step is known and the line carries index-agreement
numbers ("parity") next to the throughput: against oracle A (the fp32 CPU restatement, on the detections the
cpu_baseline leg runs anyway) and against the library's own fp32 mode on every detection of the batch.
"""

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (guide: MI355X_MICROARCH.md chip table)
PEAK_FP8_TFLOPS = 5000.0    # dense fp8 (block-scaled MFMA), --precision fp8 only (BASELINE config 5; never the headline run)
PEAK_HBM_GBS = 8000.0       # HBM3E spec peak

# HBM-side bytes of one launch of the dominant kernel template from the rocprofv3 --pmc passes of THIS code (per the
# guide's gfx950 corrections); keyed by (version, size, batch, precision).  Source file + commit are reported next to it.
PMC_TRAFFIC = {
    # gemm_bf16_kernel<RESID_HILO> (256^2 tiles), average over the proj and fc2 launches of the pipeline's steps: (2 x FETCH_SIZE 267 327.5 KiB + WRITE_SIZE 178 620.8 KiB) x 1024
    # (rounds 4 and 5, same kernel: 730 395 136 / 730 395 443; the fp32-stream kernel it replaced wrote 266 565 KiB: 6 B per element of the stream against 4 now)
    ("vitl14-reg", 518, 32, "bf16"): 730394419,
}
PMC_TRAFFIC_SOURCE = ("profiles/r6_pmc_traffic.txt (tools/pmc_bench.sh: rocprofv3 --pmc over `python bench.py --skip-probes`, every counted launch belongs to a step; "
                      "the round-6 artefact pass; rounds 4 / 5: 730.4 MB, rounds 2-3 with the fp32-stream RESID kernel: 819.7 / 820.6 MB)")

# Matrix-pipe utilisation of the ViT forward as the counters report it: sum of SQ_VALU_MFMA_BUSY_CYCLES over the bf16 step's ViT launches /
# (1024 SIMDs x their GRBM_GUI_ACTIVE / 8 cycles), from the rocprofv3 --pmc pass of THIS code over `python bench.py --skip-probes`.  It is
# higher than the FLOP fraction of the nominal 2.5 PFLOP/s because the chip holds ~2.0 of its 2.4 GHz under this load (DVFS).  A duty cycle of the
# matrix pipe, NOT north_star's "MFMA utilisation" (that is roofline_vit_end_to_end.frac, by FLOPs: 0.36, target 0.40 not met).
PMC_MFMA_UTIL = {
    # per kernel (tools/pmc_mfma_aggregate.py; round 5: RESID_HILO 0.429 (144 launches x 512.8 k cycles), fc1 on 320-row tiles 0.469 (72 x 753.7 k; the hooked block's,
    # 256 rows: 0.447), attention 0.408 (76 x 563.7 k; 0.388 x 593.5 k before the K-row permutation took the lane exchanges out), qkv on 320-row tiles 0.499
    # (76 x 530.6 k), hooked block's 128^2 launches 0.268, patch embed 0.223, ln_finalize / rowstats_cast / hilo_rows 0; round 6 below: the same within a count)
    ("vitl14-reg", 518, 32, "bf16"): {"vit_forward": 0.426, "resid_gemm": 0.431, "fc1": 0.469, "qkv": 0.500, "attention": 0.409},
    # the f16 mode's kernels take the SAME cycle counts (RESID 504.8 k against 510.7 k, fc1 760.7 k / 753.1 k, attention 565.8 k / 562.1 k, qkv 529.2 k / 530.0 k):
    # its 3.5-4 % longer launches are a lower clock under the fp16 multipliers' power draw, not more cycles
    ("vitl14-reg", 518, 32, "f16"): {"vit_forward": 0.427, "resid_gemm": 0.436, "fc1": 0.464, "qkv": 0.501, "attention": 0.407},
}
PMC_MFMA_UTIL_SOURCE = "profiles/r6_bf16_pmc_mfma.txt, profiles/r6_f16_pmc_mfma.txt (tools/pmc_mfma.sh, the round-6 artefact pass)"


def synthetic_disc_patches(size):
    """Query patches of the default disc mask (radius 0.35 S)."""
    from foundpose_amd import synthetic
    return int(synthetic.make_disc_mask(size)[7::14, 7::14].sum())


def vit_flops_per_crop(arch, size, layer):
    np_ = (size // arch.patch) ** 2
    n = 1 + arch.registers + np_
    blk = 24 * n * arch.dim ** 2 + 4 * n * n * arch.dim
    return np_ * 3 * arch.patch ** 2 * arch.dim * 2 + (layer + 1) * blk


def time_kernel(fn, iters=20):
    """Average launch duration (ms) with HIP events on the stream the kernel runs on (torch's current stream)."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def self_launch(n):
    """Re-executes this script as `python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1 --master-port <free> bench.py
    <same arguments>` (the driver's own launch line) and returns its exit code.  stdout / stderr are inherited: rank 0's JSON line is ours."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this host driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def gemm_probe_ms(precision, M, mv, n, k, epi, fold, hilo, dev, iters=20):
    """Average launch duration (ms, HIP events) of one block GEMM of the ViT at M padded / mv live rows in `precision`'s own kernel: epi 0 = qkv (bias), 1 = fc1
    (GELU), 6 = fc1 (SwiGLU), 3 = the residual GEMMs proj / fc2 -- with folded LayerNorms (fold) the launch the pipeline issues: epilogue 8 on the (hi, lo) stream
    (hilo) or 7, and the normalising epilogues of qkv / fc1."""
    from foundpose_amd import ops
    from foundpose_amd._lib import call as _call, ptr as _ptr, stream as _stream
    dt16 = torch.float16 if precision == "f16" else torch.bfloat16
    f16_bit = (1 << 21) if precision == "f16" else 0   # FP_GEMM_F16
    a = torch.randn(M, k, device=dev).to(dt16)
    w = (torch.randn(n, k, device=dev) * 0.02).to(dt16)
    bias, gamma = torch.zeros(n, device=dev), torch.ones(n, device=dev)
    out = torch.zeros(M, n // 2 if epi == 6 else n, dtype=torch.float32 if epi == 3 else dt16, device=dev)
    if precision in ("f16x3", "f16f8"):   # split operands: the algorithmic FLOPs are those of the fp32 product
        pack = ops.splitx_pack if precision == "f16f8" else ops.split16_pack
        a3, w3 = pack(a.float(), 128.0, 64), pack(w.float(), ops.pow2_scale(w.float()), 64)
        o3 = torch.zeros(M, n, dtype=torch.float32, device=dev) if epi == 3 else torch.zeros(M, n if epi == 6 else 2 * n, dtype=torch.float16, device=dev)
        return time_kernel(lambda: ops.gemm_split(a3, w3, bias, 1e-6, gamma=gamma, out=o3, epilogue=epi, out_scale=64.0, m_valid=mv, f16f8=precision == "f16f8"), iters)
    if precision == "fp8":
        a8, w8 = ops.quantize_fp8(a, 50.0), ops.quantize_fp8(w, 5000.0)
        col = torch.full((n,), 1.0 / (50.0 * 5000.0), device=dev)
        return time_kernel(lambda: ops.gemm_fp8(a8, w8, bias, col, out=out, epilogue=epi, m_valid=mv), iters)
    if fold and epi == 3:   # the kernel the pipeline launches 36 times per step: the residual update on the (hi, lo) 16-bit stream + LayerNorm row
        xb = torch.zeros(M, n, dtype=dt16, device=dev)   # sums (epilogue 8; resid_hilo=False: fp32 stream + bf16 copy, epilogue 7)
        st = torch.zeros(n // 128, M, 2, device=dev)
        if hilo:
            xl = torch.zeros(M, n, dtype=dt16, device=dev)
            return time_kernel(lambda: _call("fp_gemm_bf16_ln", _ptr(a), a.stride(0), _ptr(w), w.stride(0), M, n, k, mv, _ptr(bias), _ptr(xl), n, 8 | f16_bit,
                                             None, None, _ptr(xb), n, _ptr(st), _stream()), iters)
        return time_kernel(lambda: _call("fp_gemm_bf16_ln", _ptr(a), a.stride(0), _ptr(w), w.stride(0), M, n, k, mv, _ptr(bias), _ptr(out), n, 7 | f16_bit,
                                         None, None, _ptr(xb), n, _ptr(st), _stream()), iters)
    if fold:                # ... and the normalising epilogues of qkv / fc1
        cs, ln_row = torch.zeros(n, device=dev), torch.ones(M, 2, device=dev)
        return time_kernel(lambda: ops.gemm_bf16_ln(a, w, bias, cs, ln_row, epilogue=epi, out=out, m_valid=mv), iters)
    return time_kernel(lambda: ops.gemm_bf16(a, w, bias, gamma=gamma, out=out, epilogue=epi, m_valid=mv), iters)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="crops per GPU per step")
    ap.add_argument("--templates", type=int, default=10000, help="templates per object")
    ap.add_argument("--objects", type=int, default=1, help="objects in the bank (BASELINE config 3: --objects 8 --templates 800 --batch 256)")
    ap.add_argument("--version", default="vitl14-reg")
    ap.add_argument("--layer", type=int, default=18)
    ap.add_argument("--size", type=int, default=518)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--tie-order", default="torch", choices=["torch", "canonical"],
                    help="'torch' (default, the headline): the reference's torch.topk tie order replayed on the device; 'canonical': (value, lowest index)")
    ap.add_argument("--graph", action="store_true", help="replay the ViT forward as one hipGraph (measured: no gain, the step is GPU-bound: 34.51 vs 34.44 ms)")
    ap.add_argument("--no-token-select", action="store_true",
                    help="run the hooked block on every token (default: only on the patch tokens the query points sample; same outputs bit for bit)")
    ap.add_argument("--mask", default="disc", choices=["disc", "full"],
                    help="detection masks: the centred disc of radius 0.35 S (SURVEY 8d, the headline) or the full crop (worst case: every patch is a query point)")
    ap.add_argument("--words", type=int, default=0, help="visual words of the bank (default 2048 = configs/gen_repre/lmo.json; with --mask full 4224, so that the 1369 "
                                                        "distinct textures of a full crop still get three instance words each, like the headline workload)")
    ap.add_argument("--overlap", action="store_true", help="matching of batch i on a second stream beside the backbone of batch i+1 (engine overlap_matching; measured +0.3...0.8 %%, not the default)")
    ap.add_argument("--parity-precision", default="f16x3", choices=["f16x3", "f16f8", "fp32", "none"],
                    help="the near-exact mode timed next to the headline as `parity_mode` (f16x3: split-fp16 operands, three fp16 MFMAs per product; "
                         "fp32: the exact-fp32 MFMA mode) with its index agreement against oracle A and the fp32 mode")
    ap.add_argument("--parity-steps", type=int, default=5)
    ap.add_argument("--no-parity-fast", action="store_true", help="skip `parity_mode_fast` (the f16f8 mode timed next to the f16x3 parity mode)")
    ap.add_argument("--no-mode-f16", action="store_true", help="skip `mode_f16` (the fp16-operand mode timed next to the bf16 headline)")
    ap.add_argument("--skip-probes", action="store_true", help="time the steps and stop: no roofline probe launches, no parity passes (tools/pmc_bench.sh: every "
                                                                "launch the counters see then belongs to a step of the pipeline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the fp32-mode agreement pass (the oracle comparison rides on the cpu baseline)")
    ap.add_argument("--cpu-detections", type=int, default=5, help="detections of the CPU baseline sample (BASELINE.md section 2: median of >= 5)")
    ap.add_argument("--other-configs", default="config3,config5_share",
                    help="the other single-GPU configurations of BASELINE.json, timed in the same run and reported under `other_configs` (N = 1 only, "
                         "5 steps each): config3 = 8 objects x 800 templates, batch 256; config5_share = one GPU's share of config 5 (ViT-g/14 fp8, 50 000 "
                         "templates, batch 128).  'none' skips them")
    ap.add_argument("--no-hard", action="store_true", help="skip the margin-free `parity.hard` workload (workload.py: only the best view planted)")
    ap.add_argument("--no-latency", action="store_true", help="skip the B = 1 per-detection latency (`latency_b1`) and the crop -> pose pipeline number (`pipeline`)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU under torch.distributed.run on
            # 127.0.0.1, a free port) and hand its exit code back; rank 0 of that job prints the one JSON line on our stdout.
            sys.exit(self_launch(args.gpus))
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} (or with no launcher at all)")
    # FP_BENCH_ONE_DEVICE=1 (+ FP_BENCH_BACKEND=gloo): dry run of the multi-rank path on a single-GPU box -- every rank
    # uses cuda:0 and the records travel through host memory.  Never set by the driver; the real runs use RCCL.
    if os.environ.get("FP_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ranks_seen = 1
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("FP_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from foundpose_amd import engine as fe
    from foundpose_amd import feature_util, ops, workload
    from foundpose_amd.bank import DeviceBank
    from foundpose_amd.vit_config import ARCHS

    arch = ARCHS[args.version]
    name = f"dinov2_version={args.version}_stride=14_facet=token_layer={args.layer}_norm=1"
    B = args.batch
    # ---- planted workload: the fp32 mode of the library produces the features that are planted (the reference's arithmetic)
    ex32 = feature_util.make_feature_extractor(name, random_init_seed=1234, precision="fp32").to(dev)
    full_mask = torch.ones(args.size, args.size, dtype=torch.uint8) if args.mask == "full" else None
    W_words = args.words or (4224 if args.mask == "full" else 2048)
    n_patches = (args.size // 14) ** 2 if args.mask == "full" else int(synthetic_disc_patches(args.size))
    wpt = workload.WORDS_PER_TEXTURE if W_words // workload.WORDS_PER_TEXTURE >= n_patches else 1
    wl = workload.build_planted_workload(ex32, B, args.size, args.objects, args.templates, 256, W_words, seed=7, crop_seed=rank, mask=full_mask,
                                         words_per_texture=wpt)
    bank = DeviceBank(wl.repres, device=dev)
    images, masks, det_obj = wl.crops, wl.masks, wl.det_obj     # inputs resident in HBM before timing
    extractor = feature_util.make_feature_extractor(name, random_init_seed=1234, precision=args.precision, use_graph=args.graph).to(dev)
    if args.precision == "fp8":
        extractor.calibrate_fp8(images)  # static activation scales are part of the fp8 model (no implicit calibration)
    eng = fe.FoundPoseEngine(extractor, bank, 14.0, 5, 300, tie_order=args.tie_order, overlap_matching=args.overlap)
    if args.no_token_select:
        eng.select_tokens = False
    select_on = extractor.supports_token_selection and eng.select_tokens and eng.fused_sample

    def step(e=eng, inp=None):
        res = e.infer_batch(*(inp if inp is not None else (images, masks, det_obj)))
        if e.overlap_matching:   # the record + exchange of this batch stay on the matching stream, beside the next batch's backbone
            with torch.cuda.stream(e.side_stream):
                rec = fe.pack_result(res)
                return fe.gather_records(rec, world), res
        rec = fe.pack_result(res)
        return fe.gather_records(rec, world), res   # the one exchange step (RCCL all-gather over xGMI)

    def timed(e, steps, inp=None):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step(e, inp)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
            every = torch.empty(world, dtype=torch.float64, device=t.device)
            dist.all_gather_into_tensor(every, t)          # every rank's own clock over the same K steps (between the same two barriers)
            per_rank_s[:] = every.cpu().tolist()
            el = float(every.max().item())                  # the contract's MAX over ranks
        else:
            per_rank_s[:] = [el]
        return el, out

    per_rank_s = []

    def gather_ms(steps=5):
        """The exchange step alone: HIP events around gather_records (RCCL all-gather over xGMI) on `steps` extra, untimed steps;
        -> mean ms on this rank.  Diagnoses a scaling curve: value(N) / (N value(1)) falls short either because a rank is slow
        (per_rank_ms_per_step spreads) or because the gather is (gather_ms grows with N)."""
        tot = 0.0
        for _ in range(steps):
            res = eng.infer_batch(images, masks, det_obj)
            rec = fe.pack_result(res)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fe.gather_records(rec, world)
            e1.record()
            e1.synchronize()
            tot += e0.elapsed_time(e1)
        return tot / steps

    for _ in range(args.warmup):
        step()
    elapsed, (gathered, last) = timed(eng, args.steps)
    det_per_s = world * B * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps
    rank_ms = [1e3 * v / args.steps for v in per_rank_s]
    multi = None
    if world > 1:   # diagnosability of the scaling run (the driver computes the efficiency itself from the per-N values)
        g_ms = torch.tensor([gather_ms()], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        g_all = torch.empty(world, dtype=torch.float64, device=g_ms.device)
        dist.all_gather_into_tensor(g_all, g_ms)
        multi = {"per_rank_ms_per_step": {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3), "all": [round(v, 3) for v in rank_ms]},
                 "gather_ms": {"mean_over_ranks": round(float(g_all.mean()), 4), "max_over_ranks": round(float(g_all.max()), 4),
                               "what": "HIP events around the one all-gather of result records, 5 extra untimed steps", "record_bytes_per_rank": None}}
    if world > 1:
        multi["gather_ms"]["record_bytes_per_rank"] = int(gathered.shape[0] // world * gathered.shape[1] * 4)
        ranks_seen = int(gathered.shape[0] // B)
        cnt = torch.ones(1, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(cnt)
        assert int(cnt.item()) == world == ranks_seen, "every rank must contribute its records to the gather"
    if args.skip_probes:
        if rank == 0:
            print(json.dumps({"metric": "detections/sec (ViT+kNN match) on 518^2 crops vs 10k-template bank", "value": round(det_per_s, 2), "unit": "detections/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "probes": "skipped", "ranks_seen": ranks_seen,
                              **({"multi_gpu": multi} if multi is not None else {})}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    # the other tie order, timed next to the headline (same inputs; not part of `value`)
    other = "canonical" if args.tie_order == "torch" else "torch"
    eng_o = fe.FoundPoseEngine(extractor, bank, 14.0, 5, 300, tie_order=other)
    step(eng_o)
    el_o, _ = timed(eng_o, 5)   # a fixed count: everything but the K timed steps is the same in every run (tools/rocprof_delta.py relies on it)
    ms_other = 1e3 * el_o / 5
    # token selection of the hooked block: the selected share, and the step without it next to the headline
    sel_info = {"enabled": bool(select_on)}
    n_tok_img = 1 + arch.registers + (args.size // 14) ** 2
    sel_total = B * n_tok_img
    if select_on:
        sel_total = eng._query_points_end(*eng._query_points_begin(masks, select_tokens=True))[3][3]
        eng.select_tokens = False
        step()
        el_ns, _ = timed(eng, 5)
        eng.select_tokens = True
        sel_info.update({"selected_tokens_per_crop": round(sel_total / B, 1), "tokens_per_crop": n_tok_img,
                         "ms_per_step_all_tokens": round(1e3 * el_ns / 5, 3),
                         "note": "the hooked block computes attention queries, proj and the MLP for the patch tokens under the sampling taps of the query "
                                 "points only (keys / values: all tokens); sampled features are bit-identical (tests/test_gpu_vit.py, tests/test_gpu_parity_e2e.py)"})

    # ---- the near-exact mode, driver-timed like the headline (same inputs, same engine code, `--parity-steps` steps after one warm-up)
    pm, ex_pm = None, None
    if args.parity_precision != "none" and args.parity_precision != args.precision:
        ex_pm = ex32 if args.parity_precision == "fp32" else feature_util.make_feature_extractor(name, random_init_seed=1234, precision=args.parity_precision).to(dev)
        eng_pm = fe.FoundPoseEngine(ex_pm, bank, 14.0, 5, 300, tie_order=args.tie_order)
        step(eng_pm)
        el_pm, (_, last_pm) = timed(eng_pm, args.parity_steps)
        pm = {"precision": args.parity_precision, "value": round(world * B * args.parity_steps / el_pm, 2), "unit": "detections/s",
              "ms_per_step": round(1e3 * el_pm / args.parity_steps, 3), "steps": args.parity_steps, "n_gpus": world,
              "what": ("split-fp16 operands (hi + lo: 22 mantissa bits), every product = three fp16 MFMAs with fp32 accumulation, fp32 residual stream / "
                       "LayerNorm / softmax / exact-erf GELU; the hooked block on the sampled tokens only, like the headline (bit-identical features)" if args.parity_precision == "f16x3"
                       else "as f16x3 with the two cross terms of every GEMM product (hi lo + lo hi, ~2^-11 of the product) on the fp8 MFMA: rows carry fp16 high halves + e4m3 copies of "
                            "hi and lo, 8 instead of 12 fp16-MFMA units per 64 k; q / k / v and the attention's own products stay three-fp16-MFMA" if args.parity_precision == "f16f8"
                       else "exact-fp32 MFMA GEMMs (k-ascending fmaf chains) + fp32 attention")}
    # ... and its faster variant (f16f8: the cross terms of every GEMM product on the fp8 pipe), driver-timed the same way -> `parity_mode_fast`
    pmf, ex_pmf = None, None
    if args.parity_precision == "f16x3" and args.precision not in ("f16x3", "f16f8") and not args.no_parity_fast:
        ex_pmf = feature_util.make_feature_extractor(name, random_init_seed=1234, precision="f16f8").to(dev)
        eng_pmf = fe.FoundPoseEngine(ex_pmf, bank, 14.0, 5, 300, tie_order=args.tie_order)
        step(eng_pmf)
        el_f, (_, last_pmf) = timed(eng_pmf, args.parity_steps)
        pmf = {"precision": "f16f8", "value": round(world * B * args.parity_steps / el_f, 2), "unit": "detections/s",
               "ms_per_step": round(1e3 * el_f / args.parity_steps, 3), "steps": args.parity_steps, "n_gpus": world,
               "what": "as f16x3 with the two cross terms of every GEMM product (hi lo + lo hi, ~2^-11 of the product) on the fp8 MFMA: rows carry the fp16 high halves + e4m3 "
                       "copies of hi and lo, 8 instead of 12 fp16-MFMA units per 64 k (a product good to ~14 bits at the worst); q / k / v and the attention's own products stay "
                       "three-fp16-MFMA"}
    # ... and the fp16-operand mode: the headline pipeline (same kernels, tiles, bytes) on IEEE fp16 operands, driver-timed the same way -> `mode_f16`
    pm16, ex_pm16 = None, None
    if args.precision == "bf16" and not args.no_mode_f16 and arch.dim % 128 == 0:
        ex_pm16 = feature_util.make_feature_extractor(name, random_init_seed=1234, precision="f16").to(dev)
        eng_pm16 = fe.FoundPoseEngine(ex_pm16, bank, 14.0, 5, 300, tie_order=args.tie_order)
        step(eng_pm16)
        step(eng_pm16)
        el_16, (_, last_pm16) = timed(eng_pm16, args.steps)
        pm16 = {"precision": "f16", "value": round(world * B * args.steps / el_16, 2), "unit": "detections/s", "ms_per_step": round(1e3 * el_16 / args.steps, 3),
                "steps": args.steps, "n_gpus": world, "vs_headline": round(world * B * args.steps / el_16 / det_per_s, 4),
                "what": "the headline pipeline -- folded LayerNorms, (hi, lo) residual stream, token-selected hooked block, the same kernel templates, tiles and operand bytes -- on "
                        "IEEE fp16 operands (v_mfma_f32_32x32x16_f16: 11 significant bits per operand instead of bf16's 8), GELU in its erf form; an activation beyond "
                        "+-65504 is reported, never silently wrong"}
    if rank == 0:
        n_tok = 1 + arch.registers + (args.size // 14) ** 2
        mv = B * n_tok
        M = extractor.padded_rows(mv)   # the pipeline's row padding: the launcher picks the same tile shapes here as inside a step
        peak_mfma = PEAK_FP8_TFLOPS if args.precision == "fp8" else PEAK_BF16_TFLOPS
        # ---- roofline of the dominant kernel = the largest time bucket of a step: the LayerScale+residual GEMM template
        # (gemm_bf16_kernel<LS_RESID>), launched twice per block: attn.proj (K = D) and mlp.fc2 (K = hidden)
        fold = getattr(extractor, "fold_layernorm", False)   # bf16: the block LayerNorms live inside these GEMMs (fp_vit_model.ln_fold)
        hilo = fold and getattr(extractor, "resid_hilo", False)   # ... and the residual stream in front of the hooked block is a (hi, lo) 16-bit pair
        from foundpose_amd._lib import call as _call, ptr as _ptr, stream as _stream

        dt16 = torch.float16 if args.precision == "f16" else torch.bfloat16
        gemm_ms = lambda n, k, epi: gemm_probe_ms(args.precision, M, mv, n, k, epi, fold, hilo, dev)
        hid = arch.hidden
        ms_proj, ms_fc2 = gemm_ms(arch.dim, arch.dim, 3), gemm_ms(arch.dim, hid, 3)
        ms_fc1 = gemm_ms(hid if arch.ffn == "mlp" else 2 * hid, arch.dim, 1 if arch.ffn == "mlp" else 6)
        ms_qkv = gemm_ms(3 * arch.dim, arch.dim, 0)
        fl = lambda n, k: 2.0 * mv * n * k     # algorithmic: valid rows only
        # attention of one block at the step's shape (all tokens as queries; the precision's own kernel), 4 N^2 d per (image, head)
        attn_info = None
        if args.precision in ("bf16", "f16", "fp8", "f16x3", "f16f8"):
            xq = torch.randn(M, 3 * arch.dim, device=dev)
            if args.precision in ("f16x3", "f16f8"):
                pk = torch.cat([ops.split16_pack(xq[:, i * arch.dim:(i + 1) * arch.dim].contiguous(), 16.0) for i in range(3)], dim=1)
                ms_attn = time_kernel(lambda: ops.attention_split(pk, B, n_tok, arch.dim, arch.heads, 16.0, 16.0))
            else:
                xq16 = xq.to(dt16)
                ms_attn = time_kernel(lambda: ops.attention(xq16, B, n_tok, arch.dim, arch.heads))
            attn_flops = 4.0 * B * n_tok * n_tok * arch.dim
            attn_info = {"kernel": "attn_split_kernel (three fp16 MFMAs per product; FLOPs of the fp32 product)" if args.precision in ("f16x3", "f16f8") else "attn_bf16_w64_kernel",
                         "bound": "mfma", "launch_ms": round(ms_attn, 4), "achieved": round(attn_flops / (ms_attn * 1e-3) / 1e12, 1), "peak": PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(attn_flops / (ms_attn * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4), "flops_per_launch": attn_flops,
                         "note": "back-to-back launches on random scores (inside the pipeline, behind the qkv GEMM, the same kernel measures 5-10 % faster: profiles/*_per_step_kernels.csv)"}
            del xq
        ls_flops, ls_ms = fl(arch.dim, arch.dim) + fl(arch.dim, hid), ms_proj + ms_fc2
        ach = ls_flops / (ls_ms * 1e-3) / 1e12
        # executed FLOPs: the hooked block runs qkv on all tokens and everything else on the selected ones
        D_, n_ = arch.dim, n_tok_img
        full_blk = 24 * n_ * D_ ** 2 + 4 * n_ * n_ * D_
        s_ = sel_total / B
        last_blk = 6 * n_ * D_ ** 2 + 4 * s_ * n_ * D_ + 18 * s_ * D_ ** 2
        flops_exec = vit_flops_per_crop(arch, args.size, args.layer) - full_blk + last_blk
        vit_tf = flops_exec * det_per_s / world / 1e12
        # ---- HBM roofline of the template retrieval (descriptors of the object's templates read once per 32 detections)
        from foundpose_amd._lib import call, cosine_scratch_floats, ptr, stream
        Bq = min(max(1, B // args.objects), 128)   # detections of one object in a batch (the kernel serves them in chunks of 32)
        desc_n = ops.normalize_rows(torch.rand(Bq, W_words, device=dev))
        seg = torch.tensor([0, Bq], dtype=torch.int32, device=dev)
        nt = torch.full((Bq,), args.templates, dtype=torch.int32, device=dev)
        sims = torch.empty(cosine_scratch_floats(Bq, args.templates), device=dev)
        sc, ids = torch.empty(Bq, 5, device=dev), torch.empty(Bq, 5, dtype=torch.int32, device=dev)
        tie_mode = 1 if args.tie_order == "torch" else 0
        knn = lambda mode: time_kernel(lambda: call("fp_cosine_topk", ptr(desc_n), ptr(seg), ptr(nt), Bq, Bq, ptr(bank.descs_n), ptr(bank.obj_tpl_off),
                                                    1, args.templates, W_words, 5, ptr(sims), ptr(sc), ptr(ids), mode, stream()), iters=50)  # the first object's templates
        ms_knn, ms_knn_other = knn(tie_mode), knn(1 - tie_mode)   # WARM: back-to-back calls over one 82 MB bank, which the 256 MiB Infinity Cache keeps on the die
        # COLD: the caches emptied before every call (2 GiB streamed through a 1 GiB buffer), one call per HIP-event pair -- the bank comes from HBM
        evict = torch.zeros(1 << 28, device=dev)

        def knn_cold(mode, iters=12):
            tot = 0.0
            for _ in range(iters):
                evict.add_(1.0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                call("fp_cosine_topk", ptr(desc_n), ptr(seg), ptr(nt), Bq, Bq, ptr(bank.descs_n), ptr(bank.obj_tpl_off), 1, args.templates, W_words, 5, ptr(sims), ptr(sc), ptr(ids),
                     mode, stream())
                e1.record()
                e1.synchronize()
                tot += e0.elapsed_time(e1)
            return tot / iters
        ms_knn_cold = knn_cold(tie_mode)
        del evict
        # IN THE PIPELINE: HIP events around the retrieval call inside 5 extra steps (behind the backbone and the word search of the same batch: tens of GB of
        # ViT traffic have passed through the caches since the bank was last read) -- the figure profiles/*_per_step_kernels.csv reproduces
        # (cosine_fused_kernel + cand_merge_replay_kernel) and the one `frac` is built on
        eng.record_stage_times = True
        ms_knn_pipe = 0.0
        for _ in range(5):
            fe.pack_result(eng.infer_batch(images, masks, det_obj))
            ms_knn_pipe += 1e3 * eng.retrieval_time() / 5
        eng.record_stage_times = False
        knn_obj_dets = max(list(det_obj).count(o) for o in set(det_obj))   # detections per object in the step's retrieval call
        knn_bytes = args.templates * W_words * 4 + Bq * W_words * 4 + Bq * args.templates * 4   # bank (read once) + queries + finished scores
        knn_flops = 2.0 * Bq * args.templates * W_words
        # the same call at BASELINE config 5's bank size (50 000 templates, one 32-detection pass): the stream is long enough to
        # amortise launch + ring fill, which dominate at 10 000 templates (82 MB = 10 us of HBM time)
        T5 = 50000
        bank5 = ops.normalize_rows(torch.rand(T5, 2048, device=dev))
        q5 = ops.normalize_rows(torch.rand(32, 2048, device=dev))
        seg5, tpl5 = torch.tensor([0, 32], dtype=torch.int32, device=dev), torch.tensor([0, T5], dtype=torch.int32, device=dev)
        nt5 = torch.full((32,), T5, dtype=torch.int32, device=dev)
        sims5 = torch.empty(cosine_scratch_floats(32, T5), device=dev)
        sc5, ids5 = torch.empty(32, 5, device=dev), torch.empty(32, 5, dtype=torch.int32, device=dev)
        ms_knn5 = time_kernel(lambda: call("fp_cosine_topk", ptr(q5), ptr(seg5), ptr(nt5), 32, 32, ptr(bank5), ptr(tpl5), 1, T5, 2048, 5, ptr(sims5),
                                           ptr(sc5), ptr(ids5), tie_mode, stream()), iters=20)
        knn5_bytes = T5 * 2048 * 4 + 32 * 2048 * 4 + 32 * T5 * 4
        del bank5, sims5
        key = (args.version, args.size, B, args.precision)
        # ---- the whole matching stage of a step: PCA projection + word k-NN + tf-idf + retrieval + 5 x cyclic matching, timed
        # with HIP events inside the engine on one extra step; algorithmic bytes by 8d's formula with the fp32 element size
        eng.record_stage_times = True
        fe.pack_result(eng.infer_batch(images, masks, det_obj))   # (rank 0 alone here: no collective)
        st_t = eng.stage_times()
        eng.record_stage_times = False
        sumQ = int(masks[:, 7::14, 7::14].sum())
        P_bar = bank.feats.shape[0] / max(1, bank.descs_n.shape[0])
        match_bytes = (args.objects * args.templates * W_words * 4 + B * 5 * P_bar * 256 * 4 + args.objects * W_words * 256 * 4 + sumQ * 256 * 4
                       + B * 5 * 300 * 12)
        match_flops = 2.0 * sumQ * W_words * 256 + 2.0 * B * args.templates * W_words + 5 * 2 * 2.0 * (sumQ / B) * P_bar * 256 * B
        ms_match = 1e3 * (st_t.get("corresp", 0.0))
        ms_projn = 1e3 * st_t.get("proj", 0.0)
        # proj is the one block GEMM whose floor is HBM, not the matrix pipe: fp32 residual read + write, bf16 copy, operands
        proj_bytes = mv * arch.dim * (2 + ((2 + 2) * 2 if hilo else 4 + 4 + (2 if fold else 0))) + arch.dim * arch.dim * 2   # A + stream read + stream write (+ bf16 copy)
        result = {
            "metric": "detections/sec (ViT+kNN match) on 518^2 crops vs 10k-template bank",
            "value": round(det_per_s, 2), "unit": "detections/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic (seeded crops assembled from 682 noise patch textures, disc masks, random-init ViT weights, planted bank: each crop's "
                                             "fp32 features sit with graded noise in 5 consecutive templates, the rest are random texture sets; words = instances of the textures, see config.workload)",
            "config": {"workload": f"{args.version} layer {args.layer} ({args.layer + 1} of {arch.depth} blocks executed, early exit after the hooked block" + (", the hooked block on the sampled tokens only" if select_on else "") + "), "
                                   f"{args.size}x{args.size} crops, batch {B}/GPU, {args.objects} object(s) x {args.templates} templates "
                                   f"(N_f={bank.feats.shape[0]}), {W_words} words ({wpt} per texture), PCA {arch.dim}->256, top-5 templates, top-300 buddies, {args.mask} mask Q={int(masks[0, 7::14, 7::14].sum())}, "
                                   f"tie order '{args.tie_order}'" + (" (the reference's torch.topk order, replayed on the device)" if args.tie_order == "torch" else ""),
                       "parallelism": f"detections sharded over {world} GPU(s), one RCCL all-gather of result records per step",
                       "tie_order": args.tie_order},
            "ranks_seen": ranks_seen,
            **({"multi_gpu": multi} if multi is not None else {}),
            "tie_order_cost": {args.tie_order + "_ms_per_step": round(ms_per_step, 3), other + "_ms_per_step": round(ms_other, 3)},
            "roofline": {"kernel": ("gemm_bf16_kernel<RESID_HILO> (attn.proj + mlp.fc2 of one ViT block, residual update on the (hi, lo) bf16 stream + LayerNorm row sums: the largest time bucket of a step)"
                                    if hilo else "gemm_bf16_kernel<RESID> (attn.proj + mlp.fc2 of one ViT block, residual update + bf16 copy + LayerNorm row sums: the largest time bucket of a step)"
                                    if fold else "gemm_bf16_kernel<LS_RESID> (attn.proj + mlp.fc2 of one ViT block: the largest time bucket of a step)"), "bound": "mfma",
                         "achieved": round(ach, 1), "peak": peak_mfma, "unit": "TFLOP/s", "frac": round(ach / peak_mfma, 4),
                         "traffic": PMC_TRAFFIC.get(key), "traffic_source": PMC_TRAFFIC_SOURCE if key in PMC_TRAFFIC else None,
                         "launch_ms": round(ls_ms / 2, 4), "launch_ms_proj": round(ms_proj, 4), "launch_ms_fc2": round(ms_fc2, 4),
                         "flops_per_launch": ls_flops / 2},
            "roofline_other_gemms": {"fc1": {"launch_ms": round(ms_fc1, 4), "frac": round(fl(hid if arch.ffn == "mlp" else 2 * hid, arch.dim) / (ms_fc1 * 1e-3) / 1e12 / peak_mfma, 4)},
                                     "qkv": {"launch_ms": round(ms_qkv, 4), "frac": round(fl(3 * arch.dim, arch.dim) / (ms_qkv * 1e-3) / 1e12 / peak_mfma, 4)},
                                     "proj": {"launch_ms": round(ms_proj, 4), "frac": round(fl(arch.dim, arch.dim) / (ms_proj * 1e-3) / 1e12 / peak_mfma, 4),
                                              "hbm_bound": {"bytes_per_launch": proj_bytes, "achieved_GBs": round(proj_bytes / (ms_proj * 1e-3) / 1e9, 1),
                                                            "frac_of_hbm_peak": round(proj_bytes / (ms_proj * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                                            "note": "K = D: 92 GF of matrix work against its operands + the residual stream's read-modify-write: "
                                                                    "this launch is bounded by HBM, not by the MFMA roofline"}},
                                     "fc2": {"launch_ms": round(ms_fc2, 4), "frac": round(fl(arch.dim, hid) / (ms_fc2 * 1e-3) / 1e12 / peak_mfma, 4)}},
            "roofline_attention": attn_info,
            "roofline_vit_end_to_end": {"bound": "mfma", "achieved": round(vit_tf, 1), "peak": peak_mfma, "unit": "TFLOP/s",
                                        "frac": round(vit_tf / peak_mfma, 4), "flops_per_detection": flops_exec,
                                        "flops_per_detection_all_tokens": vit_flops_per_crop(arch, args.size, args.layer),
                                        "mfma_busy_pmc": PMC_MFMA_UTIL.get(key), "mfma_busy_pmc_source": PMC_MFMA_UTIL_SOURCE if key in PMC_MFMA_UTIL else None},
            "token_selection": sel_info,
            "roofline_knn": {"kernel": f"fp_cosine_topk, tie order '{args.tie_order}' (template-descriptor streaming + top-5, whole call = cosine_fused_kernel + cand_merge_replay_kernel)", "bound": "hbm",
                             "achieved": round(knn_bytes / (ms_knn_pipe * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": round(knn_bytes / (ms_knn_pipe * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "launch_ms": round(ms_knn_pipe, 4),
                             "what": "`achieved` / `frac` / `launch_ms` are the call as it runs INSIDE a step (HIP events around it, mean of 5 steps; north_star target 0.60: not met); "
                                     "the cache-cold and the warm-cache probes of the same call are listed beside it",
                             "launch_ms_in_pipeline": round(ms_knn_pipe, 4), "launch_ms_cold": round(ms_knn_cold, 4), "launch_ms_warm_cache": round(ms_knn, 4),
                             "frac_cold": round(knn_bytes / (ms_knn_cold * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                             "frac_warm_cache": round(knn_bytes / (ms_knn * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                             "warm_cache_note": "back-to-back launches over the same 82 MB bank: it stays in the 256 MiB Infinity Cache, so this is not an HBM rate",
                             "launch_ms_warm_cache_" + other: round(ms_knn_other, 4), "bytes_per_launch": knn_bytes, "detections_per_launch": Bq, "detections_per_launch_in_pipeline": knn_obj_dets,
                             "fp32_mfma_tflops": round(knn_flops / (ms_knn * 1e-3) / 1e12, 1), "fp32_mfma_peak": 157.3,
                             "note": "exact-fp32 scores: at 32 detections per bank pass the op sits at the fp32-MFMA / HBM ridge (16 FLOP/B vs 19.7), "
                                     "more detections per object add passes (32 at a time) and make it MFMA-bound"},
            "roofline_matching_stage": {"what": "everything behind the sampled features in one step: word 3-NN, tf-idf, template retrieval, 5 x cyclic matching, record assembly "
                                                "(engine stage `corresp`); the PCA projection (stage `proj`) is listed beside it",
                                        "bound": "hbm", "ms_per_step": round(ms_match, 4), "ms_proj": round(ms_projn, 4), "bytes_per_step": int(match_bytes),
                                        "achieved": round(match_bytes / (ms_match * 1e-3) / 1e9, 1) if ms_match > 0 else None, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                        "frac": round(match_bytes / (ms_match * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if ms_match > 0 else None,
                                        "fp32_flops_per_step": match_flops, "fp32_mfma_floor_ms": round(match_flops / 157.3e12 * 1e3, 4),
                                        "frac_of_fp32_mfma_floor": round(match_flops / 157.3e12 * 1e3 / ms_match, 4) if ms_match > 0 else None,
                                        "note": "exact fp32 arithmetic (bit parity with the reference's searches): the stage's floor is the fp32-MFMA time of its three "
                                                "distance / score products, not its 0.16 GB of algorithmic bytes"},
            "roofline_knn_50k_templates": {"kernel": f"fp_cosine_topk, tie order '{args.tie_order}', 50 000 templates x 32 detections (BASELINE config 5's bank, random descriptors)",
                                           "bound": "hbm", "achieved": round(knn5_bytes / (ms_knn5 * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                           "frac": round(knn5_bytes / (ms_knn5 * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "launch_ms": round(ms_knn5, 4),
                                           "bytes_per_launch": knn5_bytes},
        }
        lists = [last.corresp_list(b) for b in range(B)]
        parity = {"tie_order": args.tie_order, "planted": workload.planted_stats(lists, wl.targets.tolist())}
        # the tail behind the path (not part of `value`): batched PnP-RANSAC + LM on the step's correspondences,
        # best of the 5 templates, against the planted poses (north_star: pose within 1e-4 relative on R, t)
        from foundpose_amd import pnp_util
        pnp = lambda: pnp_util.select_best_coarse(pnp_util.estimate_poses(last, [wl.K.numpy()] * B, "opencv", 400, 10.0, 0.99, True))
        best = pnp()
        ms_pnp = time_kernel(pnp, iters=5)
        eR = (best["R"].cpu() - wl.R).abs().amax(dim=(1, 2))
        et = (best["t"].cpu() - wl.t).norm(dim=1) / wl.t.norm(dim=1)
        parity["pose_vs_planted"] = {"found": int(best["found"].sum()), "max_abs_dR": float(eR.max()), "max_rel_dt": float(et.max()),
                                     "within_1e-4": int(((eR < 1e-4) & (et < 1e-4)).sum()), "pnp_ms_per_batch": round(ms_pnp, 3),
                                     "settings": "400 RANSAC iterations, 10 px, confidence 0.99, LM refinement (configs/infer/lmo.json)"}
        lists_pm = [last_pm.corresp_list(b) for b in range(B)] if pm is not None else None
        if pm is not None:
            pm["planted"] = workload.planted_stats(lists_pm, wl.targets.tolist())
        lists_pmf = [last_pmf.corresp_list(b) for b in range(B)] if pmf is not None else None
        if pmf is not None:
            pmf["planted"] = workload.planted_stats(lists_pmf, wl.targets.tolist())
        lists_pm16 = [last_pm16.corresp_list(b) for b in range(B)] if pm16 is not None else None
        if pm16 is not None:
            pm16["planted"] = workload.planted_stats(lists_pm16, wl.targets.tolist())
        if not args.no_parity:  # the library's fp32 mode on every detection of the batch (same bank, same tie order)
            eng32 = fe.FoundPoseEngine(ex32, bank, 14.0, 5, 300, tie_order=args.tie_order)
            res32 = eng32.infer_batch(images, masks, det_obj)
            lists32 = [res32.corresp_list(b) for b in range(B)]
            parity["vs_fp32_mode"] = workload.parity_stats(lists, lists32)
            if pm is not None and args.parity_precision != "fp32":
                pm["vs_fp32_mode"] = workload.parity_stats(lists_pm, lists32)
            if pmf is not None:
                pmf["vs_fp32_mode"] = workload.parity_stats(lists_pmf, lists32)
            if pm16 is not None:
                pm16["vs_fp32_mode"] = workload.parity_stats(lists_pm16, lists32)
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is timed at N = 1 only
            dbg = eng.infer_batch(images, masks, det_obj, keep_debug=True)   # one untimed step that keeps the visual-word ids (stage attribution)
            q_cnt = [int(masks[b, 7::14, 7::14].sum()) for b in range(B)]
            q_off = [0]
            for c_ in q_cnt:
                q_off.append(q_off[-1] + c_)
            dev_words = [dbg.word_ids[q_off[b]:q_off[b + 1]].cpu().numpy() for b in range(B)] if (args.size % 14 == 0 and dbg.word_ids is not None) else None
            result["cpu_baseline"], parity["vs_oracle_a"], ora, extra, oracle_feats = cpu_baseline(arch, args, bank, wl, lists, dev_words, want_oracle_b=args.precision in ("bf16", "fp8"))
            parity.update(extra)
            if pm is not None:
                pm["vs_oracle_a"] = workload.parity_stats(lists_pm[:len(ora)], ora)
            if pmf is not None:
                pmf["vs_oracle_a"] = workload.parity_stats(lists_pmf[:len(ora)], ora)
            if pm16 is not None:
                pm16["vs_oracle_a"] = workload.parity_stats(lists_pm16[:len(ora)], ora)
        else:
            oracle_feats = None
        if world == 1 and not args.no_hard:
            # ---- the margin-free workload: same crops / words / projector, only the best view planted, slots 2..5 = wrong views
            exs = {"fp32": ex32, args.precision: extractor}
            if ex_pm is not None and args.parity_precision in ("f16x3", "f16f8"):
                exs[args.parity_precision] = ex_pm
            if ex_pmf is not None:
                exs["f16f8"] = ex_pmf
            if ex_pm16 is not None:
                exs["f16"] = ex_pm16
            parity["hard"] = hard_parity(args, wl, exs, oracle_feats, dict(batch=B, full_mask=full_mask, W_words=W_words, wpt=wpt, rank=rank, dev=dev))
        if pm is not None:  # north_star's index bar, stated as booleans next to the mode's throughput
            pm["index_exact_vs_oracle_a"] = ("vs_oracle_a" in pm and pm["vs_oracle_a"]["corresp_equal"] == pm["vs_oracle_a"]["slots_compared"]
                                             and pm["vs_oracle_a"]["templates_equal"] == pm["vs_oracle_a"]["detections"]) if "vs_oracle_a" in pm else None
            pm["index_exact_vs_fp32_mode"] = (pm["vs_fp32_mode"]["corresp_equal"] == pm["vs_fp32_mode"]["slots_compared"]) if "vs_fp32_mode" in pm else None
            result["parity_mode"] = pm
        if pmf is not None:
            for key in ("vs_oracle_a", "vs_fp32_mode"):
                if key in pmf:
                    pmf["index_exact_" + key] = pmf[key]["corresp_equal"] == pmf[key]["slots_compared"] and pmf[key]["templates_equal"] == pmf[key]["detections"]
            result["parity_mode_fast"] = pmf
        if pm16 is not None:
            result["mode_f16"] = pm16
        result["parity"] = parity
        if world == 1 and not args.no_latency:
            result["latency_b1"], result["pipeline"] = latency_and_pipeline(args, wl, bank, extractor, eng, result.get("cpu_baseline"))
        if world == 1 and args.other_configs != "none":
            # free the headline workload first: config 5's share brings a 19-GB bank and a 1.1-B-parameter backbone in three precisions
            del eng, eng_o, extractor, ex32, bank, wl, images, masks, last, gathered
            if pm is not None:
                del eng_pm, ex_pm, last_pm
            if pmf is not None:
                del eng_pmf, ex_pmf, last_pmf
            if pm16 is not None:
                del eng_pm16, ex_pm16, last_pm16
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            result["other_configs"] = {c: other_config(c, args, dev, rank) for c in args.other_configs.split(",") if c}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(arch, args, bank, wl, gpu_lists, dev_words=None, want_oracle_b=False):
    """The oracle's reference-equivalent CPU path on the host cores, bounded sample of the same workload; its fp32
    features then go through the oracle's pinned matching arithmetic (reference tie order) and are compared index for
    index with what the GPU produced for the same detections."""
    from foundpose_amd import synthetic, workload
    from oracle import baseline
    # torch-CPU on ViT-sized matrices stops scaling (and regresses) far below the 256 hardware threads of the
    # GPU box's host; 32 threads is what the reference's own defaults would be tuned to on such a machine.
    cores = min(os.cpu_count() or 1, int(os.environ.get("FP_CPU_BASELINE_THREADS", "32")))
    torch.set_num_threads(cores)
    sd = synthetic.make_vit_state_dict(arch, seed=1234)
    repre = wl.repres[wl.det_obj[0]]
    proj = repre.feat_raw_projectors[0]
    cpu_bank = {
        "feat_vectors": repre.feat_vectors.cpu(), "feat_to_template_ids": repre.feat_to_template_ids.cpu(),
        "feat_cluster_centroids": repre.feat_cluster_centroids.cpu(), "feat_cluster_idfs": repre.feat_cluster_idfs.cpu(),
        "template_descs": repre.template_descs.cpu(), "pca_components": proj.components.cpu(), "pca_mean": proj.mean.cpu(),
    }
    n = min(args.cpu_detections, sum(1 for o in wl.det_obj if o == wl.det_obj[0]))
    imgs, msk = wl.crops[:n].cpu(), wl.masks[:n].cpu()
    tw = time.perf_counter()
    baseline.run_detection(sd, arch, args.layer, imgs[0], msk[0], cpu_bank)  # warm-up (thread pools, page-in)
    if time.perf_counter() - tw > 15.0:
        n = 1  # keep the default bench run within a few minutes on slow hosts
    stages, feats, per_det = {}, [], []
    for i in range(n):
        t0 = time.perf_counter()
        t, _, qpf = baseline.run_detection(sd, arch, args.layer, imgs[i], msk[i], cpu_bank, return_features=True)
        per_det.append(time.perf_counter() - t0)
        feats.append(qpf)
        for k, v in t.items():
            stages[k] = stages.get(k, 0.0) + v / n
    med = sorted(per_det)[len(per_det) // 2] if len(per_det) % 2 else 0.5 * (sorted(per_det)[len(per_det) // 2 - 1] + sorted(per_det)[len(per_det) // 2])
    base = {"value": round(1.0 / med, 4), "unit": "detections/s", "cores": cores, "kind": "port",
            "sample": f"median of {n} detection(s) after 1 warm-up (BASELINE.md section 2), batch of one, fp32 torch-CPU on {cores} threads, all {arch.depth} blocks run like the reference",
            "s_per_detection": [round(v, 3) for v in per_det], "mean_value": round(n / sum(per_det), 4),
            "s_per_stage": {k: round(v, 4) for k, v in stages.items()}}
    # BASELINE.md section 2 words the plan as torch.set_num_threads(os.cpu_count()); the figure above uses 32 threads because torch-CPU collapses
    # beyond that on the GPU boxes' hosts (a whole detection on all 256 hardware threads took 78 s on the first round-5 box, against 2.2 s on 32).
    # Both are reported, the all-core one on a BOUNDED sample so that the run stays within minutes: the first two ViT blocks of one detection
    # (embedding included) at every thread count of the sweep, extrapolated to the 24 blocks + the matching stage measured above.
    all_cores = os.cpu_count() or 1
    if all_cores > cores and os.environ.get("FP_CPU_BASELINE_ALL_CORES", "1") != "0":
        from oracle import vit as ov
        sweep = {}
        for nt in sorted({cores, 64, 128, all_cores}):
            if nt > all_cores:
                continue
            torch.set_num_threads(nt)
            with torch.no_grad():
                ov.extractor_forward(sd, arch, imgs[:1], 0, True)                       # warm-up: thread pool at this width
                t0 = time.perf_counter()
                ov.extractor_forward(sd, arch, imgs[:1], 1, True)                       # embedding + blocks 0, 1
                sweep[nt] = time.perf_counter() - t0
        torch.set_num_threads(cores)
        rest = sum(v for k, v in stages.items() if k != "feat_extract")
        est = lambda s2: 1.0 / (s2 * arch.depth / 2.0 + rest)
        base["all_cores"] = {"cores": all_cores, "unit": "detections/s", "value_extrapolated": round(est(sweep[all_cores]), 4),
                             "s_two_blocks_by_threads": {str(k): round(v, 3) for k, v in sweep.items()},
                             "extrapolated_value_by_threads": {str(k): round(est(v), 4) for k, v in sweep.items()},
                             "sample": f"bounded: embedding + the first 2 of {arch.depth} ViT blocks of one detection per thread count, x {arch.depth}/2 + the measured "
                                       f"non-ViT stages; torch.set_num_threads(os.cpu_count() = {all_cores}) is BASELINE.md section 2's wording, {cores} threads is what the "
                                       "headline cpu_baseline uses because torch-CPU regresses beyond it on this host"}
    # ---- oracle A on those detections: same fp32 features -> pinned matching arithmetic, the reference's tie order
    off = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(torch.bincount(cpu_bank["feat_to_template_ids"].long(), minlength=repre.template_descs.shape[0]), 0)])
    small = {"feat_cluster_centroids": cpu_bank["feat_cluster_centroids"].numpy(), "feat_cluster_idfs": cpu_bank["feat_cluster_idfs"].numpy(),
             "template_descs": cpu_bank["template_descs"].numpy(), "template_desc_opts": repre.template_desc_opts._asdict()}
    fetch = lambda tid: (cpu_bank["feat_vectors"][int(off[tid]):int(off[tid + 1])].numpy(), int(off[tid]))
    mode = "torch" if args.tie_order == "torch" else "canonical"
    ora, words_a = zip(*[baseline.exact_matching(qp.numpy(), qf.numpy(), small, fetch, 5, 300, mode, return_words=True) for qp, qf in feats])
    ora = list(ora)
    par = workload.parity_stats(gpu_lists[:n], ora)
    par["oracle"] = f"oracle A: fp32 CPU features of {n} detection(s) through oracle/match.py (pinned to the reference fixtures), tie order '{args.tie_order}'"
    import numpy as np
    srt = lambda ws: [np.sort(np.asarray(w), axis=1) for w in ws]   # hard assignment: a patch's words count as a set
    par["by_stage"] = workload.stage_flips(gpu_lists[:n], ora, srt(dev_words[:n]) if dev_words is not None else None, srt(words_a) if dev_words is not None else None)
    extra = {}
    if want_oracle_b:
        # oracle B (hard part 2): the same detections with every GEMM / attention operand rounded to bf16 at the device's cast
        # points (oracle/vit.py quant="bf16"), fp32 accumulation, then the fp32 matching arithmetic -- what the bf16 mode is a
        # realisation of.  The device differs from it by summation order, the folded LayerNorm's rounding points and the GELU polynomial
        # (features within 1.5e-2, tests/test_gpu_vit.py), so on a workload WITHOUT engineered margins agreement is a rate, attributed per stage;
        # on the margin-verified fixture it is exact (tests/test_gpu_parity_e2e.py::test_bf16_mode_index_exact_vs_oracle_b_on_fixture_with_verified_margins)
        nb = min(n, 3)
        pc, pm_ = (cpu_bank["pca_components"], cpu_bank["pca_mean"])
        fb = [baseline.oracle_a_features(sd, arch, args.layer, imgs[i], msk[i], pc, pm_, quant="bf16") for i in range(nb)]
        orb, words_b = zip(*[baseline.exact_matching(qp.numpy(), qf.numpy(), small, fetch, 5, 300, mode, return_words=True) for qp, qf in fb])
        vb = workload.parity_stats(gpu_lists[:nb], list(orb))
        vb["by_stage"] = workload.stage_flips(gpu_lists[:nb], list(orb), srt(dev_words[:nb]) if dev_words is not None else None, srt(words_b) if dev_words is not None else None)
        vb["oracle"] = f"oracle B: bf16-operand CPU features (oracle/vit.py quant='bf16') of {nb} detection(s) through oracle/match.py, tie order '{args.tie_order}'"
        vb["oracle_b_vs_oracle_a"] = workload.parity_stats(list(orb), ora[:nb])   # how far bf16 operands alone move the reference's answer
        extra["vs_oracle_b"] = vb
    return base, par, ora, extra, feats


def oracle_lists(feats, repre, tie_order):
    """Oracle A's matching half (oracle/match.py through baseline.exact_matching) on ready-made oracle features against `repre`.
    -> (per-detection correspondence lists, per-detection word ids)."""
    from oracle import baseline
    f2t = repre.feat_to_template_ids.cpu()
    fv = repre.feat_vectors.cpu()
    off = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(torch.bincount(f2t.long(), minlength=repre.template_descs.shape[0]), 0)])
    small = {"feat_cluster_centroids": repre.feat_cluster_centroids.cpu().numpy(), "feat_cluster_idfs": repre.feat_cluster_idfs.cpu().numpy(),
             "template_descs": repre.template_descs.cpu().numpy(), "template_desc_opts": repre.template_desc_opts._asdict()}
    fetch = lambda tid: (fv[int(off[tid]):int(off[tid + 1])].numpy(), int(off[tid]))
    mode = "torch" if tie_order == "torch" else "canonical"
    ora, words = zip(*[baseline.exact_matching(qp.numpy(), qf.numpy(), small, fetch, 5, 300, mode, return_words=True) for qp, qf in feats])
    return list(ora), list(words)


def hard_parity(args, wl, extractors, oracle_feats, ctx):
    """`parity.hard`: index agreement where nothing has an engineered margin (foundpose_amd/workload.py, HARD_*): the crops, masks, visual words,
    projector and poses of the headline workload, but only each detection's BEST view is planted; templates 2..5 of every retrieval are unrelated
    texture sets (wrong views, as on real data), so most query patches have no counterpart there.  Every extractor mode runs the same batch; each is
    compared with oracle A (on the detections the CPU baseline computed features for) and with the library's fp32 mode (all detections), with the
    per-stage attribution of workload.stage_flips.  Rates, not assertions: on a margin-free input even two fp32 implementations differ."""
    import numpy as np
    from foundpose_amd import engine as fe, workload
    from foundpose_amd.bank import DeviceBank
    B, dev = ctx["batch"], ctx["dev"]
    wlh = workload.build_planted_workload(extractors["fp32"], B, args.size, args.objects, args.templates, 256, ctx["W_words"], seed=7, crop_seed=ctx["rank"],
                                          mask=ctx["full_mask"], words_per_texture=ctx["wpt"], hard=True)
    assert torch.equal(wlh.crops, wl.crops) and torch.equal(wlh.masks, wl.masks), "the hard workload reuses the headline crops"
    bank_h = DeviceBank(wlh.repres, device=dev)
    q_cnt = [int(wlh.masks[b, 7::14, 7::14].sum()) for b in range(B)]
    q_off = np.concatenate([[0], np.cumsum(q_cnt)])
    srt = lambda ws: [np.sort(np.asarray(w), axis=1) for w in ws]
    lists, words, results = {}, {}, {}
    for prec, ex in extractors.items():
        res = fe.FoundPoseEngine(ex, bank_h, 14.0, 5, 300, tie_order=args.tie_order).infer_batch(wlh.crops, wlh.masks, wlh.det_obj, keep_debug=True)
        results[prec] = res
        lists[prec] = [res.corresp_list(b) for b in range(B)]
        words[prec] = srt([res.word_ids[q_off[b]:q_off[b + 1]].cpu().numpy() for b in range(B)]) if (args.size % 14 == 0 and res.word_ids is not None) else None
    out = {"workload": f"as config.workload, but only template t_b of each detection is planted (features + {workload.HARD_NOISE[0]} sigma noise, "
                       f"{workload.HARD_PATCHES[0]} of the query patches); slots 2..5 are retrieved among unrelated random texture sets: no engineered margins",
           "tie_order": args.tie_order}
    for prec in lists:
        out[prec + "_planted_top1"] = workload.planted_stats(lists[prec], wlh.targets.tolist(), n_planted=1)["planted_top1"]
    if oracle_feats is not None:
        n = len(oracle_feats)
        ora, words_a = oracle_lists(oracle_feats, wlh.repres[wlh.det_obj[0]], args.tie_order)
        out["oracle_a_planted_top1"] = workload.planted_stats(ora, wlh.targets.tolist()[:n], n_planted=1)["planted_top1"]
        for prec in lists:
            st = workload.parity_stats(lists[prec][:n], ora)
            st["by_stage"] = workload.stage_flips(lists[prec][:n], ora, words[prec][:n] if words[prec] is not None else None, srt(words_a) if words[prec] is not None else None)
            out[prec + "_vs_oracle_a"] = st
    for prec in lists:
        if prec != "fp32":
            st = workload.parity_stats(lists[prec], lists["fp32"])
            st["by_stage"] = workload.stage_flips(lists[prec], lists["fp32"], words[prec], words["fp32"])
            out[prec + "_vs_fp32_mode"] = st
    out["pose_under_noise"] = pose_under_noise(wlh, bank_h, results, B)
    return out


def pose_under_noise(wl, bank, results, B, sigmas=(0.5, 2.0)):
    """north_star's third clause -- the final pose within 1e-4 relative on R, t -- measured where it can fail: the planted 2D-3D pairs carry sigma px of
    reprojection noise (workload.noisy_vertices), so a mode whose correspondence indices differ from the fp32 mode's feeds the PnP-RANSAC tail
    (/root/reference/scripts/infer.py:552-602, utils/pnp_util.py:20-84; here csrc/pnp.hip with a fixed seed) another inlier set and gets another pose.
    Per sigma and mode: best coarse pose of every detection against the fp32 mode's on the same noisy bank, and the fp32 mode's own distance from the
    planted pose (the noise floor the differences sit on)."""
    from foundpose_amd import pnp_util, workload
    out = {"what": "planted vertices re-derived from query pixels displaced by N(0, sigma^2) px; 400 RANSAC iterations, 10 px, confidence 0.99, LM refinement, seed 0; "
                   "poses compared over all detections: max |dR| (entries), max relative |dt|, detections within 1e-4 on both, detections with identical poses"}
    cams = [wl.K.numpy()] * B
    for sg in sigmas:
        V = workload.noisy_vertices(wl, sg, seed=11)
        best = {prec: pnp_util.select_best_coarse(pnp_util.estimate_poses(workload.with_vertices(res, bank, V, wl.det_obj), cams, "opencv", 400, 10.0, 0.99, True))
                for prec, res in results.items()}
        entry = {}
        if "fp32" in best:
            truth = {"found": torch.ones(B, dtype=torch.bool, device=best["fp32"]["R"].device), "R": wl.R.to(best["fp32"]["R"].device), "t": wl.t.to(best["fp32"]["R"].device)}
            entry["fp32_mode_vs_planted_pose"] = workload.pose_agreement(best["fp32"], truth)
            for prec in best:
                if prec != "fp32":
                    entry[prec + "_vs_fp32_mode"] = workload.pose_agreement(best[prec], best["fp32"])
        out[f"sigma_{sg}px"] = entry
    return out


def latency_and_pipeline(args, wl, bank, extractor, eng, cpu_base):
    """What a maintainer runs, timed (extra keys, not part of `value`):
    latency_b1 -- ONE detection at a time through the drop-in per-detection calls in the reference loop's shape (scripts/infer.py:468-542: extractor(image),
      filter_points_by_mask, sample_feature_map_at_points, project_features, establish_correspondences), host-synchronous like the reference, per stage with
      the reference's `times` keys, next to the CPU baseline's s_per_stage;
    pipeline -- the 32-detection batch from the UNCROPPED image: crop producer (infer.py:411-450) -> extractor + matching -> PnP-RANSAC tail (infer.py:552-602)."""
    from foundpose_amd import corresp_util, crop_util, feature_util, pnp_util, projector_util
    import numpy as np
    B, S = wl.crops.shape[0], args.size
    repre = wl.repres[wl.det_obj[0]]
    grid = feature_util.generate_grid_points((S, S), 14.0).cuda()
    keys = ("feat_extract", "grid_sample", "proj", "corresp")
    tot = {k: [] for k in keys}
    n_lat = min(B, 12)

    def one(b):
        t = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fmap = extractor(wl.crops[b:b + 1])["feature_maps"][0]                        # infer.py:470-471
        torch.cuda.synchronize()
        t["feat_extract"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        qp = feature_util.filter_points_by_mask(grid, wl.masks[b])                     # infer.py:478
        qf = feature_util.sample_feature_map_at_points(fmap, qp, (S, S)).contiguous()   # infer.py:494-498
        torch.cuda.synchronize()
        t["grid_sample"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        qf = projector_util.project_features(qf, repre.feat_raw_projectors).contiguous()   # infer.py:507-510
        torch.cuda.synchronize()
        t["proj"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        c = corresp_util.establish_correspondences(qp, qf, repre, "tfidf", "cyclic_buddies", 5, 300)   # infer.py:531-542
        torch.cuda.synchronize()
        t["corresp"] = time.perf_counter() - t0
        return t, c
    one(0)
    one(1 % B)   # warm-up: the B = 1 workspace, the per-object device bank of the drop-in functions
    for b in range(n_lat):
        t, _ = one(b)
        for k in keys:
            tot[k].append(t[k])
    med = lambda v: float(np.median(v))
    lat = {"what": "batch of ONE through the drop-in per-detection calls (reference loop shape, scripts/infer.py:468-542), host-synchronous after every stage; "
                   f"median of {n_lat} detections after 2 warm-ups; the whole feature map is produced (no token selection in the plain extractor call)",
           "precision": extractor.precision, "ms_per_stage": {k: round(1e3 * med(tot[k]), 3) for k in keys},
           "ms_per_detection": round(1e3 * sum(med(tot[k]) for k in keys), 3),
           "detections_per_s": round(1.0 / sum(med(tot[k]) for k in keys), 1)}
    if cpu_base is not None:
        lat["cpu_baseline_s_per_stage"] = cpu_base.get("s_per_stage")
    # ---- crop -> pose: one synthetic 'scene' whose B detections' crop boxes tile an uncropped image built from the headline crops
    cols = int(np.ceil(np.sqrt(B)))
    rows = (B + cols - 1) // cols
    Hs, Ws = rows * S, cols * S
    img = torch.zeros(Hs, Ws, 3, device="cuda")
    masks = torch.zeros(B, Hs, Ws, dtype=torch.uint8, device="cuda")
    boxes = []
    for b in range(B):
        r, c = divmod(b, cols)
        img[r * S:(r + 1) * S, c * S:(c + 1) * S] = wl.crops[b].permute(1, 2, 0)
        masks[b, r * S:(r + 1) * S, c * S:(c + 1) * S] = wl.masks[b]
        boxes.append([c * S + 0.1 * S, r * S + 0.1 * S, (c + 1) * S - 0.1 * S, (r + 1) * S - 0.1 * S])
    cam = crop_util.PinholePlaneCameraModel(Ws, Hs, (1.2 * Ws, 1.2 * Ws), (Ws / 2.0, Hs / 2.0), np.eye(4))

    def pipe():
        res, cams = eng.infer_detections(img, masks, boxes, cam, (S, S), 0.2, wl.det_obj)
        return pnp_util.select_best_coarse(pnp_util.estimate_poses(res, cams, "opencv", 400, 10.0, 0.99, True))

    def stages():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        crops, cmasks, cams = crop_util.crop_detections(img, masks, boxes, cam, (S, S), 0.2)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        res = eng.infer_batch(crops, cmasks, wl.det_obj)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        pnp_util.select_best_coarse(pnp_util.estimate_poses(res, cams, "opencv", 400, 10.0, 0.99, True))
        torch.cuda.synchronize()
        return t1 - t0, t2 - t1, time.perf_counter() - t2
    pipe()
    pipe()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_pipe = 5
    for _ in range(n_pipe):
        best = pipe()
    torch.cuda.synchronize()
    ms_pipe = 1e3 * (time.perf_counter() - t0) / n_pipe
    st = np.median(np.array([stages() for _ in range(3)]), axis=0)
    pipe_info = {"what": f"{B} detections from ONE uncropped {Ws}x{Hs} image: crop producer (calc_crop_box, crop camera, remap of image and mask: infer.py:411-450) -> "
                         "extractor + matching (the timed step of `value`) -> batched PnP-RANSAC + LM, best of 5 templates (infer.py:552-602); synthetic scene tiled from the "
                         "headline crops (the re-sampled crops are new images: poses are not checked here)",
                 "pipeline_ms_per_step": round(ms_pipe, 3), "detections_per_s": round(B / (ms_pipe * 1e-3), 1),
                 "ms_crop_producer": round(1e3 * st[0], 3), "ms_infer_batch": round(1e3 * st[1], 3), "ms_pnp": round(1e3 * st[2], 3),
                 "poses_found": int(best["found"].sum())}
    return lat, pipe_info


OTHER_CONFIGS = {
    "config3": dict(version="vitl14-reg", layer=18, precision="bf16", objects=8, templates=800, batch=256,
                    what="BASELINE config 3: ViT-L/14 bf16, full LM-O-sized bank (8 objects x 800 templates) in HBM, batch 256, 1 GPU"),
    "config5_share": dict(version="vitg14-reg", layer=39, precision="fp8", objects=1, templates=50000, batch=128,
                          what="one GPU's share of BASELINE config 5: ViT-g/14 fp8 (CDNA4 e4m3 MFMA), 50 000-template bank, batch 128 of the 1024"),
}


def other_config(label, args, dev, rank, steps=5, parity_steps=3):
    """One of BASELINE.json's other single-GPU configurations, timed by the same clock in the same run: `steps` steps after 2 warm-ups in the configuration's
    precision, `parity_steps` in the near-exact f16x3 mode, one pass of the library's fp32 mode for the index agreement of both."""
    from foundpose_amd import engine as fe, feature_util, synthetic, workload
    from foundpose_amd.bank import DeviceBank
    from foundpose_amd.vit_config import ARCHS
    import gc
    c = OTHER_CONFIGS[label]
    arch = ARCHS[c["version"]]
    name = f"dinov2_version={c['version']}_stride=14_facet=token_layer={c['layer']}_norm=1"
    sd = synthetic.make_vit_state_dict(arch, seed=1234)       # generated once, shared by the three precisions
    B = c["batch"]
    ex32 = feature_util.make_feature_extractor(name, state_dict=sd, precision="fp32").to(dev)
    wl = workload.build_planted_workload(ex32, B, args.size, c["objects"], c["templates"], 256, 2048, seed=7, crop_seed=rank)
    bank = DeviceBank(wl.repres, device=dev)
    inp = (wl.crops, wl.masks, wl.det_obj)

    def run(ex, n_steps, warm):
        eng = fe.FoundPoseEngine(ex, bank, 14.0, 5, 300, tie_order=args.tie_order)
        for _ in range(warm):
            fe.pack_result(eng.infer_batch(*inp))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            res = eng.infer_batch(*inp)
            fe.pack_result(res)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        return round(B * n_steps / el, 2), round(1e3 * el / n_steps, 3), [res.corresp_list(b) for b in range(B)]
    res32 = fe.FoundPoseEngine(ex32, bank, 14.0, 5, 300, tie_order=args.tie_order).infer_batch(*inp)
    lists32 = [res32.corresp_list(b) for b in range(B)]
    del res32
    ex32 = None
    gc.collect()
    torch.cuda.empty_cache()
    ex = feature_util.make_feature_extractor(name, state_dict=sd, precision=c["precision"]).to(dev)
    if c["precision"] == "fp8":
        ex.calibrate_fp8(wl.crops)
    val, ms, lists = run(ex, steps, 2)
    # roofline of the configuration's dominant kernel template (the residual GEMMs proj + fc2 of one block, as for the headline), measured live at ITS shapes
    n_tok = 1 + arch.registers + (args.size // 14) ** 2
    mv = B * n_tok
    Mp = ex.padded_rows(mv)
    fold = getattr(ex, "fold_layernorm", False)
    hilo = fold and getattr(ex, "resid_hilo", False)
    ms_proj = gemm_probe_ms(c["precision"], Mp, mv, arch.dim, arch.dim, 3, fold, hilo, dev, iters=5)
    ms_fc2 = gemm_probe_ms(c["precision"], Mp, mv, arch.dim, arch.hidden, 3, fold, hilo, dev, iters=5)
    ms_fc1 = gemm_probe_ms(c["precision"], Mp, mv, arch.hidden if arch.ffn == "mlp" else 2 * arch.hidden, arch.dim, 1 if arch.ffn == "mlp" else 6, fold, hilo, dev, iters=5)
    peak = PEAK_FP8_TFLOPS if c["precision"] == "fp8" else PEAK_BF16_TFLOPS
    fl = lambda n, k: 2.0 * mv * n * k
    ls_tf = (fl(arch.dim, arch.dim) + fl(arch.dim, arch.hidden)) / ((ms_proj + ms_fc2) * 1e-3) / 1e12
    fc1_tf = fl(arch.hidden if arch.ffn == "mlp" else 2 * arch.hidden, arch.dim) / (ms_fc1 * 1e-3) / 1e12
    flops_det = vit_flops_per_crop(arch, args.size, c["layer"])
    roof = {"kernel": ("gemm_bf16_kernel<..., F8> LS_RESID" if c["precision"] == "fp8" else "gemm_bf16_kernel<RESID_HILO>") + " (attn.proj + mlp.fc2 of one block, this configuration's rows)",
            "bound": "mfma", "achieved": round(ls_tf, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ls_tf / peak, 4), "launch_ms": round((ms_proj + ms_fc2) / 2, 4),
            "launch_ms_proj": round(ms_proj, 4), "launch_ms_fc2": round(ms_fc2, 4), "flops_per_launch": (fl(arch.dim, arch.dim) + fl(arch.dim, arch.hidden)) / 2,
            "fc1": {"launch_ms": round(ms_fc1, 4), "achieved": round(fc1_tf, 1), "frac": round(fc1_tf / peak, 4)},
            "vit_end_to_end": {"achieved": round(flops_det * val / 1e12, 1), "frac": round(flops_det * val / 1e12 / peak, 4), "flops_per_detection_all_tokens": flops_det,
                               "note": "all-token FLOPs of the executed blocks x detections/s (the hooked block runs on the sampled tokens only: an upper bound on the executed fraction)"},
            "profile": f"profiles/r6_{label}_per_step_kernels.csv (tools/profile_bench.sh), profiles/r6_{label}_pmc_mfma.txt (tools/pmc_mfma.sh)"}
    out = {"what": c["what"], "roofline": roof, "value": val, "unit": "detections/s", "ms_per_step": ms, "steps": steps, "warmup": 2, "dtype": c["precision"], "n_gpus": 1,
           "workload": f"{c['version']} layer {c['layer']}, {args.size}x{args.size} crops, batch {B}, {c['objects']} object(s) x {c['templates']} templates (N_f={bank.feats.shape[0]}), "
                       f"2048 words, disc masks, tie order '{args.tie_order}', planted like the headline workload",
           "planted": workload.planted_stats(lists, wl.targets.tolist()), "vs_fp32_mode": workload.parity_stats(lists, lists32)}
    if c["precision"] == "fp8":
        n16, n8 = ex.saturation_counts()
        out["fp8_clamped_threads"] = n8
    ex = None
    gc.collect()
    torch.cuda.empty_cache()
    for key, prec in (("mode_f16", "f16"), ("parity_mode", "f16x3"), ("parity_mode_fast", "f16f8")):
        if prec == "f16" and args.no_mode_f16:
            continue
        ex3 = feature_util.make_feature_extractor(name, state_dict=sd, precision=prec).to(dev)
        v3, ms3, lists3 = run(ex3, steps if prec == "f16" else parity_steps, 2 if prec == "f16" else 1)
        st3 = workload.parity_stats(lists3, lists32)
        out[key] = {"precision": prec, "value": v3, "unit": "detections/s", "ms_per_step": ms3, "steps": steps if prec == "f16" else parity_steps,
                    "planted": workload.planted_stats(lists3, wl.targets.tolist()), "vs_fp32_mode": st3,
                    "index_exact_vs_fp32_mode": st3["corresp_equal"] == st3["slots_compared"] and st3["templates_equal"] == st3["detections"]}
        ex3 = None
        gc.collect()
        torch.cuda.empty_cache()
    del bank, wl, inp, sd
    gc.collect()
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    main()
