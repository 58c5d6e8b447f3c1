#!/usr/bin/env python
"""Precision-schedule sweep (VERDICT r5 #1b): blocks 0..k-1 in the f16 mode, blocks k..18 in f16f8 / f16x3, at BASELINE config 2 (ViT-L/14-reg layer 18,
518^2, batch 32, 10 000 templates).  Per schedule: detections/s (bench.py's timed step) and correspondence slots equal to the library's fp32 mode on the
planted and on the margin-free (hard) workload.  Kill criterion of the review: a schedule >= 700 detections/s with 160/160 planted and >= 158/160 hard.
    python tools/schedule_sweep.py [--ks 0,3,6,9,12,15] [--tails f16f8,f16x3] > gpurun_out/schedule_sweep.txt"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from foundpose_amd import engine as fe, feature_util, workload
from foundpose_amd.bank import DeviceBank

ap = argparse.ArgumentParser()
ap.add_argument("--ks", default="0,3,6,9,12,15")
ap.add_argument("--tails", default="f16f8,f16x3")
ap.add_argument("--rev-ks", default="2,4,6,9,12,15", help="reversed schedules: near-exact blocks 0..k-1, f16 behind them")
ap.add_argument("--steps", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda", 0)
name = "dinov2_version=vitl14-reg_stride=14_facet=token_layer=18_norm=1"
B, S = 32, 518
ex32 = feature_util.make_feature_extractor(name, random_init_seed=1234, precision="fp32").to(dev)
wls = {"planted": workload.build_planted_workload(ex32, B, S, 1, 10000, 256, 2048, seed=7, crop_seed=0, words_per_texture=workload.WORDS_PER_TEXTURE),
       "hard": workload.build_planted_workload(ex32, B, S, 1, 10000, 256, 2048, seed=7, crop_seed=0, words_per_texture=workload.WORDS_PER_TEXTURE, hard=True)}
banks = {k: DeviceBank(w.repres, device=dev) for k, w in wls.items()}
ref = {}
for k, w in wls.items():
    r = fe.FoundPoseEngine(ex32, banks[k], 14.0, 5, 300, tie_order="torch").infer_batch(w.crops, w.masks, w.det_obj)
    ref[k] = [r.corresp_list(b) for b in range(B)]
del ex32
torch.cuda.empty_cache()

def run(ex):
    out = {}
    for k, w in wls.items():
        eng = fe.FoundPoseEngine(ex, banks[k], 14.0, 5, 300, tie_order="torch")
        r = eng.infer_batch(w.crops, w.masks, w.det_obj)
        st = workload.parity_stats([r.corresp_list(b) for b in range(B)], ref[k])
        out[k] = (st["corresp_equal"], st["slots_compared"], st["templates_equal"])
    eng = fe.FoundPoseEngine(ex, banks["planted"], 14.0, 5, 300, tie_order="torch")
    w = wls["planted"]
    for _ in range(2):
        fe.pack_result(eng.infer_batch(w.crops, w.masks, w.det_obj))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps):
        fe.pack_result(eng.infer_batch(w.crops, w.masks, w.det_obj))
    torch.cuda.synchronize()
    return B * args.steps / (time.perf_counter() - t0), out

print("schedule                 det/s   planted slots  hard slots   (templates equal planted / hard)")
for prec in ("bf16", "f16"):
    ex = feature_util.make_feature_extractor(name, random_init_seed=1234, precision=prec).to(dev)
    v, o = run(ex)
    print(f"{prec:22s} {v:8.1f}   {o['planted'][0]:3d}/{o['planted'][1]}        {o['hard'][0]:3d}/{o['hard'][1]}      {o['planted'][2]} / {o['hard'][2]}", flush=True)
    del ex; torch.cuda.empty_cache()
ap_rev = [int(v) for v in args.rev_ks.split(",") if v]
for head in args.tails.split(","):   # the other way round: near-exact blocks FIRST, the f16 pipeline behind them
    for k in ap_rev:
        ex = feature_util.make_feature_extractor(name, random_init_seed=1234, precision="f16", head_blocks=k, head_precision=head).to(dev)
        v, o = run(ex)
        print(f"{head:6s}[0:{k:2d}] + f16[{k:2d}:19] {v:8.1f}   {o['planted'][0]:3d}/{o['planted'][1]}        {o['hard'][0]:3d}/{o['hard'][1]}      {o['planted'][2]} / {o['hard'][2]}", flush=True)
        del ex; torch.cuda.empty_cache()
for tail in args.tails.split(","):
    for k in [int(v) for v in args.ks.split(",") if v]:
        ex = feature_util.make_feature_extractor(name, random_init_seed=1234, precision=tail, head_blocks=k).to(dev)
        v, o = run(ex)
        print(f"f16[0:{k:2d}] + {tail:6s}[{k:2d}:19] {v:8.1f}   {o['planted'][0]:3d}/{o['planted'][1]}        {o['hard'][0]:3d}/{o['hard'][1]}      {o['planted'][2]} / {o['hard'][2]}", flush=True)
        del ex; torch.cuda.empty_cache()
