#!/bin/bash
# The non-headline BASELINE configurations on one box: config 3 (8 objects x 800 templates, batch 256), one GPU's share of config 5
# (ViT-g/14 fp8, 50 000 templates, batch 128) and the full-mask worst case.   gpurun -- bash tools/other_configs.sh <tag>
tag=${1:-rX}
cd $GRAFT_REPO_ROOT
python bench.py --objects 8 --templates 800 --batch 256 --steps 5 --warmup 2 --cpu-detections 1 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_config3.json
python bench.py --version vitg14-reg --layer 39 --precision fp8 --templates 50000 --batch 128 --steps 5 --warmup 2 --cpu-detections 1 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_config5_share.json
python bench.py --mask full --cpu-detections 1 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_full_mask.json
for f in config3 config5_share full_mask; do python -c "
import json; d=json.load(open('gpurun_out/${tag}_bench_$f.json')); p=d.get('parity_mode') or {}
print('$f', d['value'], d['ms_per_step'], 'parity_mode', p.get('value'), p.get('index_exact_vs_fp32_mode'), (p.get('vs_fp32_mode') or {}).get('corresp_equal'), (p.get('vs_fp32_mode') or {}).get('slots_compared'))"; done
# the reference's default backbone family (no register tokens) at the headline shape, and a batch whose GEMM tile counts are whole rounds
python bench.py --version vitl14 --cpu-detections 1 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_vitl14_noreg.json
python bench.py --batch 35 --no-cpu-baseline --parity-precision none 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_batch35.json
for f in vitl14_noreg batch35; do python -c "
import json; d=json.load(open('gpurun_out/${tag}_bench_$f.json')); print('$f', d['value'], d['ms_per_step'], d['roofline_vit_end_to_end']['frac'])"; done
