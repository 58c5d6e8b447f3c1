#!/bin/bash
# Where the bf16 attention kernel's time goes: variant libraries of attn.hip without (a) the K/V LDS-DMA stream, (b) also the per-tile
# wait + barrier, (c) also the LDS fragment reads (= the instruction stream alone, cf. tools/ubench/attn_mix.hip), timed in one process
# on one box.  Build here (cross-compiles), run on the GPU box:  bash tools/attn_ablate.sh build ; gpurun -- bash tools/attn_ablate.sh run
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  bash tools/build_variant.sh attn_nodma attn.hip -DFP_ATTN_NO_DMA
  bash tools/build_variant.sh attn_nodma_nobar attn.hip -DFP_ATTN_NO_DMA -DFP_ATTN_NO_BARRIER
  bash tools/build_variant.sh attn_nomem attn.hip -DFP_ATTN_NO_DMA -DFP_ATTN_NO_BARRIER -DFP_ATTN_NO_LDS
  exit 0
fi
cp foundpose_amd/lib/libfoundpose_amd.so /tmp/lib_orig.so
for v in orig attn_nodma attn_nodma_nobar attn_nomem orig; do
  if [ $v = orig ]; then cp /tmp/lib_orig.so foundpose_amd/lib/libfoundpose_amd.so; else cp foundpose_amd/lib/$v.so foundpose_amd/lib/libfoundpose_amd.so; fi
  echo "== $v"; python tools/bench_kernels.py attn 2>&1 | grep "variant=0\|variant=2"
done
cp /tmp/lib_orig.so foundpose_amd/lib/libfoundpose_amd.so
