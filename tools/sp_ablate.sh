#!/bin/bash
# VERDICT r4 item 4, the cheap bound: what would the f16x3 GEMMs gain if the two cross terms (hi*lo, lo*hi) ran on the half-cost fp8 pipe?  A 64-k fp8
# MFMA costs two 16-k fp16 MFMAs, so "1 fp16 + 2 half-cost" has the MFMA time of TWO fp16 MFMAs per product with the SAME operand bytes -- exactly what the
# FP_SP_ABLATE=1 build (one cross term dropped: wrong results) executes.  FP_SP_ABLATE=2 drops both: the floor the operand traffic alone sets.
#   gpurun -- bash tools/sp_ablate.sh      (after tools/build_variant.sh sp_ablate1 gemm_split.hip -DFP_SP_ABLATE=1; ... sp_ablate2 gemm_split.hip ... =2 -- the split-fp16 instantiations live in translation unit 3, gemm_split.hip)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in libfoundpose_amd sp_ablate1 sp_ablate2; do
    echo "== $v"; FOUNDPOSE_AMD_LIB=$PWD/foundpose_amd/lib/$v.so python tools/bench_kernels.py gemmsplit 2>/dev/null | grep -v "^$"
  done
done
for v in libfoundpose_amd sp_ablate1 sp_ablate2; do
  echo "== pipeline f16x3, $v"; FOUNDPOSE_AMD_LIB=$PWD/foundpose_amd/lib/$v.so python bench.py --precision f16x3 --parity-precision none --skip-probes --steps 5 --warmup 2 2>/dev/null | tail -1 | cut -c1-200
done
