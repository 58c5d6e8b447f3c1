#!/bin/bash
# Where the role-split split-fp16 attention kernel's time goes: variant libraries of attn.hip, timed in one process order on one box.
#   bash tools/attn_split_ablate.sh build ; gpurun -- bash tools/attn_split_ablate.sh run
set -e
cd "$(dirname "$0")/.."
VARIANTS="spp_grp1 spp_nosoftmax spp_nomatrix spp_nomidbar"
if [ "$1" = build ]; then
  bash tools/build_variant.sh spp_grp1 attn.hip -DSPP_GRP=1
  bash tools/build_variant.sh spp_nosoftmax attn.hip -DSPP_NO_SOFTMAX
  bash tools/build_variant.sh spp_nomatrix attn.hip -DSPP_NO_MATRIX
  bash tools/build_variant.sh spp_nomidbar attn.hip -DSPP_NO_MIDBAR
  exit 0
fi
cp foundpose_amd/lib/libfoundpose_amd.so /tmp/lib_orig.so
for v in orig $VARIANTS orig; do
  if [ $v = orig ]; then cp /tmp/lib_orig.so foundpose_amd/lib/libfoundpose_amd.so; else cp foundpose_amd/lib/$v.so foundpose_amd/lib/libfoundpose_amd.so; fi
  echo "== $v"; python tools/bench_kernels.py attnsplit 2>&1 | grep "variant=" | tail -4
done
cp /tmp/lib_orig.so foundpose_amd/lib/libfoundpose_amd.so
