#!/bin/bash
# Ablations of cosine_fused_kernel (run in the build container, then on the GPU box):
#   bash tools/cos_ablate.sh build      -> foundpose_amd/lib/cos_{nomfma,noreduce,neither}.so
#   bash tools/cos_ablate.sh run        -> kernel durations of each variant (rocprofv3 kernel trace), on the GPU box
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  bash tools/build_variant.sh cos_nomfma match.hip -DFP_COS_NO_MFMA
  bash tools/build_variant.sh cos_noreduce match.hip -DFP_COS_NO_REDUCE
  bash tools/build_variant.sh cos_neither match.hip -DFP_COS_NO_MFMA -DFP_COS_NO_REDUCE
  bash tools/build_variant.sh cos_slots2 match.hip -DFP_COS_SLOTS=2
  bash tools/build_variant.sh cos_nodma match.hip -DFP_COS_NO_DMA
  bash tools/build_variant.sh cos_nodma_noreduce match.hip -DFP_COS_NO_DMA -DFP_COS_NO_REDUCE
  bash tools/build_variant.sh cos_slots4 match.hip -DFP_COS_SLOTS=4
  exit 0
fi
R=$PWD
cp foundpose_amd/lib/libfoundpose_amd.so /tmp/base.so
for v in ${VARIANTS:-base cos_nomfma cos_noreduce cos_neither cos_slots2 cos_slots4}; do
  if [ $v = base ]; then cp /tmp/base.so foundpose_amd/lib/libfoundpose_amd.so; else cp foundpose_amd/lib/$v.so foundpose_amd/lib/libfoundpose_amd.so; fi
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p_$v && rocprofv3 --kernel-trace -d /tmp/p_$v -o r -- python $R/tools/bench_kernels.py cos > /dev/null 2>&1)
  python - "$v" $(find /tmp/p_$v -name "*.db" | head -1) <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[2])
for r in db.execute("select name, grid_x, grid_z, count(*), avg(end-start), min(end-start) from kernels where name like '%cosine_fused%' or name like '%topn_rows_strict%' or name like '%cand_merge%' group by name, grid_x, grid_z"):
    print(f"{sys.argv[1]:14s} grid {r[1]:7d} x{r[2]}: {r[3]:4d} launches  avg {r[4]/1e3:7.1f} us  min {r[5]/1e3:7.1f} us")
PY
done
cp /tmp/base.so foundpose_amd/lib/libfoundpose_amd.so
