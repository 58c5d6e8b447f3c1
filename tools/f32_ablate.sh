#!/bin/bash
# Ablations of f32_tile_kernel at the matching shapes of one bench step (PCA, word 3-NN, cyclic tiles):
#   bash tools/f32_ablate.sh build   (build container)  -> foundpose_amd/lib/f32_{noepi,nomfma,nostage}.so
#   bash tools/f32_ablate.sh run     (GPU box)          -> kernel durations per variant (rocprofv3 kernel trace)
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  bash tools/build_variant.sh f32_noepi f32_tile.hip -DFP_F32_NO_EPI
  bash tools/build_variant.sh f32_nomfma f32_tile.hip -DFP_F32_NO_MFMA
  bash tools/build_variant.sh f32_nostage f32_tile.hip -DFP_F32_NO_STAGE
  bash tools/build_variant.sh f32_mfma_only f32_tile.hip -DFP_F32_NO_STAGE -DFP_F32_NO_EPI
  exit 0
fi
R=$PWD
cp foundpose_amd/lib/libfoundpose_amd.so /tmp/base.so
for v in ${VARIANTS:-base f32_noepi f32_nomfma f32_nostage f32_mfma_only}; do
  if [ $v = base ]; then cp /tmp/base.so foundpose_amd/lib/libfoundpose_amd.so; else cp foundpose_amd/lib/$v.so foundpose_amd/lib/libfoundpose_amd.so; fi
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p_$v && rocprofv3 --kernel-trace -d /tmp/p_$v -o r -- python $R/tools/bench_kernels.py match > /tmp/out_$v.txt 2>&1)
  [ $v = base ] && cat /tmp/out_$v.txt | grep -v amdgpu.ids
  python - "$v" $(find /tmp/p_$v -name "*.db" | head -1) <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[2])
for r in db.execute("select name, grid_x, grid_y, grid_z, count(*), avg(end-start), min(end-start) from kernels where name like '%f32_tile%' or name like '%cyclic_select%' or name like '%knn_merge%' group by name, grid_x, grid_y, grid_z"):
    nm = r[0].split("::")[-1][:32]
    print(f"{sys.argv[1]:14s} {nm:32s} grid {r[1]:6d} x {r[2]:4d} x {r[3]:4d}: {r[4]:4d} launches  avg {r[5]/1e3:7.1f} us  min {r[6]/1e3:7.1f} us")
PY
done
cp /tmp/base.so foundpose_amd/lib/libfoundpose_amd.so
