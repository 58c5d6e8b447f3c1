#!/bin/bash
# HBM-side traffic of the kernels of the BENCHMARKED command: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes
# over `python bench.py` itself (short step count; guide MI355X_MICROARCH.md, HBM section).  Per-kernel averages go to
# gpurun_out/<tag>_pmc_traffic.txt; bench.py's PMC_TRAFFIC quotes them with this file's name and the commit.
#   bash tools/pmc_bench.sh <tag>          (on the GPU box)
tag=${1:-pmc}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${tag}_pmc_traffic.txt
cd /tmp && export TMPDIR=/tmp
echo "# rocprofv3 --pmc over: python bench.py --steps 3 --warmup 1 --skip-probes (averages per launch AND launch grid, KiB as reported)" > $out
echo "# FETCH_SIZE on gfx950 counts 64 B per 128-B request of wide coalesced streams: double it (guide, HBM section); WRITE_SIZE as reported" >> $out
for ctr in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pmcb_$ctr
  rm -rf $d
  timeout 400 rocprofv3 --pmc $ctr -d $d -o r -- python $R/bench.py --steps 3 --warmup 1 --skip-probes > /dev/null 2>&1 || echo "pass $ctr failed/timeout" >> $out
  db=$(find $d -name "*.db" 2>/dev/null | head -1)
  [ -n "$db" ] && python $R/tools/pmc_summary.py $db "" | grep -i "gemm_bf16\|attn_bf16\|cosine_\|cand_merge\|f32_tile\|layernorm\|ln_sample\|topn_rows\|cyclic" >> $out
done
cat $out
