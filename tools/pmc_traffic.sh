#!/bin/bash
# HBM-side traffic of the fc1 GEMM and of the attention kernel: FETCH_SIZE / WRITE_SIZE in SEPARATE rocprofv3 --pmc passes
# (guide MI355X_MICROARCH.md, HBM section), each under its own timeout.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for what in "fc1 256" "attn"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/pmc_$(echo $what | tr ' ' '_')_$ctr
    timeout 150 rocprofv3 --pmc $ctr -d $d -o r -- python $R/tools/pmc_gemm.py $what > /dev/null 2>&1 || echo "pass $what $ctr failed/timeout"
    db=$(find $d -name "*.db" 2>/dev/null | head -1)
    [ -n "$db" ] && python $R/tools/pmc_summary.py $db _bf16
  done
done
