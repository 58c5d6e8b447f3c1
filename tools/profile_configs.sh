#!/bin/bash
# rocprofv3 evidence for the two non-headline single-GPU configurations: per-step kernel table + matrix-pipe counters each.
#   gpurun -- bash tools/profile_configs.sh <tag>     -> gpurun_out/<tag>_config3_*, gpurun_out/<tag>_config5_share_*
tag=${1:-r6}
cd $GRAFT_REPO_ROOT
C3="--objects 8 --templates 800 --batch 256 --parity-precision none --no-parity --no-mode-f16"
C5="--version vitg14-reg --layer 39 --precision fp8 --templates 50000 --batch 128 --parity-precision none --no-parity --no-mode-f16"
bash tools/profile_bench.sh ${tag}_config3 $C3
bash tools/profile_bench.sh ${tag}_config5_share $C5
bash tools/pmc_mfma.sh ${tag}_config3 $C3
bash tools/pmc_mfma.sh ${tag}_config5_share $C5
for c in config3 config5_share; do echo "== $c"; head -8 gpurun_out/${tag}_${c}_per_step_kernels.csv | cut -c1-160; head -12 gpurun_out/${tag}_${c}_pmc_mfma.txt | cut -c1-200; done
