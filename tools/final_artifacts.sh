#!/bin/bash
# The round's judged artefacts from ONE checkout on ONE box: full GPU test log, the default bench line, the per-step rocprofv3 kernel
# table of the same command, HBM-side traffic and matrix-pipe utilisation counters.   gpurun -- bash tools/final_artifacts.sh <tag>
tag=${1:-rX}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/${tag}_pytest_gpu.log
python tools/f16_probe.py 2>&1 | grep "^1 " > gpurun_out/${tag}_f16_probe.txt
python bench.py < /dev/null 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_final.json
bash tools/profile_bench.sh ${tag}_final > /dev/null 2>&1
bash tools/profile_bench.sh ${tag}_f16 --precision f16 --parity-precision none > /dev/null 2>&1
bash tools/profile_bench.sh ${tag}_f16x3 --precision f16x3 --parity-precision none > /dev/null 2>&1
bash tools/profile_bench.sh ${tag}_f16f8 --precision f16f8 --parity-precision none > /dev/null 2>&1
bash tools/pmc_bench.sh ${tag} > /dev/null 2>&1
bash tools/pmc_mfma.sh ${tag}_bf16 > /dev/null 2>&1
bash tools/pmc_mfma.sh ${tag}_f16 --precision f16 > /dev/null 2>&1
bash tools/pmc_mfma.sh ${tag}_f16x3 --precision f16x3 > /dev/null 2>&1
tail -3 gpurun_out/${tag}_pytest_gpu.log
python -c "import json; d=json.load(open('gpurun_out/${tag}_bench_final.json')); print(d['value'], d['ms_per_step'], d['roofline']); print(d.get('parity_mode',{}).get('value'))"
head -8 gpurun_out/${tag}_final_per_step_kernels.csv
