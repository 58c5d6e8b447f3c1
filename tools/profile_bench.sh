#!/bin/bash
# rocprofv3 kernel trace of bench.py at two step counts -> per-step kernel table (tools/rocprof_delta.py)
#   bash tools/profile_bench.sh <tag> [extra bench.py args]      (on the GPU box; writes gpurun_out/<tag>_*)
tag=${1:-prof}; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_small /tmp/p_big   # (a second call on the same box must not find the first call's traces)
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/p_small -o r -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --other-configs none --no-hard --no-latency --no-parity-fast --no-mode-f16 "$@" > /tmp/small.json 2>/dev/null
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/p_big -o r -- python $R/bench.py --steps 24 --warmup 2 --no-cpu-baseline --other-configs none --no-hard --no-latency --no-parity-fast --no-mode-f16 "$@" > /tmp/big.json 2>/dev/null
python $R/tools/rocprof_delta.py $(find /tmp/p_small -name "*.db" | head -1) $(find /tmp/p_big -name "*.db" | head -1) 20 $R/gpurun_out/${tag}_per_step_kernels.csv
python $R/tools/rocprof_summary.py $(find /tmp/p_big -name "*.db" | head -1) $R/gpurun_out/${tag}_kernel_stats.csv > /dev/null 2>&1 || true
tail -1 /tmp/big.json > $R/gpurun_out/${tag}_profiled_bench.json
