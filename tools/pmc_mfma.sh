#!/bin/bash
# Matrix-pipe utilisation of the kernels of the BENCHMARKED command: one rocprofv3 --pmc pass (SQ + GRBM counters only, no trace domains)
# over `python bench.py --skip-probes [args]`; util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8
# (the counter is reported summed over the 8 XCDs: 4.38 M for a 263-us launch = 8 x 548 k cycles at 2.08 GHz).
#   bash tools/pmc_mfma.sh <tag> [extra bench.py args]        (on the GPU box; writes gpurun_out/<tag>_pmc_mfma.txt)
tag=${1:-pmc}; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${tag}_pmc_mfma.txt
cd /tmp && export TMPDIR=/tmp
d=/tmp/pmcm_$tag
rm -rf $d
timeout 1500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $d -o r -- python $R/bench.py --steps 3 --warmup 1 --skip-probes "$@" > /dev/null 2>&1
db=$(find $d -name "*.db" 2>/dev/null | head -1)
python - "$db" "$*" > $out <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
k = {}
for name, ctr, val, n in rows:
    k.setdefault(name, {})[ctr] = (val, n)
print(f"# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE over: python bench.py --steps 3 --warmup 1 --skip-probes {sys.argv[2]}")
print("# per launch averages; cycles = GRBM_GUI_ACTIVE / 8 XCDs; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles); VALU/WAVE = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (both in quad-cycles)")
print(f"{'kernel':72s} {'launches':>8s} {'cycles':>12s} {'MFMA_BUSY':>14s} {'mfma_util':>9s} {'VALU/WAVE':>9s}")
for name, c in sorted(k.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', (0, 0))[0] * kv[1].get('GRBM_GUI_ACTIVE', (0, 1))[1]):
    gui, n = c.get('GRBM_GUI_ACTIVE', (0, 0))
    mf = c.get('SQ_VALU_MFMA_BUSY_CYCLES', (0, 0))[0]
    wc = c.get('SQ_WAVE_CYCLES', (0, 0))[0]
    va = c.get('SQ_ACTIVE_INST_VALU', (0, 0))[0]
    if gui * n < 1e6:
        continue
    cyc = gui / 8.0
    print(f"{name[:72]:72s} {n:8d} {cyc:12.0f} {mf:14.0f} {mf / (1024 * cyc) if cyc else 0:9.3f} {va / wc if wc else 0:9.3f}")
PY
cat $out
