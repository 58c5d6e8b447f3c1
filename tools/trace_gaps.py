#!/usr/bin/env python
"""Idle time between kernels in the steady-state steps of a rocprofv3 --kernel-trace run of bench.py:
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -- python bench.py --skip-probes --steps 6 --warmup 3
    python tools/trace_gaps.py <dir>/.../*_kernel_trace.csv [steps]
A step starts at its query_flags_kernel (fp_query_select); the last `steps` (default 3) whole steps are analysed."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
starts = [i for i, (_, _, n) in enumerate(ev) if "query_flags_kernel" in n]
seg = ev[starts[-nsteps - 1]:starts[-1]]
gaps, end, prev = [], seg[0][1], seg[0][2]
for s, e, n in seg[1:]:
    if s > end:
        gaps.append(((s - end) / 1e3, prev, n))
    if e > end:
        end, prev = e, n
span = (seg[-1][1] - seg[0][0]) / 1e3
print(f"{nsteps} steps: {span / nsteps / 1e3:.3f} ms per step, {len(seg) / nsteps:.0f} launches per step, "
      f"idle {sum(g[0] for g in gaps) / nsteps:.1f} us per step in {len(gaps) / nsteps:.1f} gaps per step")
agg = defaultdict(lambda: [0, 0.0])
short = lambda n: n.split("(")[0][-48:]
for g, a, b in gaps:
    agg[(short(a), short(b))][0] += 1
    agg[(short(a), short(b))][1] += g
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:20]:
    print(f"{v[1] / nsteps:8.1f} us per step  x{v[0] / nsteps:.1f}  after {k[0] or '?'}  before {k[1] or '?'}")
