"""Print per-kernel counter averages from a rocprofv3 --pmc rocpd .db, one line per (kernel, launch grid): launches of one kernel
template with different grids are different problems (proj vs fc2 of the residual GEMM; the retrieval kernel at 10 000 and at
50 000 templates) and their traffic must not be averaged together.
    python tools/pmc_summary.py <results.db> [name filter]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
filt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select kernel_name, grid_size_x, grid_size_y, grid_size_z, counter_name, avg(value), count(*) from counters_collection "
                  "group by kernel_name, grid_size_x, grid_size_y, grid_size_z, counter_name order by kernel_name, grid_size").fetchall()
for r in rows:
    if filt in r[0]:
        print(f"{r[0][:70]:70s} grid=({r[1]},{r[2]},{r[3]})".ljust(96) + f" {r[4]:12s} {r[5]:16.1f} n={r[6]}")
