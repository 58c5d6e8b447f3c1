#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db) into a per-kernel stats CSV
(the same columns `--stats` prints: calls, total, average, min, max, percentage).

    python tools/rocprof_summary.py gpurun_out/prof/bench_results.db profiles/r1_bench_kernel_stats.csv [--after-warmup-frac 0.0]
"""

import csv
import sqlite3
import sys


def main():
    db_path, out_path = sys.argv[1], sys.argv[2]
    db = sqlite3.connect(db_path)
    rows = db.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for r in rows:
            w.writerow([r[0], r[1], int(r[2]), f"{r[3]:.1f}", int(r[4]), int(r[5]), f"{100.0 * r[2] / total:.2f}"])
    print(f"wrote {out_path}: {len(rows)} kernels, {total / 1e6:.2f} ms of kernel time")


if __name__ == "__main__":
    main()
