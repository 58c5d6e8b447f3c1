"""Print per-kernel counter averages from a rocprofv3 --pmc rocpd .db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
filt = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [d[0] for d in db.execute("select * from counters_collection limit 1").description]
rows = db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
for r in rows:
    if filt in r[0]:
        print(f"{r[0][:70]:70s} {r[1]:32s} {r[2]:16.1f} n={r[3]}")
