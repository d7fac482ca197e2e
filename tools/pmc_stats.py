"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database (view `counters_collection`)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "%"
rows = db.execute(
    "select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection where kernel_name like ? "
    "group by kernel_name, counter_name order by kernel_name, counter_name", (pat,)).fetchall()
last = None
for k, c, n, avg, tot in rows:
    if k != last:
        print(k[:140])
        last = k
    print(f"    {c:34s} dispatches {n:5d}  avg {avg:16.1f}")
