"""Summarise a rocprofv3 rocpd database (--kernel-trace) into a per-kernel stats table (markdown/CSV-ish text)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute(
    "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(accum_vgpr_count),"
    " max(sgpr_count), max(lds_size), max(scratch_size) from kernels group by name order by sum(duration) desc"
).fetchall()
tot = sum(r[2] for r in rows) or 1
print(f"{'kernel':110s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s} vgpr agpr sgpr  lds scratch")
for r in rows:
    name = r[0] if len(r[0]) <= 110 else r[0][:107] + "..."
    print(f"{name:110s} {r[1]:6d} {r[2] / 1e6:10.3f} {r[3] / 1e3:10.2f} {r[4] / 1e3:9.2f} {r[5] / 1e3:9.2f} {100 * r[2] / tot:6.2f} {r[6]:4d} {r[7]:4d} {r[8]:4d} {r[9]:5d} {r[10]:4d}")
print(f"total kernel time {tot / 1e6:.3f} ms")
