"""How much do kernels of different HIP streams overlap in a rocprofv3 --kernel-trace database (rocpd)?
    python tools/trace_overlap.py <db>  ->  per queue: launches, busy ms; union of all intervals, sum of all intervals, ms with >= 2 kernels in flight"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute(f"select start, end, {qcol or '0'}, name from kernels order by start").fetchall()
print("columns:", cols)
per_q = {}
for s, e, q, _ in rows:
    n, t = per_q.get(q, (0, 0))
    per_q[q] = (n + 1, t + (e - s))
for q, (n, t) in sorted(per_q.items(), key=lambda kv: -kv[1][1]):
    print(f"queue {q}: {n} launches, {t / 1e6:.2f} ms of kernel time")
ev = sorted([(s, 1) for s, _, _, _ in rows] + [(e, -1) for _, e, _, _ in rows])
depth, last, union, multi = 0, None, 0, 0
for t, d in ev:
    if last is not None and depth > 0:
        union += t - last
        if depth > 1:
            multi += t - last
    depth += d
    last = t
tot = sum(e - s for s, e, _, _ in rows)
print(f"sum of kernel intervals {tot / 1e6:.2f} ms, union {union / 1e6:.2f} ms, with >= 2 kernels in flight {multi / 1e6:.2f} ms, span {(rows[-1][1] - rows[0][0]) / 1e6:.2f} ms")
