"""Do consecutive kernels of a stream overlap in time?  From a rocprofv3 rocpd database: for every kernel, how far its START lies before the END
of the kernel dispatched in front of it (same queue).    python scripts/overlap_check.py <results.db> [name filter]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
qcol = "queue_id" if "queue_id" in cols else None
rows = db.execute(f"select name, start, end{', ' + qcol if qcol else ''} from kernels order by start").fetchall()
prev = {}
stats = {}
for r in rows:
    name, st, en = r[0], r[1], r[2]
    q = r[3] if qcol else 0
    if q in prev:
        pname, pen = prev[q]
        key = (pname[:48], name[:48])
        ov = pen - st
        s = stats.setdefault(key, [0, 0, 0])
        s[0] += 1
        if ov > 0:
            s[1] += 1
            s[2] = max(s[2], ov)
    prev[q] = (name, en)
for (a, b), (n, k, mx) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    if flt in a or flt in b:
        print(f"{k:6d} of {n:6d} overlap (max {mx / 1000:7.1f} us): {b}  started before the end of  {a}")
