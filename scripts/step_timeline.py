"""Timeline of ONE steady-state captured step from a rocprofv3 kernel trace (rocpd database): start offset, duration, queue, and the idle
gap in front of every kernel.   python scripts/step_timeline.py <results.db> [which step from the end, default 20]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = db.execute(f"select name, start, end, {q} from kernels order by start").fetchall()
import os
key = os.environ.get("TL_KEY", "sasrec_batch_prep_k")       # the kernel a step starts with (sasrec_step_stage_k: the pipelined step)
prep = [i for i, r in enumerate(rows) if r[0].startswith(key)]
i0, i1 = prep[-back - 1], prep[-back]
t0 = rows[i0][1]
last_end = None
for name, st, en, qu in rows[i0:i1 + 1]:
    gap = "" if last_end is None else f"{(st - last_end) / 1e3:7.2f}"
    print(f"{(st - t0) / 1e3:8.2f} us  +{(en - st) / 1e3:7.2f} us  gap {gap:>8s}  q{qu}  {name[:60]}")
    last_end = max(last_end or en, en)
print(f"step: {(rows[i1][1] - t0) / 1e3:.2f} us")
