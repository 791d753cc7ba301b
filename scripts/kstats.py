"""Per-step kernel time table from a rocprofv3 rocpd database:  python scripts/kstats.py <results.db> <steps incl. warmup>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2])
rows = db.execute("select name, count(*), sum(end-start) from kernels group by name order by 3 desc").fetchall()
tot = 0.0
for n, c, t in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{t / steps / 1000:8.1f} us/step  calls/step {c / steps:5.2f}  avg {t / c / 1000:8.1f} us  {n[:80]}")
for n, c, t in rows:
    tot += t / steps / 1000
print(f"{tot:8.1f} us/step total")
