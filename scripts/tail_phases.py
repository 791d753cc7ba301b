"""Where the step's tail launch (csrc/enc_tail.hip: scatter-add workgroups that go on with the weight-gradient jobs) spends its time, from the
per-workgroup shader-clock stamps of the diagnostic build (make -C recboard_amd/csrc encprof).    python scripts/tail_phases.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from recboard_amd import lib  # noqa: E402
lib.LIB_PATH = os.path.join(ROOT, "recboard_amd", "librecengine_encprof.so")
L = lib.load()
import bench  # noqa: E402
from recboard_amd.sasrec import SASRecEngine  # noqa: E402
cfg = dict(bench.BEAUTY, B=int(os.environ.get("TAIL_B", "512")))
LARGE = len(sys.argv) > 1 and sys.argv[1] == "large"      # the sparse tail of a large-table step at D = 128 (config 5's, on a 4 M-row table)
if LARGE:
    from recboard_amd.large import SASRecLargeTableEngine
    cfg = dict(cfg, items=4_000_000, D=128)
bs = [tuple(torch.from_numpy(x).cuda() for x in b) for b in bench.synth_batches(cfg, 8, 1)]
m = (SASRecLargeTableEngine(cfg["items"], 50, 128, 2, dropout_rate=0.5, lr=5e-4, weight_decay=1e-6, seed=1, table_init="counter") if LARGE else
     SASRecEngine(cfg["items"], 50, 64, 2, dropout_rate=0.5, lr=5e-4, weight_decay=1e-6, seed=1))
m.prep_in_tail = True
for i in range(40):
    m.train_step_graph(*bs[i % 8], next_batch=bs[(i + 1) % 8])
torch.cuda.synchronize()
NWG = 256
NM = 32
buf = (ctypes.c_ulonglong * (NM * NWG))()
L.re_dbg_tail_marks.argtypes, L.re_dbg_tail_marks.restype = [ctypes.c_void_p, ctypes.c_int], ctypes.c_int
assert L.re_dbg_tail_marks(buf, NWG) == 0
t = np.array(list(buf), dtype=np.int64).reshape(NWG, NM)
# (the shader clocks of different XCDs are not aligned: everything below is relative to the workgroup's own start)
sc = (t[:, 1] - t[:, 0]) / 1e3
tot = (t[:, 15] - t[:, 0]) / 1e3
njobs = (t[:, 2:8:2] > 0).sum(1)
print(f"kilo-cycles of the shader clock, per workgroup, relative to its own start ({NWG} workgroups)")
print("scatter-add part   min/med/p90/max", sc.min(), np.median(sc), np.percentile(sc, 90), sc.max(), " (slowest: workgroup", int(sc.argmax()), ")")
print("tickets taken per workgroup: histogram", np.bincount(njobs).tolist())
durs, tk = [], []
for w in range(NWG):
    prev = t[w, 1]
    for i in range(njobs[w]):
        e = t[w, 3 + 2 * i]
        if e > 0:
            durs.append((e - prev) / 1e3); tk.append(int(t[w, 2 + 2 * i]) - 1)
            prev = e
d, tk = np.array(durs), np.array(tk)
print("ticket duration    min/med/p90/max", d.min(), np.median(d), np.percentile(d, 90), d.max(), " tickets", len(d))
# queue order (enc_tail.hip: tail_jobs): the next batch's plan job, the matrix tickets, the next batch's element-wise jobs, the position tickets
n_mat_t = 240 if (cfg["B"] >= 1024 and not LARGE) else 144      # (enc_wgrad_job.h: wg_nsplit_tail -- 40 row splits from 1 024 sequences on)
n_pos_t = 72
n_prep_t = max(0, int(tk.max()) + 1 - n_mat_t - n_pos_t)
n_plan_t = 1 if n_prep_t else 0
for lo, hi, name in ((0, n_plan_t, "plan ticket"), (n_plan_t, n_plan_t + n_mat_t, "matrix tickets"), (n_plan_t + n_mat_t, n_prep_t + n_mat_t, "element-wise tickets"),
                     (n_prep_t + n_mat_t, 10 ** 9, "position tickets")):
    sel = (tk >= lo) & (tk < hi)
    if sel.any():
        print(f"  {name}: n {int(sel.sum())} med {np.median(d[sel]):.1f} max {d[sel].max():.1f}")
print("workgroup lifetime min/med/p90/max", tot.min(), np.median(tot), np.percentile(tot, 90), tot.max())
pw = [w for w in range(NWG) if (njobs[w] > 0 and t[w, 2] == 1) or w == NWG - 1]    # (ticket 0: the plan's rest; the last workgroup: its spans)
for w in pw:
    st = [int(x) for x in t[w, 8:14]]
    base = t[w, 1]
    print("plan job in workgroup", w, ": phase stamps relative to its scatter end (kcycles):", [round((x - base) / 1e3, 1) for x in st if x > 0])
for w in np.argsort(tot)[-6:]:
    print("  wg", int(w), "scatter", sc[w], "tickets", int(njobs[w]), [int(t[w, 2 + 2 * i]) - 1 for i in range(njobs[w])], "lifetime", tot[w])

# stamps inside the matrix jobs (enc_wgrad_job.h WG_STAMP: 0 entry, 1 first fetch issued, 2 first stage in LDS, 3.. end of each stage's products)
rows = []
for w in range(NWG):
    if w in pw or njobs[w] != 1 or t[w, 8] == 0 or t[w, 10] == 0:
        continue
    rows.append([(t[w, 8 + i] - t[w, 1]) / 1e3 if t[w, 8 + i] > 0 else np.nan for i in range(6)] + [(t[w, 3] - t[w, 1]) / 1e3])
if rows:
    r = np.array(rows)
    print("matrix job stamps relative to the scatter end, medians (kcycles): entry, fetch issued, stage 0 staged, products 0, 1, 2 | job end", np.round(np.nanmedian(r, 0), 1).tolist(), " n", len(rows))

rel = lambda i: np.round(np.median([(t[w, i] - t[w, 0]) / 1e3 for w in range(NWG) if t[w, i] > 0 and w not in pw] or [np.nan]), 1)
print("medians relative to the workgroup's start (kcycles): scatter start", rel(20), "key count known", rel(21), "first chunk ranked", rel(25), "placed", rel(26), "keys scanned", rel(22), "rows added", rel(23),
      "rows stored", rel(24), "| scatter end", rel(1), "| at first barrier", rel(19), "passed", rel(16), "ticket read", rel(17), "job called", rel(18), "job entry", rel(8), "| end", rel(15))
for w in np.argsort(sc)[-3:]:      # the owners of the hottest rows: scatter_owner.h stamps (0 start, 1 count known, 5 first chunk, 6 lists placed, 2 scanned, 3 added, 4 stored)
    print("  slow scatter, workgroup", int(w), {k: round(float((t[w, i] - t[w, 0]) / 1e3), 1) for k, i in
                                                 (("start", 20), ("count", 21), ("chunk0", 25), ("placed", 26), ("scanned", 22), ("added", 23), ("stored", 24)) if t[w, i] > 0})
