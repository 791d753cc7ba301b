"""The bench's evaluation state (LayerNorm-encoded queries, xavier-initialised item table, synthetic Beauty sequences as seen
lists) through both paths of score_topk: identical results, fallback count, timings."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from recboard_amd import lib, ops
lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), "librecengine_dbg.so")   # the re_dbg_* switches live in the diagnostic twin (make -C recboard_amd/csrc dbg)
from recboard_amd.sasrec import SASRecEngine
L = lib.load()
for n, a in (("re_dbg_score_x2", [ctypes.c_int]), ("re_dbg_score_x2_maxerr", [ctypes.c_int]), ("re_dbg_score_x2_stats", [ctypes.c_void_p, ctypes.c_int]),
             ("re_dbg_score_x2_info", [ctypes.c_void_p])):
    getattr(L, n).argtypes = a; getattr(L, n).restype = None
cfg = bench.BEAUTY
U, N, K = cfg["users"], cfg["items"], 50
torch.manual_seed(1)
model = SASRecEngine(cfg["items"], 50, 64, 2, dropout_rate=0.5, loss="BCE", lr=cfg["lr"], weight_decay=cfg["wd"])
if len(sys.argv) > 1:   # train like the bench does before its evaluation leg
    hb = bench.synth_batches(cfg, 8, seed=1)
    bt = [tuple(torch.from_numpy(x).cuda() for x in b[:3]) for b in hb]
    model.train()
    for i in range(int(sys.argv[1])):
        model.train_step(*bt[i % len(bt)])
model.eval()
eval_seq = torch.from_numpy(np.concatenate([b[0] for b in bench.synth_batches(cfg, (U + 511) // 512, 99)])[:U]).cuda()
seen = [np.unique(s[s > 0] - 1) for s in eval_seq.cpu().numpy()]
sp = np.zeros(U + 1, np.int64); sp[1:] = np.cumsum([len(x) for x in seen])
seen_ptr, seen_idx = torch.from_numpy(sp).cuda(), torch.from_numpy(np.concatenate(seen)).cuda()
with torch.no_grad():
    q = torch.cat([model.encode(eval_seq[i:i + 512])[0][:, -1, :] for i in range(0, U, 512)]).contiguous()
items = model.params["Item.embeddings.weight"].detach()[1:]
if os.environ.get("X2_DUMP"):   # hand the state to scripts/x2_cycles.py
    torch.save({"q": q.cpu(), "E": items.cpu().clone(), "sp": seen_ptr.cpu(), "si": seen_idx.cpu()}, os.environ["X2_DUMP"])
def t(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / it
L.re_dbg_score_x2(0)
v0, i0 = ops.score_topk(q, items, seen_ptr, seen_idx, K); t0 = t(lambda: ops.score_topk(q, items, seen_ptr, seen_idx, K))
L.re_dbg_score_x2(1); L.re_dbg_score_x2_maxerr(1)
out = (ctypes.c_uint32 * 2)(); torch.cuda.synchronize(); L.re_dbg_score_x2_stats(out, 1)
v1, i1 = ops.score_topk(q, items, seen_ptr, seen_idx, K); torch.cuda.synchronize()
L.re_dbg_score_x2_stats(out, 1); L.re_dbg_score_x2_maxerr(0)
info = (ctypes.c_float * 8)(); L.re_dbg_score_x2_info(info)
t1 = t(lambda: ops.score_topk(q, items, seen_ptr, seen_idx, K))
print(f"identical={torch.equal(i0, i1) and torch.equal(v0, v1)} flagged={out[0]} exact {t0:.3f} ms split {t1:.3f} ms; |q| {q.norm(dim=1).mean():.3f} |e| max {items.norm(dim=1).max():.4f}")
print("last flagged:", [round(float(x), 6) for x in info])
uq = torch.unique(q, dim=0).shape[0]
print("distinct query rows:", uq, "of", U, " score sd:", float((q[:64] @ items.T).std()))
# list-maintenance counters of the main kernel on this state (diagnostic mode 3)
for n, a in (("re_dbg_score_diag", [ctypes.c_int]), ("re_dbg_score_counters", [ctypes.c_void_p, ctypes.c_int])):
    getattr(L, n).argtypes = a; getattr(L, n).restype = None
buf = (ctypes.c_ulonglong * 4)()
L.re_dbg_score_diag(3); L.re_dbg_score_counters(buf, 1)
ops.score_topk(q, items, seen_ptr, seen_idx, K); torch.cuda.synchronize()
L.re_dbg_score_counters(buf, 1); L.re_dbg_score_diag(0)
nw = 512 * 4
print(f"split kernel on this state: drains/wave {buf[0]/nw:.1f}  rounds/wave {buf[1]/nw:.1f}  hits/lane {buf[2]/nw/64:.1f}")
