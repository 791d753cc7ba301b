"""Step timings of the other BASELINE configs on one MI355X (parity-test configs, not the bench line):
  C1 MF-BPR   Beauty  (22 363 users, 12 101 items, B = 2048)
  C3 LightGCN Yelp2018 (77 277 users, 45 638 items, 1 949 342 edges, 3 layers, B = 2048)
  C4 DeepFM   Frappe-like (10 fields, D = 10, 3x400 MLP + BN, B = 4096)
Prints one JSON line per config, with the SpMM kernel's achieved bandwidth."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from recboard_amd import ops
from recboard_amd.deepfm import DeepFMEngine
from recboard_amd.gen import LightGCNEngine, MFEngine
from recboard_amd.graph import to_normalized_adj


def timeit(fn, iters=50, warmup=10):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters


def ev(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


rng = np.random.default_rng(1)
# ---- C1
U, N, B = 22363, 12101, 2048
m = MFEngine(U, N, 64, lr=1e-3, weight_decay=1e-6)
w = 1.0 / np.arange(1, N + 1); w /= w.sum()
u = torch.from_numpy(rng.integers(0, U, (B, 1))).cuda(); p = torch.from_numpy(rng.choice(N, (B, 1), p=w)).cuda()
n = torch.from_numpy(rng.integers(0, N, (B, 1))).cuda()
dt = timeit(lambda: m.train_step(u, p, n))
print(json.dumps({"config": "C1 MF-BPR Beauty B=2048", "ms_per_step": round(dt * 1e3, 4), "triplets_per_s": round(B / dt, 1)}))
# ---- C3
U, N, E = 77277, 45638, 1949342
deg_u = np.clip(rng.lognormal(2.6, 1.0, U), 1, 2000); deg_u = deg_u / deg_u.sum()
wi = 1.0 / np.arange(1, N + 1) ** 0.8; wi /= wi.sum()
eu = rng.choice(U, int(E * 1.08), p=deg_u); ei = rng.choice(N, int(E * 1.08), p=wi)
crow, col, val = to_normalized_adj(U, N, eu, ei)
nnz = len(col)
lg = LightGCNEngine(U, N, crow, col, val, 64, 3, lr=5e-3, weight_decay=1e-3)
with torch.no_grad():
    for q in lg.params.values():
        q.normal_(0, 0.1)
u = torch.from_numpy(rng.integers(0, U, (B, 1))).cuda(); p = torch.from_numpy(rng.choice(N, (B, 1), p=wi)).cuda()
n = torch.from_numpy(rng.integers(0, N, (B, 1))).cuda()
dt = timeit(lambda: lg.train_step(u, p, n), iters=20, warmup=5)
t_sp = ev(lambda: lg._spmm(lg.X0, lg.Xa))
rows = U + N
bytes_alg = nnz * 12 + rows * (8 + 4 * 64) + nnz * 256   # (col,val) + crow/Y + X rows (cache-resident upper figure)
print(json.dumps({"config": f"C3 LightGCN Yelp nnz={nnz} long_rows={lg.plan.nlong} chunks={lg.plan.nchunks} B=2048", "ms_per_step": round(dt * 1e3, 4),
                  "triplets_per_s": round(B / dt, 1), "spmm_ms": round(t_sp, 4),
                  "spmm_GBs_incl_X_rows": round(bytes_alg / t_sp / 1e6, 1), "spmm_GBs_hbm_stream": round((nnz * 12 + rows * 264) / t_sp / 1e6, 1),
                  "spmm_GFLOPs": round(2 * 64 * nnz / t_sp / 1e6, 1)}))
# ---- C4
counts = [957, 4082, 7, 7, 2, 3, 2, 9, 80, 233]
B = 4096
d = DeepFMEngine(counts, 10, (400, 400, 400), batch_norm=True, hidden_dropout_rate=0.2, lr=1e-3, embedding_decay=0.05)
x = torch.stack([torch.from_numpy(rng.integers(0, c, B)) for c in counts], 1).cuda()
y = torch.from_numpy((rng.random((B, 1)) < 0.3).astype(np.int64)).cuda()
dt = timeit(lambda: d.train_step(x, y), iters=30, warmup=5)
t_bag = ev(lambda: ops.fm_bag_fwd(d.T, d.TL.reshape(-1), d.bias, d.offsets, x))
print(json.dumps({"config": "C4 DeepFM Frappe-like F=10 D=10 B=4096 (all native)", "ms_per_step": round(dt * 1e3, 4),
                  "rows_per_s": round(B / dt, 1), "fm_bag_fwd_ms": round(t_bag, 4)}))
