"""BASELINE config 5 on ONE GPU: SASRec d=128 on a synthetic 100 M-item table (item popularity Zipf(1.05),
S = 50, B = 512, BCE with one uniform negative, table ~ N(0, 0.02^2)).  The table (51 GB) and its Adam moments (2 x 51 GB)
live in HBM; a step touches ~3*B*S rows of them through the row-sparse optimizer.  Encoder: the fused D = 128 kernels (32-row work
items, longer sequences as chained parts), embedding front end / backward fused in.  Prints one JSON line.
    python scripts/bench_c5.py [--items 100000000] [--dim 128] [--steps 20]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--items", type=int, default=100_000_000)
ap.add_argument("--dim", type=int, default=128)
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
args = ap.parse_args()

from recboard_amd import ops
from recboard_amd.large import SASRecLargeTableEngine

N, D, B, S = args.items, args.dim, args.batch, 50
t0 = time.time()
model = SASRecLargeTableEngine(N, S, D, 2, dropout_rate=0.5, loss="BCE", lr=1e-3, weight_decay=1e-6, seed=1)
torch.cuda.synchronize()
t_init = time.time() - t0
rng = np.random.default_rng(1)
batches = []
for _ in range(4):
    lens = np.clip(rng.geometric(1 / 5.9, B) + 1, 1, S - 1)
    seq = np.zeros((B, S), np.int64)
    for b in range(B):
        seq[b, S - lens[b]:] = np.minimum(rng.zipf(1.05, lens[b]), N)          # ids 1..N, Zipf(1.05) popularity
    pos = np.where(seq > 0, np.minimum(rng.zipf(1.05, (B, S)), N) - 1, 0)
    neg = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
    batches.append(tuple(torch.from_numpy(a).cuda() for a in (seq, pos, neg)))
def timed(step):
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.steps, loss
dt_eager, _ = timed(lambda i: model.train_step(*batches[i % 4]))
dt, loss = timed(lambda i: model.train_step_graph(*batches[i % 4]))   # RAW batch in: one preparation launch + one graph replay

# where the step's time goes: the row-sparse optimizer alone, and the embedding front end alone
def ev(fn, iters=10):
    fn(); a, b = torch.cuda.Event(True), torch.cuda.Event(True); a.record()
    for _ in range(iters): fn()
    b.record(); b.synchronize(); return a.elapsed_time(b) / iters
seq, pos, neg = batches[0]
aux = model.prepare_batch(seq, pos, neg)
C = torch.randn(3 * B * S, D, device="cuda") * 1e-3
t_opt = ev(lambda: ops.sparse_adam_rows(C, aux.rows_all, model.E, model.Em, model.Ev, 5, 1e-3, padding_idx=0))
t_emb = ev(lambda: ops.sasrec_embed(model.E, model.params["Position.weight"].detach(), seq, float(D ** 0.5), 0.5, 7))
free, total = torch.cuda.mem_get_info()
print(json.dumps({"config": f"C5 SASRec d={D} L=2 maxlen=50 BCE, {N} items (table {4*(N+1)*D/1e9:.1f} GB + 2 moment tables), B={B}, 1 GPU",
                  "ms_per_step": round(dt * 1e3, 3), "samples_per_s": round(B / dt, 1), "launch": "hipGraph replay of the whole step",
                  "ms_per_step_eager_launches": round(dt_eager * 1e3, 3), "final_loss": round(float(loss), 5),
                  "sparse_adam_rows_ms": round(t_opt, 4), "rows_per_step": 3 * B * S, "embed_gather_ms": round(t_emb, 4),
                  "table_init_s": round(t_init, 1), "hbm_used_GB": round((total - free) / 1e9, 1),
                  "encoder": f"{model.encoder} (D = {D})"}))
