"""The Beauty step launched kernel by kernel (no hipGraph) against the captured step:  python scripts/step_eager.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from recboard_amd.sasrec import SASRecEngine
cfg = bench.BEAUTY
bs = [tuple(torch.from_numpy(a).cuda() for a in b) for b in bench.synth_batches(cfg, 8, 1)]
def by_launch(m):
    """the captured step's launches, enqueued one by one instead of replayed"""
    def f(seq, pos, neg):
        key = (seq.shape[0], seq.shape[1], True, m.training)
        g = getattr(m, "_graphs", {}).get(key)
        if g is None:
            return m.train_step_graph(seq, pos, neg)
        A = m.arena
        m._stage(g, seq, pos, neg, A.step + 1)
        loss = m._step_body(g["pb"], 0, seed_dev=g["state"], adam_hyper=g["hyper"])
        A.step += 1
        return loss[0]
    return f


for mode in ("graph", "launches", "eager"):
    m = SASRecEngine(cfg["items"], 50, 64, 2, dropout_rate=0.5, lr=5e-4, weight_decay=1e-6, seed=1)
    f = m.train_step_graph if mode == "graph" else by_launch(m) if mode == "launches" else (lambda s, p, n: m.train_step(s, p, n, None))
    for i in range(30):
        f(*bs[i % 8])
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for i in range(300):
            loss = f(*bs[i % 8])
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 300)
    print(f"{mode}: {best * 1e6:.1f} us/step  loss {float(loss):.5f}", flush=True)
