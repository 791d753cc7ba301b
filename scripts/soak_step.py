"""Soak of the captured training steps over many DIFFERENT random batches (plan variety: short-only, many long sequences, split and
unsplit long items, single-item sequences): losses stay finite, the split halves' hand-over never times out, the graph and the eager
step stay in agreement.    python scripts/soak_step.py [--batches 1500]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from recboard_amd.sasrec import SASRecEngine
from recboard_amd.large import SASRecLargeTableEngine

ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, default=1500)
args = ap.parse_args()
rng = np.random.default_rng(0)
B, S = 512, 50


def batch(N, mean_len, p_long):
    lens = np.clip(rng.geometric(1.0 / mean_len, B) + 1, 1, S - 1)
    long_rows = rng.random(B) < p_long
    lens[long_rows] = rng.integers(17, S, int(long_rows.sum()))
    col = np.arange(S)[None, :]
    real = col >= (S - lens)[:, None]
    seq = np.where(real, rng.integers(1, N + 1, (B, S)), 0)
    pos = np.where(real, rng.integers(0, N, (B, S)), 0)
    neg = np.where(real, rng.integers(0, N, (B, S)), 0)
    return tuple(torch.from_numpy(a.astype(np.int64)).cuda() for a in (seq, pos, neg))


for name, make in (("SASRecEngine d=64", lambda: SASRecEngine(12101, S, 64, 2, dropout_rate=0.5, loss="BCE", lr=1e-3, seed=1)),
                   ("SASRecLargeTableEngine d=128", lambda: SASRecLargeTableEngine(2_000_000, S, 128, 2, dropout_rate=0.5, loss="BCE", lr=1e-3, seed=1))):
    # eng: every step is told the NEXT batch (prepared by jobs of its tail launch); twin: the same batches, each prepared in front of its step
    eng, twin = make(), make()
    N = eng.N
    t0 = time.time()
    worst = 0.0
    nxt = batch(N, 2.0, 0.0)
    for i in range(args.batches):
        mean_len = (2.0, 5.9, 12.0)[(i + 1) % 3]
        p_long = (0.0, 0.05, 0.3, 0.9)[((i + 1) // 3) % 4]
        b, nxt = nxt, batch(N, mean_len, p_long)
        loss = eng.train_step_graph(*b, next_batch=nxt if i % 37 != 36 else None)      # (now and then nobody announces the next batch)
        ref = twin.train_step_graph(*b)
        if i % 20 == 0:
            assert torch.equal(loss, ref), (name, i, float(loss), float(ref))
        if i % 100 == 0:
            v = float(loss)
            assert np.isfinite(v), (name, i, v)
            eng.check_handover()
            worst = max(worst, v)
    torch.cuda.synchronize()
    eng.check_handover()
    assert torch.equal(eng.arena.data, twin.arena.data), name
    print(f"{name}: {args.batches} batches ok (pipelined = unpipelined, bit for bit), {time.time() - t0:.1f} s, last loss {float(loss):.4f}", flush=True)
