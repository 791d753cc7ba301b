"""enc_step_k by batch shape (run under scripts/prof_seq.sh <tag> enc_step ...): Beauty-shaped lengths with split on / off, and the
same lengths clipped to 32 / 16 rows (what the launch costs when no long sequence exists)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recboard_amd.sasrec import SASRecEngine
B, S, N = 512, 50, 12101
rng = np.random.default_rng(0)
base = np.clip(rng.geometric(1 / 5.9, B) + 1, 1, S - 1)
for name, lens, split in (("beauty split", base, True), ("beauty whole", base, False), ("clip32", np.minimum(base, 32), True),
                          ("clip16", np.minimum(base, 16), True), ("3 long of 48", np.concatenate([np.minimum(base[:-3], 16), [48, 48, 48]]), True),
                          ("3 long of 48 whole", np.concatenate([np.minimum(base[:-3], 16), [48, 48, 48]]), False)):
    seq = np.zeros((B, S), np.int64)
    for b in range(B):
        seq[b, S - lens[b]:] = rng.integers(1, N + 1, lens[b])
    pos = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
    neg = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
    batch = tuple(torch.from_numpy(a).cuda() for a in (seq, pos, neg))
    m = SASRecEngine(N, S, 64, 2, dropout_rate=0.5, loss="BCE", lr=5e-4, seed=1)
    m.split_long = split
    for _ in range(4):
        m.train_step(*batch)
    torch.cuda.synchronize()
    hdr = m.prepare_batch(*batch).plan.view(torch.int32)[:4].cpu().numpy()
    print(name, "items", hdr[0], "tiles", hdr[1])
