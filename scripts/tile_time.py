"""The D = 64 encoder step's launches by batch shape (run under rocprofv3 --kernel-trace; diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recboard_amd.sasrec import SASRecEngine
S, N = 50, 12101
rng = np.random.default_rng(0)
base = np.clip(rng.geometric(1 / 5.9, 1024) + 1, 1, S - 1)
for name, lens in (("clip16 B128", np.minimum(base[:128], 16)), ("clip16 B256", np.minimum(base[:256], 16)), ("clip16 B440", np.minimum(base[:440], 16)), ("clip16 B512", np.minimum(base[:512], 16)),
                   ("clip16 B900", np.minimum(base[:900], 16)), ("beauty B512", base[:512]), ("beauty B400", base[:400])):
    B = len(lens)
    seq = np.zeros((B, S), np.int64)
    for b in range(B):
        seq[b, S - lens[b]:] = rng.integers(1, N + 1, lens[b])
    pos = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
    neg = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
    batch = tuple(torch.from_numpy(a).cuda() for a in (seq, pos, neg))
    m = SASRecEngine(N, S, 64, 2, dropout_rate=0.5, loss="BCE", lr=5e-4, seed=1)
    for _ in range(13):
        l = m.train_step(*batch)
    torch.cuda.synchronize()
    m.check_handover()
    hdr = m.prepare_batch(*batch).plan.view(torch.int32)[:8].cpu().numpy()
    print(name, "items", hdr[0], "tiles", hdr[1], "long items", hdr[2], "mode", hdr[7], "loss", float(l))
