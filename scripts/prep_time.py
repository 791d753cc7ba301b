"""rocprofv3 target: batch-prep variants (plan only / full / small batch); read the kernel durations from the trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recboard_amd import ops
rng = np.random.default_rng(0)
def mk(B, S=50, N=12101):
    lens = np.clip(rng.geometric(1 / 5.9, B) + 1, 1, S - 1)
    seq = np.zeros((B, S), np.int64)
    for b in range(B):
        seq[b, S - lens[b]:] = rng.integers(1, N + 1, lens[b])
    pos = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
    return tuple(torch.from_numpy(a).cuda() for a in (seq, pos, pos.copy()))
for B in (512, 64, 2048):
    seq, pos, neg = mk(B)
    blob = torch.zeros(ops.prep_layout(B, 50)[1], dtype=torch.uint8, device="cuda")
    for _ in range(20):
        ops.sasrec_batch_prep(seq)                       # plan only  (grid 1)
    torch.cuda.synchronize()
    for _ in range(20):
        ops.sasrec_batch_prep(seq, pos, neg, blob=blob)  # full
    torch.cuda.synchronize()
