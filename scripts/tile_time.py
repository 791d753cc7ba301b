"""enc_tile_step_k by tiles per workgroup and batch shape (run under rocprofv3 --kernel-trace; diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recboard_amd.sasrec import SASRecEngine
B, S, N = 512, 50, 12101
rng = np.random.default_rng(0)
base = np.clip(rng.geometric(1 / 5.9, B) + 1, 1, S - 1)
for name, lens, tpw, ncu in (("clip16 tpw1", np.minimum(base, 16), 1, 1024), ("clip16 tpw2", np.minimum(base, 16), 2, 150), ("clip16 tpw4 G1", np.minimum(base, 16), 4, 1024),
                             ("clip16 tpw4 G4", np.minimum(base, 16), 4, 64), ("beauty tpw4 G1", base, 4, 1024), ("beauty tpw4 G2", base, 4, 180)):
    seq = np.zeros((B, S), np.int64)
    for b in range(B):
        seq[b, S - lens[b]:] = rng.integers(1, N + 1, lens[b])
    pos = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
    neg = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
    batch = tuple(torch.from_numpy(a).cuda() for a in (seq, pos, neg))
    m = SASRecEngine(N, S, 64, 2, dropout_rate=0.5, loss="BCE", lr=5e-4, seed=1)
    m.tiles_per_wg = tpw
    m._plan_ncu = lambda ncu=ncu: ncu
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        l = m.train_step(*batch)
    t0.record()
    for _ in range(10):
        l = m.train_step(*batch)
    t1.record(); torch.cuda.synchronize()
    hdr = m.prepare_batch(*batch).plan.view(torch.int32)[:4].cpu().numpy()
    print(name, "items", hdr[0], "tiles", hdr[1], "G", hdr[3], "loss", float(l), f"eager step {t0.elapsed_time(t1) / 10 * 1e3:.0f} us")
