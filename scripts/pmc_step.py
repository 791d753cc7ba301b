"""A few fused SASRec steps + one full-catalog scoring launch: the target of `rocprofv3 --pmc` runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from recboard_amd import ops
from recboard_amd.sasrec import SASRecEngine
cfg = bench.BEAUTY
m = SASRecEngine(cfg["items"], 50, 64, 2, dropout_rate=0.5, lr=5e-4, weight_decay=1e-6)
hb = bench.synth_batches(cfg, 2, 1)
bs = [tuple(torch.from_numpy(a).cuda() for a in b) for b in hb]
for i in range(6):
    s, p, n = bs[i % 2]
    m.train_step(s, p, n, m.prepare_batch(s, p, n))      # eager launches (batch preparation + the step's kernels), as the graph replays them
m.fuse_tail = False                                       # ... and with the tail as separate launches (enc_wgrad_k, scatter_owner_k): per-kernel attribution
for i in range(6):
    s, p, n = bs[i % 2]
    m.train_step(s, p, n, m.prepare_batch(s, p, n))
U, N = 22363, 12101
gq = torch.Generator(device="cuda").manual_seed(11)
q = torch.randn(U, 64, device="cuda", generator=gq); E = torch.randn(N, 64, device="cuda", generator=gq)
sp = torch.arange(0, U + 1, device="cuda") * 8
si = torch.sort(torch.randint(0, N, (U, 8), device="cuda"), 1).values.reshape(-1)
for _ in range(3):
    ops.score_topk(q, E, sp, si, 50)
W = torch.randn(16 * 1024 * 1024, 64, device="cuda"); idx = torch.randint(0, W.shape[0], (4 * 1024 * 1024,), device="cuda")
for _ in range(3):
    ops.gather_rows(W, idx)
torch.cuda.synchronize()
