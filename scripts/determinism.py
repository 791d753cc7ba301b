"""Is the captured SASRec step reproducible run to run?  Two engines, the same seeds and batches, N steps each: compare bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from recboard_amd.sasrec import SASRecEngine
cfg = bench.BEAUTY
bs = [tuple(torch.from_numpy(a).cuda() for a in b) for b in bench.synth_batches(cfg, 8, 1)]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
res = []
for rep in range(2):
    m = SASRecEngine(cfg["items"], 50, D, 2, dropout_rate=0.5, lr=5e-4, weight_decay=1e-6, seed=1)
    for k in sys.argv[3:]:
        setattr(m, k.lstrip("-"), False)
    hist = []
    for i in range(steps):
        nxt = bs[(i + 1) % 8] if os.environ.get("DET_NEXT") else None      # (DET_NEXT=1: the pipelined step -- the next batch prepared by the tail launch)
        hist.append(m.train_step_graph(*bs[i % 8], next_batch=nxt).clone())
    torch.cuda.synchronize()
    res.append((torch.stack(hist), m.arena.data.clone()))
d = (res[0][0] != res[1][0]).nonzero().reshape(-1)
print("first step whose loss differs:", int(d[0]) if d.numel() else None, "| parameters identical:", bool(torch.equal(res[0][1], res[1][1])))
import hashlib
print("sha1 of the parameters:", hashlib.sha1(res[0][1].cpu().numpy().tobytes()).hexdigest()[:16], "last loss", float(res[0][0][-1]))
if os.environ.get("DET_OUT"):
    import numpy as np
    np.save(os.environ["DET_OUT"], res[0][0].cpu().numpy())
