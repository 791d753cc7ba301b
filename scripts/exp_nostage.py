"""What the stage launch in front of every replay costs: the pipelined SASRec step as shipped (stage launch + graph replay) against the two
captured copies replayed back to back without it (timing only: the step scalars stay those of the last staged step)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from recboard_amd.sasrec import SASRecEngine  # noqa: E402
cfg = bench.BEAUTY
bs = [tuple(torch.from_numpy(x).cuda() for x in b) for b in bench.synth_batches(cfg, 8, 1)]
m = SASRecEngine(cfg["items"], 50, 64, 2, dropout_rate=0.5, lr=5e-4, weight_decay=1e-6, seed=1)
for i in range(40):
    m.train_step_graph(*bs[i % 8], next_batch=bs[(i + 1) % 8])
torch.cuda.synchronize()
N = 400
t0 = time.time()
for i in range(N):
    m.train_step_graph(*bs[i % 8], next_batch=bs[(i + 1) % 8])
torch.cuda.synchronize()
print("stage launch + replay: %.4f ms per step" % ((time.time() - t0) / N * 1e3))
tp = next(iter(m._tail_pipes.values()))
gs = [g["graph"] for g in tp["graphs"]]
t0 = time.time()
for i in range(N):
    gs[i & 1].replay()
torch.cuda.synchronize()
print("replays alone:         %.4f ms per step" % ((time.time() - t0) / N * 1e3))
