"""Host time of one train_step_graph call (launches are asynchronous: after a sync the first calls return as fast as the host can issue them)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from recboard_amd.sasrec import SASRecEngine
cfg = bench.BEAUTY
bs = [tuple(torch.from_numpy(a).cuda() for a in b) for b in bench.synth_batches(cfg, 8, 1)]
for pipe in (False, True):
    m = SASRecEngine(cfg["items"], 50, 64, 2, dropout_rate=0.5, lr=5e-4, weight_decay=1e-6, seed=1)
    for i in range(40):
        m.train_step_graph(*bs[i % 8], next_batch=bs[(i + 1) % 8] if pipe else None)
    best = 1e9
    for rep in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(8):
            m.train_step_graph(*bs[i % 8], next_batch=bs[(i + 1) % 8] if pipe else None)
        best = min(best, (time.perf_counter() - t0) / 8)
    torch.cuda.synchronize()
    print(f"pipelined={pipe}: host {best * 1e6:.1f} us per call", flush=True)
