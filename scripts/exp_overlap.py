"""Do kernels on a side stream overlap with the captured training step?  Enqueue the scatter plan (4 launches, ~25 us serial)
of some batch on a second stream every step and compare the step time with and without it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from recboard_amd import ops, lib
from recboard_amd.sasrec import SASRecEngine
cfg = bench.BEAUTY
m = SASRecEngine(cfg["items"], 50, 64, 2, dropout_rate=0.5, lr=5e-4, weight_decay=1e-6)
hb = bench.synth_batches(cfg, 8, 1)
blobs = []
for b in hb:
    seq, pos, neg = (torch.from_numpy(a).cuda() for a in b)
    blobs.append(m.pack_batch(seq, pos, neg))
B, S = hb[0][0].shape
L = lib.load()
rows = torch.randint(0, cfg["items"] + 1, (3 * B * S,), device="cuda")
ws = torch.empty(int(L.re_scatter_add_rows_workspace_bytes(rows.numel(), 64, cfg["items"] + 1)), dtype=torch.uint8, device="cuda")
side = torch.cuda.Stream()
def run(n, with_side):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        m.train_step_graph(blobs[i % 8], B, S)
        if with_side:
            with torch.cuda.stream(side):
                ops.scatter_plan(rows, 64, cfg["items"] + 1, ws, padding_idx=0)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
run(50, False); run(50, True)
for rep in range(3):
    print(f"graph only {run(400, False):.4f} ms/step   graph + plan on a side stream {run(400, True):.4f} ms/step")
# the plan alone
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(400): ops.scatter_plan(rows, 64, cfg["items"] + 1, ws, padding_idx=0)
torch.cuda.synchronize(); print(f"plan alone {(time.perf_counter() - t0) / 400 * 1e3:.4f} ms")
