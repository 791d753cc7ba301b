"""Where does the prefetch pipeline lose its gain?  Time the captured step with the plan prefetch under variants of the
cross-stream synchronisation (timing only: the no-wait variants are not safe)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from recboard_amd.sasrec import SASRecEngine
cfg = bench.BEAUTY
hb = bench.synth_batches(cfg, 8, 1)
B, S = hb[0][0].shape
def make():
    m = SASRecEngine(cfg["items"], 50, 64, 2, dropout_rate=0.5, lr=5e-4, weight_decay=1e-6)
    blobs = [m.pack_batch(*(torch.from_numpy(a).cuda() for a in b)) for b in hb]
    torch.cuda.synchronize()
    return m, blobs
def run(m, blobs, n, prefetch):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        m.train_step_graph(blobs[i % 8], B, S)
        if prefetch:
            m.prefetch_plan(blobs[(i + 1) % 8], B, S)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
m, blobs = make(); run(m, blobs, 50, False)
print(f"no prefetch                {run(m, blobs, 400, False):.4f} ms/step")
m, blobs = make(); run(m, blobs, 50, True)
print(f"prefetch, all waits        {run(m, blobs, 400, True):.4f} ms/step")
orig_wait = torch.cuda.Stream.wait_event
torch.cuda.Stream.wait_event = lambda self, ev: None
m, blobs = make(); run(m, blobs, 50, True)
print(f"prefetch, no waits         {run(m, blobs, 400, True):.4f} ms/step")
orig_rec = torch.cuda.Event.record
torch.cuda.Event.record = lambda self, stream=None: None
m, blobs = make(); run(m, blobs, 50, True)
print(f"prefetch, no waits/records {run(m, blobs, 400, True):.4f} ms/step")
# CPU cost of the enqueue path alone
torch.cuda.Stream.wait_event = orig_wait; torch.cuda.Event.record = orig_rec
m, blobs = make(); run(m, blobs, 50, True)
t0 = time.perf_counter()
for i in range(200):
    m.prefetch_plan(blobs[(i + 1) % 8], B, S)
cpu = (time.perf_counter() - t0) / 200 * 1e3
torch.cuda.synchronize()
print(f"CPU time of one prefetch_plan call: {cpu:.4f} ms")
