"""ms per captured SASRec step (Beauty shapes, pipelined as the bench runs it) on a chosen build of the library:
    python scripts/step_time_lib.py [product|fenced|hovn] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from recboard_amd import lib  # noqa: E402
which = sys.argv[1] if len(sys.argv) > 1 else "product"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
if which != "product":
    lib.LIB_PATH = os.path.join(ROOT, "recboard_amd", f"librecengine_{which}.so")
lib.load()
import bench  # noqa: E402
from recboard_amd.sasrec import SASRecEngine  # noqa: E402
cfg = bench.BEAUTY
bs = [tuple(torch.from_numpy(x).cuda() for x in b) for b in bench.synth_batches(cfg, 8, 1)]
m = SASRecEngine(cfg["items"], 50, 64, 2, dropout_rate=0.5, lr=5e-4, weight_decay=1e-6, seed=1)
best = 1e9
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        m.train_step_graph(*bs[i % 8], next_batch=bs[(i + 1) % 8])
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / steps * 1e3)
print(f"{which}: {best:.4f} ms per step (best of 4 x {steps})")
