"""The captured SASRec step (Beauty shapes, D = 64) on one library build at several batch sizes: ms per step, launched like the bench's
large-batch leg (the next batch prepared by the step's tail) and plainly.     python scripts/tile_wgs_ab.py product|one 512 1024 2048 4096 8192"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from recboard_amd import lib  # noqa: E402
name = sys.argv[1]
if name != "product":
    lib.LIB_PATH = os.path.join(ROOT, "recboard_amd", f"librecengine_{name}.so")
lib.load()
import bench  # noqa: E402
from recboard_amd.sasrec import SASRecEngine  # noqa: E402
for B in [int(x) for x in sys.argv[2:]]:
    cfg = dict(bench.BEAUTY, B=B)
    m = SASRecEngine(cfg["items"], cfg["S"], cfg["D"], cfg["L"], dropout_rate=cfg["p_drop"], loss="BCE", lr=cfg["lr"], weight_decay=cfg["wd"], seed=1)
    m.prep_in_tail = True
    if os.environ.get("TILE_ALWAYS"):
        m.tile_step = "always"          # (the plan's speed rule overridden: the tile kernels whatever the batch)
    bs = [tuple(torch.from_numpy(a).cuda() for a in b) for b in bench.synth_batches(cfg, 4, seed=11)]
    out = []
    for nxt in (False, True):
        for i in range(8):
            m.train_step_graph(*bs[i % 4], next_batch=bs[(i + 1) % 4] if nxt else None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(60):
            m.train_step_graph(*bs[i % 4], next_batch=bs[(i + 1) % 4] if nxt else None)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 60 * 1e3)
    m.check_handover()
    print(f"{name} (tile workgroups per CU {m._tile_wgs()}) B {B}: {out[0]:.4f} ms/step plain, {out[1]:.4f} with the next batch prepared in the tail", flush=True)
