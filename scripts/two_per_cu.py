"""Timing only: the tile kernel at one and at two workgroups per CU (LDS request 84 / 60 KB) on the diagnostic builds -- hovn: SGPR spills
through VGPR lanes (the product's code generation; UNSAFE at two per CU, profiles/r4_handover_notes.txt), hovs: SGPR spills to scratch.
    python scripts/two_per_cu.py hovs 60 2048"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from recboard_amd import lib  # noqa: E402
name, kb, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
if name != "product":
    lib.LIB_PATH = os.path.join(ROOT, "recboard_amd", f"librecengine_{name}.so")
L = lib.load()
if name != "product":
    L.re_dbg_tile_handover.argtypes, L.re_dbg_tile_handover.restype = [ctypes.c_int, ctypes.c_int], ctypes.c_int
    assert L.re_dbg_tile_handover(0, kb) == 0
import bench  # noqa: E402
from recboard_amd.sasrec import SASRecEngine  # noqa: E402
cfg = dict(bench.BEAUTY, B=B)
m = SASRecEngine(cfg["items"], cfg["S"], cfg["D"], cfg["L"], dropout_rate=cfg["p_drop"], loss="BCE", lr=cfg["lr"], weight_decay=cfg["wd"], seed=1)
bs = [tuple(torch.from_numpy(a).cuda() for a in b) for b in bench.synth_batches(cfg, 4, seed=11)]
for i in range(6):
    m.train_step_graph(*bs[i % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(40):
    m.train_step_graph(*bs[i % 4])
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 40 * 1e3
m.check_handover()
print(f"{name} lds {kb} KB B {B}: {ms:.4f} ms/step", flush=True)
