"""Do the looped tile step's results depend on WHICH workgroup runs which tile after which?  Diagnostic build (make -C recboard_amd/csrc hov), one
workgroup per CU: odd workgroups are held back at their start and every fourth one in front of every further tile, by several tile lifetimes --
the ticket order changes, the results must not.      python scripts/tile_order_check.py [--B 2048] [--steps 3]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from recboard_amd import lib
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=2048)
ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
lib.LIB_PATH = os.path.join(ROOT, "recboard_amd", "librecengine_hov.so")
L = lib.load()
import bench
from recboard_amd.sasrec import SASRecEngine
L.re_dbg_tile_delay.argtypes, L.re_dbg_tile_delay.restype = [ctypes.c_uint], ctypes.c_int
cfg = dict(bench.BEAUTY, B=a.B)
bs = [tuple(torch.from_numpy(x).cuda() for x in b) for b in bench.synth_batches(cfg, 4, 1)]


def run(delay):
    assert L.re_dbg_tile_delay(delay) == 0
    m = SASRecEngine(cfg["items"], 50, 64, 2, dropout_rate=0.5, lr=5e-4, weight_decay=1e-6, seed=1)
    out = []
    for i in range(a.steps):
        loss = m.train_step_fused(*bs[i % 4])
        torch.cuda.synchronize()
        out.append((float(loss), m.arena.grad.clone(), m.arena.data.clone()))
    m.check_handover()
    return out


r0 = run(0)
bad = 0
for delay in (20000, 100000, 400000):
    r = run(delay)
    for i, (x, y) in enumerate(zip(r0, r)):
        same = x[0] == y[0] and torch.equal(x[1], y[1]) and torch.equal(x[2], y[2])
        print(f"delay {delay:7d} cycles step {i}: loss {y[0]:.6f} vs {x[0]:.6f}  identical {same}  gradient entries that differ {int((x[1] != y[1]).sum())}")
        bad += 0 if same else 1
print("RESULT:", "the results do not depend on the order in which workgroups take tiles" if bad == 0 else f"{bad} steps DEPEND on the tile order")
