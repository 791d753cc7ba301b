"""Does a tile of the looped tile step read LDS that the tile BEFORE it in the same workgroup left?  Diagnostic build (make -C recboard_amd/csrc hov):
every workgroup refills its tile-working LDS with a bit pattern in front of EVERY tile; the step's results must not depend on the pattern.
    python scripts/lds_between_tiles_check.py [--B 2048] [--steps 3]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from recboard_amd import lib
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=2048)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--lib", default="hov")
a = ap.parse_args()
lib.LIB_PATH = os.path.join(ROOT, "recboard_amd", f"librecengine_{a.lib}.so")
L = lib.load()
import bench
from recboard_amd.sasrec import SASRecEngine
L.re_dbg_tile_fill.argtypes, L.re_dbg_tile_fill.restype = [ctypes.c_uint], ctypes.c_int
cfg = dict(bench.BEAUTY, B=a.B)
bs = [tuple(torch.from_numpy(x).cuda() for x in b) for b in bench.synth_batches(cfg, 4, 1)]


def run(pattern):
    assert L.re_dbg_tile_fill(pattern) == 0
    m = SASRecEngine(cfg["items"], 50, 64, 2, dropout_rate=0.5, lr=5e-4, weight_decay=1e-6, seed=1)
    out = []
    for i in range(a.steps):
        loss = m.train_step_fused(*bs[i % 4])
        torch.cuda.synchronize()
        out.append((float(loss), m.arena.grad.clone(), m.arena.data.clone()))
    return out


r0 = run(0)
bad = 0
for name, pat in (("tiny", 1), ("NaN", 0x7FC00000), ("1.0", 0x3F800000), ("-3e38", 0xFF7FFFFF)):
    r = run(pat)
    for i, (x, y) in enumerate(zip(r0, r)):
        same = x[0] == y[0] and torch.equal(x[1], y[1]) and torch.equal(x[2], y[2])
        nan = bool(torch.isnan(y[1]).any() or torch.isnan(y[2]).any())
        print(f"fill {name:6s} step {i}: loss {y[0]:.6f} vs {x[0]:.6f}  identical {same}  NaN {nan}  gradient entries that differ {int((x[1] != y[1]).sum())}")
        bad += 0 if same else 1
print("RESULT:", "no tile reads LDS it did not write" if bad == 0 else f"{bad} steps DEPEND on what the LDS held in front of a tile")
