"""Does the tile step lean on its waves running in step?  Diagnostic build (make -C recboard_amd/csrc hov), one workgroup per CU: behind every barrier
of the kernel the four waves of a workgroup wait 0 .. 3 x N cycles, a different wave longest each time; the results must not change.
    python scripts/wave_skew_check.py [--B 512] [--steps 3]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from recboard_amd import lib
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=512)
ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
lib.LIB_PATH = os.path.join(ROOT, "recboard_amd", "librecengine_hov.so")
L = lib.load()
import bench
from recboard_amd.sasrec import SASRecEngine
L.re_dbg_tile_skew.argtypes, L.re_dbg_tile_skew.restype = [ctypes.c_uint], ctypes.c_int
cfg = dict(bench.BEAUTY, B=a.B)
bs = [tuple(torch.from_numpy(x).cuda() for x in b) for b in bench.synth_batches(cfg, 4, 1)]


def run(skew):
    assert L.re_dbg_tile_skew(skew) == 0
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
for skew in (200, 2000, 20000):
    r = run(skew)
    for i, (x, y) in enumerate(zip(r0, r)):
        same = x[0] == y[0] and torch.equal(x[1], y[1]) and torch.equal(x[2], y[2])
        print(f"skew {skew:6d} cycles step {i}: loss {y[0]:.6f} vs {x[0]:.6f}  identical {same}  gradient entries that differ {int((x[1] != y[1]).sum())}")
        bad += 0 if same else 1
print("RESULT:", "the results do not depend on the waves running in step" if bad == 0 else f"{bad} steps DEPEND on the waves' relative timing: a missing barrier or wait")
