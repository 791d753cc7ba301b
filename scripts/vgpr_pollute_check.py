"""Does the SASRec tile step read vector-register lanes it never wrote?  Every step is run behind a kernel that fills the register files with one
bit pattern (scripts/micro/vgpr_pollute.hip); with pattern 0 and with a NaN pattern the step's results must be the same bits.
    python scripts/vgpr_pollute_check.py [--B 512] [--steps 3] [--looped 0]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from recboard_amd.sasrec import SASRecEngine

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=512)
ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
P = ctypes.CDLL(os.path.join(ROOT, "scripts", "micro", "libpollute.so"))
P.pollute.argtypes, P.pollute.restype = [ctypes.c_uint, ctypes.c_void_p], ctypes.c_int
cfg = dict(bench.BEAUTY, B=a.B)
bs = [tuple(torch.from_numpy(x).cuda() for x in b) for b in bench.synth_batches(cfg, 4, 1)]


def run(pattern):
    m = SASRecEngine(cfg["items"], 50, 64, 2, dropout_rate=0.5, lr=5e-4, weight_decay=1e-6, seed=1)
    out = []
    for i in range(a.steps):
        assert P.pollute(pattern, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
        loss = m.train_step_fused(*bs[i % 4])
        torch.cuda.synchronize()
        out.append((float(loss), m.arena.grad.clone(), m.arena.data.clone()))
    return out

r0 = run(0x00000000)
r1 = run(0x7FC00000)          # quiet NaN
r2 = run(0x3F800000)          # 1.0
bad = 0
for name, r in (("NaN", r1), ("1.0", r2)):
    for i, (x, y) in enumerate(zip(r0, r)):
        same = x[0] == y[0] and torch.equal(x[1], y[1]) and torch.equal(x[2], y[2])
        nan = bool(torch.isnan(y[1]).any() or torch.isnan(y[2]).any())
        nd = int((x[1] != y[1]).sum())
        print(f"pattern {name} step {i}: loss {y[0]:.6f} vs {x[0]:.6f}  identical {same}  NaN in results {nan}  gradient entries that differ {nd}")
        bad += 0 if same else 1
print("RESULT:", "the step's results do not depend on what the registers held" if bad == 0 else f"{bad} steps DEPEND on stale register contents")
