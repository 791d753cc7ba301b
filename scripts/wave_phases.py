"""Cycle accounting of one tile of the wave-per-tile step (csrc/enc_wave.hip) from the phase stamps of the diagnostic build
(make -C recboard_amd/csrc encprof).    python scripts/wave_phases.py [--kind beauty|8|33|...]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recboard_amd import lib
lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), "librecengine_encprof.so")
import numpy as np
import torch
from recboard_amd.sasrec import SASRecEngine

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="beauty")
ap.add_argument("--B", type=int, default=512)
ap.add_argument("--tpw", type=int, default=4)
ap.add_argument("--ncu", type=int, default=1024)
args = ap.parse_args()
B, S, D, L, N = args.B, 50, 64, 2, 12101
m = SASRecEngine(N, S, D, L, dropout_rate=0.5, loss="BCE", lr=5e-4, weight_decay=1e-6, seed=1)
rng = np.random.default_rng(0)
lens = np.clip(rng.geometric(1 / 5.9, B) + 1, 1, S - 1) if args.kind == "beauty" else np.full(B, int(args.kind))
seq = np.zeros((B, S), np.int64)
for b in range(B):
    seq[b, S - lens[b]:] = rng.integers(1, N + 1, lens[b])
pos = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
neg = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
seq, pos, neg = (torch.from_numpy(a).cuda() for a in (seq, pos, neg))
Lb = lib.load()
for _ in range(5):
    m.train_step(seq, pos, neg)
torch.cuda.synchronize()
hdr = m.prepare_batch(seq, pos, neg).plan.view(torch.int32)[:9].cpu().numpy()
print("items", hdr[0], "tiles", hdr[1], "long items", hdr[2], "mode", hdr[7])
buf = (ctypes.c_ulonglong * 96)()
Lb.re_dbg_enc_marks_wave.restype = ctypes.c_int
assert Lb.re_dbg_enc_marks_wave(buf) == 0
t = np.array(list(buf), dtype=np.int64)
nz = np.nonzero(t)[0]
t = t[: nz[-1] + 1]
names = ["x0"] + ["LN_a,put", "QKV,publish/wait", "scores,softmax", "PV,put", "Wo,LN_f,put", "W1,put", "W2"] * L + ["lastLN,head,sync"] + \
        ["lastLN'"] + ["dz,W2'", "W1',LN_f',put", "Wo'", "attn dP/dS", "attn dQ/dK/dV,xch,put", "Wq'Wk'Wv',LN_a'"] * L + ["embed'"]
d = np.diff(t)
print("total", int(t[-1] - t[0]), "ticks")
for i, x in enumerate(d):
    print(f"  {names[i] if i < len(names) else '?':18s} {int(x):7d}")

# the first 8 workgroups side by side (the tiles of the first long sequences: 4-tile sequences first), ticks since the earliest stamp
big = (ctypes.c_ulonglong * (8 * 96))()
Lb.re_dbg_enc_marks_blocks.restype = ctypes.c_int
if Lb.re_dbg_enc_marks_blocks(big) == 0:
    T = np.array(list(big), dtype=np.int64).reshape(8, 96)
    n = int((T[0] > 0).sum())
    print("(the counters of different XCDs are not aligned: ticks since each workgroup's own first stamp)")
    print(" " * 27 + "".join(f"    wg{b:d} " for b in range(8)))
    for i in range(n):
        print(f"{i:3d} {(names[i - 1] if 0 < i <= len(names) else 'start')[:22]:22s}" + "".join(f"{int(T[b, i] - T[b, 0]):8d}" for b in range(8)))
