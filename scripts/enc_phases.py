"""Cycle accounting of one encoder work item (the plan's largest) from the phase stamps of the diagnostic build
(make -C recboard_amd/csrc encprof).    python scripts/enc_phases.py [--kind 49]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recboard_amd import lib
lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), "librecengine_encprof.so")
import numpy as np
import torch
from recboard_amd import ops
from recboard_amd.sasrec import SASRecEngine

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="beauty")
ap.add_argument("--B", type=int, default=512)
args = ap.parse_args()
B, S, D, L, N = args.B, 50, 64, 2, 12101
m = SASRecEngine(N, S, D, L, dropout_rate=0.5, loss="BCE", lr=5e-4, weight_decay=1e-6, seed=1)
m.fused_item_kernel = False   # the stamps live in the forward / backward kernels of their own (the one-launch item kernel has none)
rng = np.random.default_rng(0)
if args.kind == "beauty":
    lens = np.clip(rng.geometric(1 / 5.9, B) + 1, 1, S - 1)
elif args.kind == "le16":
    lens = np.clip(rng.geometric(1 / 5.9, B) + 1, 1, 16)
elif args.kind == "17-32":
    lens = rng.integers(17, 33, B)
else:
    lens = np.full(B, int(args.kind))
seq = np.zeros((B, S), np.int64)
for b in range(B):
    seq[b, S - lens[b]:] = rng.integers(1, N + 1, lens[b])
pos = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
neg = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
seq, pos, neg = (torch.from_numpy(a).cuda() for a in (seq, pos, neg))
Lb = lib.load()
for _ in range(5):
    loss = m.train_step(seq, pos, neg)
torch.cuda.synchronize()
pb = m.prepare_batch(seq, pos, neg)
hdr = pb.plan.view(torch.int32)[:9].cpu().numpy()
print("items", hdr[0], "tiles", hdr[1], "largest item: nt", (int(hdr[8]) >> 24) & 15, " plan kernel ticks: spans", hdr[5], "passes", hdr[6], "rows", hdr[7])
for which, fn, names in (
        ("fwd", Lb.re_dbg_enc_marks_fwd, ["decode", "x0", "LN_a", "QKV", "scores", "softmax", "PV", "Wo", "LN_f", "W1", "W2"]),
        ("bwd", Lb.re_dbg_enc_marks_bwd, None)):
    buf = (ctypes.c_ulonglong * 96)()
    fn.restype = ctypes.c_int
    assert fn(buf) == 0
    t = np.array(list(buf), dtype=np.int64)
    nz = np.nonzero(t)[0]
    t = t[: nz[-1] + 1]
    d = np.diff(t)
    print(which, "total", int(t[-1] - t[0]), "ticks;  phase deltas:", " ".join(str(int(x)) for x in d))
