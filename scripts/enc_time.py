"""Timing of the fused encoder launches by sequence-length class (which work-item size sets a launch's time).
    python scripts/enc_time.py [--B 512]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from recboard_amd import lib
if os.environ.get("RE_LIB_VARIANT"):   # an experimental build of the library (make var / a hand-linked .so next to librecengine.so)
    lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), os.environ["RE_LIB_VARIANT"])
from recboard_amd import ops
from recboard_amd.sasrec import SASRecEngine

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=512)
ap.add_argument("--kinds", default="beauty,le16,17-32,33-48,49")
args = ap.parse_args()
B, S, D, L, N = args.B, 50, 64, 2, 12101
m = SASRecEngine(N, S, D, L, dropout_rate=0.5, loss="BCE", lr=5e-4, weight_decay=1e-6, seed=1)
rng = np.random.default_rng(0)


def batch(kind):
    if kind == "beauty":
        lens = np.clip(rng.geometric(1 / 5.9, B) + 1, 1, S - 1)
    elif kind == "le16":
        lens = np.clip(rng.geometric(1 / 5.9, B) + 1, 1, 16)
    elif kind == "17-32":
        lens = rng.integers(17, 33, B)
    elif kind == "33-48":
        lens = rng.integers(33, 49, B)
    else:
        lens = np.full(B, int(kind))     # every sequence this long
    seq = np.zeros((B, S), np.int64)
    for b in range(B):
        seq[b, S - lens[b]:] = rng.integers(1, N + 1, lens[b])
    pos = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
    neg = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
    return tuple(torch.from_numpy(a).cuda() for a in (seq, pos, neg))


def ev(fn, iters=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / iters * 1e3


P = m.params
E, Pp = P["Item.embeddings.weight"].detach(), P["Position.weight"].detach()
lw, lb = P["lastLN.weight"].detach(), P["lastLN.bias"].detach()
bt, bg = m._block_tensors(), m._block_tensors(m.arena.grad)
G = m.arena.views(m.arena.grad)
W = m._buffers(B, S)
for kind in args.kinds.split(","):
    seq, pos, neg = batch(kind)
    pb = m.prepare_batch(seq, pos, neg)
    hdr = pb.plan.view(torch.int32)[:8].cpu().numpy()
    t_prep = ev(lambda: m.prepare_batch(seq, pos, neg))
    t_fe = ev(lambda: ops.sasrec_embed_encoder_fwd(E, Pp, seq, 8.0, bt, lw, lb, L, 0.0, 1, plan=pb.plan, out=W["u"]))
    t_ft = ev(lambda: ops.sasrec_embed_encoder_fwd(E, Pp, seq, 8.0, bt, lw, lb, L, 0.5, 1, need_tape=True, plan=pb.plan, out=W["u"], tape=W["tape"]))
    dU = torch.randn(B, S, D, device="cuda") * 1e-3
    C = W["contrib"][:B * S].view(B, S, D)
    t_b = ev(lambda: ops.sasrec_encoder_embed_bwd(dU, seq, 8.0, bt, lw, lb, L, 0.5, 1, W["tape"], bg, G["lastLN.weight"], G["lastLN.bias"],
                                                 G["Position.weight"], out=C, ws=W["ws_bwd"], plan=pb.plan))
    print(f"{kind:7s} items {hdr[0]:4d} tiles {hdr[1]:5d} long {hdr[2]:4d} G {hdr[3]}   prep {t_prep:6.1f}  fwd(eval) {t_fe:6.1f}  fwd(train) {t_ft:6.1f}  "
          f"bwd(3 launches) {t_b:6.1f} us")
