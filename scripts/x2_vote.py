"""Sweep of the split score kernel's workgroup-drain vote threshold (re_dbg_score_vote), Beauty shape, iid scores."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recboard_amd import ops, lib
L = lib.load()
L.re_dbg_score_vote.argtypes = [ctypes.c_int]; L.re_dbg_score_vote.restype = None
U, N, D = 22363, 12101, 64
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(U, D, device="cuda", generator=g); E = torch.randn(N, D, device="cuda", generator=g)
sp = torch.arange(0, U + 1, device="cuda") * 8
si = torch.sort(torch.randint(0, N, (U, 8), device="cuda", generator=g), 1).values.reshape(-1)
def t(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / it
for v in (-1, 2, 4, 6, 8, 10, 11):
    L.re_dbg_score_vote(v)
    print(f"vote_at {v:3d}: {t(lambda: ops.score_topk(q, E, sp, si, 50)):.3f} ms", flush=True)
L.re_dbg_score_vote(-1)
