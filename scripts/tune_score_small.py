import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recboard_amd import ops, lib
L = lib.load(); L.re_dbg_score_variant.argtypes = [ctypes.c_int, ctypes.c_int64]; L.re_dbg_score_variant.restype = None
g = torch.Generator(device="cuda").manual_seed(1)
N, D = 12101, 64
E = torch.randn(N, D, device="cuda", generator=g)
def t(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / it
for U in (512, 2048, 4096):
    q = torch.randn(U, D, device="cuda", generator=g)
    sp = torch.arange(0, U + 1, device="cuda") * 8
    si = torch.sort(torch.randint(0, N, (U, 8), device="cuda", generator=g), 1).values.reshape(-1)
    for minseg in (1, 4, 8, 16, 32, 64, 190):
        L.re_dbg_score_variant(3, minseg)
        print(f"B={U:5d} minseg={minseg:3d}: {t(lambda: ops.score_topk(q, E, sp, si, 50)):.3f} ms")
