import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recboard_amd import ops, lib
L = lib.load()
L.re_dbg_score_diag.argtypes = [ctypes.c_int]; L.re_dbg_score_diag.restype = None
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
L.re_dbg_score_diag(1); a = t(lambda: ops.score_topk(q, E, sp, si, 50))
L.re_dbg_score_diag(0); b = t(lambda: ops.score_topk(q, E, sp, si, 50))
print(f"{sys.argv[1] if len(sys.argv) > 1 else '':28s} no hits {a:.3f} ms   normal {b:.3f} ms")
