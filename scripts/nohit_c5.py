import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recboard_amd import ops, lib
L = lib.load()
L.re_dbg_score_diag.argtypes = [ctypes.c_int]; L.re_dbg_score_diag.restype = None
L.re_dbg_score_counters.argtypes = [ctypes.c_void_p, ctypes.c_int]; L.re_dbg_score_counters.restype = None
def t(fn, it=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / it
for U, N in ((512, 12_500_000), (77277, 45638)):
    g = torch.Generator(device="cuda").manual_seed(1)
    q = torch.randn(U, 64, device="cuda", generator=g); E = torch.randn(N, 64, device="cuda", generator=g)
    sp = torch.arange(0, U + 1, device="cuda") * 8
    si = torch.sort(torch.randint(0, N, (U, 8), device="cuda", generator=g), 1).values.reshape(-1)
    L.re_dbg_score_diag(1); a = t(lambda: ops.score_topk(q, E, sp, si, 50))
    L.re_dbg_score_diag(0); b = t(lambda: ops.score_topk(q, E, sp, si, 50))
    buf = (ctypes.c_ulonglong * 4)()
    L.re_dbg_score_diag(3); L.re_dbg_score_counters(buf, 1)
    ops.score_topk(q, E, sp, si, 50); torch.cuda.synchronize(); L.re_dbg_score_counters(buf, 1); L.re_dbg_score_diag(0)
    fl = 2 * 64 * U * N
    print(f"{U} x {N}: no hits {a:.3f} ms ({fl/a/1e9:.1f} TF)   normal {b:.3f} ms ({fl/b/1e9:.1f} TF)   drains {buf[0]} rounds {buf[1]} hits {buf[2]} (hits per lane-list {buf[2]/(512*4*64):.1f})")
