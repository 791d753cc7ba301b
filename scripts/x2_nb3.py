"""A/B of the three-buffer (two stages ahead) form of the split score kernel on the long-catalog shape."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recboard_amd import ops, lib
L = lib.load()
L.re_dbg_score_nb3.argtypes = [ctypes.c_int]; L.re_dbg_score_nb3.restype = None
def t(fn, it=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / it
g = torch.Generator(device="cuda").manual_seed(1)
for U, N, D in ((512, 12_500_000, 64), (512, 6_000_000, 128), (2048, 2_000_000, 64)):
    q = torch.randn(U, D, device="cuda", generator=g); E = torch.randn(N, D, device="cuda", generator=g)
    prep = ops.score_prepare(E)
    for on in (0, 1, 0, 1):
        L.re_dbg_score_nb3(on)
        print(f"{U} x {N} D={D} three buffers={on}: {t(lambda: ops.score_topk(q, E, None, None, 50, prep=prep)):.3f} ms", flush=True)
    del E, prep
L.re_dbg_score_nb3(1)
