"""score_topk per call for evaluation batches of different sizes against the Beauty catalog: exact path / split path / split
path with the table prepared once."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recboard_amd import ops, lib
lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), "librecengine_dbg.so")   # the re_dbg_* switches live in the diagnostic twin (make -C recboard_amd/csrc dbg)
L = lib.load()
L.re_dbg_score_x2.argtypes = [ctypes.c_int]; L.re_dbg_score_x2.restype = None
L.re_dbg_score_reg_nub.argtypes = [ctypes.c_int64]; L.re_dbg_score_reg_nub.restype = None
L.re_dbg_score_variant.argtypes = [ctypes.c_int, ctypes.c_int64]; L.re_dbg_score_variant.restype = None
N, D = 12101, 64
g = torch.Generator(device="cuda").manual_seed(1)
E = torch.randn(N, D, device="cuda", generator=g)
prep = ops.score_prepare(E)
def t(fn, it=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / it
for U in (256, 512, 1024, 2048, 4096, 8192):
    q = torch.randn(U, D, device="cuda", generator=g)
    sp = torch.arange(0, U + 1, device="cuda") * 8
    si = torch.sort(torch.randint(0, N, (U, 8), device="cuda", generator=g), 1).values.reshape(-1)
    L.re_dbg_score_x2(0); t0 = t(lambda: ops.score_topk(q, E, sp, si, 50))
    L.re_dbg_score_x2(1); t1 = t(lambda: ops.score_topk(q, E, sp, si, 50)); t2 = t(lambda: ops.score_topk(q, E, sp, si, 50, prep=prep))
    row = []
    for minseg in (1, 4, 8, 16, 32):
        L.re_dbg_score_reg_nub(1); L.re_dbg_score_variant(3, minseg)
        v1, i1 = ops.score_topk(q, E, sp, si, 50, prep=prep)
        row.append(f"minseg {minseg}: {t(lambda: ops.score_topk(q, E, sp, si, 50, prep=prep))*1e3:6.1f}")
    L.re_dbg_score_reg_nub(16); L.re_dbg_score_variant(3, 1)
    v0, i0 = ops.score_topk(q, E, sp, si, 50)
    print("          split path forced (prepared), us:", "  ".join(row), " identical:", bool(torch.equal(i0, i1) and torch.equal(v0, v1)))
    print(f"B={U:5d}: exact-only {t0*1e3:7.1f} us   default {t1*1e3:7.1f} us   prepared {t2*1e3:7.1f} us   ({2*D*U*N/t2/1e9:.1f} TF)", flush=True)
